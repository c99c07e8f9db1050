"""Plain device copy / fill rates at the sizes of one step, for comparison with the step kernel."""
import torch

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for mb in (77, 154, 308, 1024):
    n = mb * 1000 * 1000
    src = torch.empty(n, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
    s32 = src.view(torch.int32)
    us_f = t(lambda: dst.fill_(1)); us_c = t(lambda: dst.copy_(src)); us_r = t(lambda: s32.sum())
    print(f"{mb:5d} MB: fill {us_f:7.1f} us ({n/us_f/1e6:5.2f} TB/s)  copy {us_c:7.1f} us ({2*n/us_c/1e6:5.2f} TB/s)  read-sum {us_r:7.1f} us ({n/us_r/1e6:5.2f} TB/s)", flush=True)
