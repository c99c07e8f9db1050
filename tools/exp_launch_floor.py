"""Back-to-back launch interval of trivial kernels on this GPU, for comparison with the one-environment step."""
import time

import torch

dev = torch.device("cuda", 0)
for n in (64, 2048 * 64, 1 << 22):
    x = torch.zeros(n, dtype=torch.float32, device=dev)
    for _ in range(200):
        x.add_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(2000):
        x.add_(1.0)
    e1.record()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"add_ on {n} floats: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us per launch by events, host issue {t_issue / 2000 * 1e6:.2f} us", flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for _ in range(50):
                x.add_(1.0)
        torch.cuda.synchronize()
        e0.record(s)
        for _ in range(40):
            g.replay()
        e1.record(s)
    torch.cuda.synchronize()
    print(f"  graph of 50: {e0.elapsed_time(e1) / 2000 * 1e3:.2f} us per launch", flush=True)
