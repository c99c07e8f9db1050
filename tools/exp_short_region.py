"""Where the time of a short timed region goes: per-step GPU durations of 20 steps after a device synchronisation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import _lib, envs, layouts
from multigrid_amd.batched import BatchedMultiGrid
spec, gen = envs.make_spec("MultiGrid-Empty-16x16-v0", num_agents=8, agent_view_size=7)
e = 131072
env = BatchedMultiGrid(spec, e, device=0, autoreset=_lib.AUTORESET_SAME_STEP, seed=1234)
env.upload_layouts(*layouts.make_pool(gen, 1024, seed0=0)[:3]); env.reset()
scatter = "--no-scatter" not in sys.argv
if scatter:
    env.set_state(steps=torch.randint(0, spec["max_steps"], (e,), device="cuda", dtype=torch.int32))
acts = torch.randint(0, 7, (16, e, 8), device="cuda", dtype=torch.uint8)
for t in range(int(os.environ.get('WARM', 5))): env.step(acts[t % 16])
for rep in range(3):
    torch.cuda.synchronize()
    if "--sleep" in sys.argv: time.sleep(0.05)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
    h0 = time.perf_counter()
    evs[0].record()
    host = []
    for t in range(20):
        a = time.perf_counter(); env.step(acts[t % 16]); host.append((time.perf_counter() - a) * 1e6); evs[t + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - h0) * 1e6
    d = [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(20)]
    print(f"rep {rep} scatter {scatter}: wall {wall:.0f} us, events total {sum(d):.0f} us; per step us:", [round(x) for x in d])
    print("   host enqueue us per step:", [round(x) for x in host])
