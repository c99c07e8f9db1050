"""Time to fill a pool of start states: on the device (mg_generate_layouts) and with the host generators."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import envs, layouts
from multigrid_amd.batched import BatchedMultiGrid

for name, agents in (("MultiGrid-Empty-Random-6x6-v0", 3), ("MultiGrid-RedBlueDoors-8x8-v0", 2),
                     ("MultiGrid-BlockedUnlockPickup-v0", 2), ("MultiGrid-LockedHallway-4Rooms-v0", 4)):
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7)
    env = BatchedMultiGrid(spec, 64, device=0)
    env.generate_layouts(gen, 1024, seed=0)
    torch.cuda.synchronize()
    n = 65536
    t0 = time.perf_counter()
    env.generate_layouts(gen, n, seed=1)  # synchronous
    dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    layouts.make_pool(gen, 512, seed0=0)
    host = (time.perf_counter() - t0) / 512
    print(f"{name:36s} device {n} layouts in {dev * 1e3:7.2f} ms ({dev / n * 1e9:6.0f} ns each); host generator {host * 1e6:7.0f} us each", flush=True)
    env.close()
