"""Step time of the SyncVectorEnv-style adapter (tensor mode) next to the bare step launch it wraps."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd.vector import MultiGridVectorEnv  # noqa: E402

for name, agents, e in (("MultiGrid-Empty-16x16-v0", 8, 131072), ("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536),
                        ("MultiGrid-BlockedUnlockPickup-v0", 2, 65536), ("MultiGrid-RedBlueDoors-8x8-v0", 2, 4096)):
    env = MultiGridVectorEnv(name, e, agents=agents, agent_view_size=7, num_layouts=256)
    env.reset(seed=0)
    acts = torch.randint(0, 7, (16, e, agents), device="cuda:0", dtype=torch.uint8)
    for which in ("adapter", "bare"):
        fn = env.step if which == "adapter" else env._sim.step
        for t in range(300):
            fn(acts[t % 16])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(1000):
            fn(acts[t % 16])
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name} {e} x {agents}: {which:8s} {dt / 1000 * 1e6:7.1f} us per step (host issue {t_issue / 1000 * 1e6:6.1f} us)", flush=True)
    env.close()
