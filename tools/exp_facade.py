"""Step time of the per-environment dict facade (gym_api.MultiGridEnv), the reference's own API shape."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_amd import gym_api  # noqa: E402

for name, agents in (("MultiGrid-Empty-8x8-v0", 2), ("MultiGrid-LockedHallway-4Rooms-v0", 4), ("MultiGrid-Empty-16x16-v0", 8)):
    env = gym_api.make(name, agents=agents, agent_view_size=7, layout_rng=np.random.default_rng(0))
    env.reset(seed=0)
    rng = np.random.default_rng(1)
    acts = rng.integers(0, 7, size=(3000, agents))
    for t in range(200):
        env.step({i: int(acts[t, i]) for i in range(agents)})
    t0 = time.perf_counter()
    n = 0
    for t in range(200, 2200):
        env.step({i: int(acts[t, i]) for i in range(agents)})
        n += 1
        if env.is_done():
            env.reset()
    dt = (time.perf_counter() - t0) / n
    print(f"{name} x {agents} agents: {dt * 1e6:.1f} us per step() of the dict facade (incl. is_done and resets)", flush=True)
    env.close()
