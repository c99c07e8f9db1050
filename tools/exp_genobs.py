"""Time mg_gen_obs alone (header + grid in, images out: the observation half of a step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import envs, layouts
from multigrid_amd.batched import BatchedMultiGrid

def run(cfg, agents, e, steps=200):
    spec, gen = envs.make_spec(cfg, num_agents=agents, agent_view_size=7)
    env = BatchedMultiGrid(spec, e, device=0, autoreset=1, seed=1)
    pool = layouts.make_pool(gen, 256)
    env.upload_layouts(*pool[:3]); env.reset()
    acts = torch.randint(0, 7, (8, e, agents), device=env.device, dtype=torch.uint8)
    for t in range(40): env.step(acts[t % 8])
    for t in range(10): env.gen_obs()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): env.gen_obs()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    print(f"{cfg:36s} A={agents} E={e:7d} tile={os.environ.get('MGRID_TILE','auto'):5s} gen_obs {us:8.1f} us/launch", flush=True)
    env.close()

if __name__ == "__main__":
    cases = (("MultiGrid-Empty-16x16-v0", 8, 131072), ("MultiGrid-BlockedUnlockPickup-v0", 2, 65536), ("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536))
    for i in [int(x) for x in os.environ.get("CASES", "0").split(",")]:
        run(*cases[i])
