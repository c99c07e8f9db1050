"""Timing of the step launch with one-hot observations (config 5 shard)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import envs, layouts
from multigrid_amd.batched import BatchedMultiGrid

def run(cfg, agents, e, compact, steps=100):
    spec, gen = envs.make_spec(cfg, num_agents=agents, agent_view_size=7)
    env = BatchedMultiGrid(spec, e, device=0, autoreset=1, seed=1, one_hot=True, compact_image=compact)
    pool = layouts.make_pool(gen, 256)
    env.upload_layouts(*pool[:3]); env.reset()
    acts = torch.randint(0, 7, (8, e, agents), device=env.device, dtype=torch.uint8)
    for t in range(10): env.step(acts[t % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): env.step(acts[t % 8])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    gb = e * agents * 49 * 24 / 1e9
    print(f"{cfg} A={agents} E={e} one-hot compact={compact}: {us:8.1f} us/launch, one-hot stores alone {gb / us * 1e6:7.0f} GB/s", flush=True)
    env.close()

if __name__ == "__main__":
    run("MultiGrid-Empty-16x16-v0", 8, 131072, False)
    run("MultiGrid-Empty-16x16-v0", 8, 131072, True)
    run("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536, False)
    run("MultiGrid-BlockedUnlockPickup-v0", 2, 65536, False)
    run("MultiGrid-Empty-8x8-v0", 4, 65536, False)
    run("MultiGrid-RedBlueDoors-8x8-v0", 8, 65536, False)
