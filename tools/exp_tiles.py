"""Tile-shape sweep: time mg_step for several (NE, G) shapes per configuration (MGRID_TILE override)."""
import os, sys
os.environ["MGRID_TUNING"] = "1"  # the library reads MGRID_TILE only with this opt-in
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import envs, layouts
from multigrid_amd.batched import BatchedMultiGrid

CASES = {
    2: ("MultiGrid-RedBlueDoors-8x8-v0", 2, 4096),
    3: ("MultiGrid-BlockedUnlockPickup-v0", 2, 65536),
    4: ("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536),
    5: ("MultiGrid-Empty-16x16-v0", 8, 131072),
}


def run(cid, tile, steps=300):
    cfg, agents, e = CASES[cid]
    if tile:
        os.environ["MGRID_TILE"] = tile
    else:
        os.environ.pop("MGRID_TILE", None)
    spec, gen = envs.make_spec(cfg, num_agents=agents, agent_view_size=7)
    env = BatchedMultiGrid(spec, e, device=0, autoreset=1, seed=1)
    pool = layouts.make_pool(gen, 256)
    env.upload_layouts(*pool[:3]); env.reset()
    acts = torch.randint(0, 7, (8, e, agents), device=env.device, dtype=torch.uint8)
    for t in range(30): env.step(acts[t % 8])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): env.step(acts[t % 8])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    print(f"cfg{cid} A={agents} E={e:7d} tile={tile or 'auto':6s} {us:8.1f} us/launch", flush=True)
    env.close()


if __name__ == "__main__":
    sel = [int(x) for x in os.environ.get("CASES", "2,3,4").split(",")]
    for cid in sel:
        a = CASES[cid][1]
        run(cid, None)
        for ne in (64 // a, 32 // a, 16 // a, 8 // a):
            for g in (1, 2, 4):
                if ne >= 1 and ne * g <= 64:
                    run(cid, f"{ne},{g}")
