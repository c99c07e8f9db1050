"""Per-phase clock stamps from the diagnostic build (tools/libmgrid_exp.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MGRID_TUNING", "1")  # MGRID_DBG and MGRID_TILE are read only with this opt-in
os.environ["MGRID_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmgrid_exp.so")
import numpy as np, torch
from multigrid_amd import _lib, envs, layouts
from multigrid_amd.batched import BatchedMultiGrid

NAMES = {1: "hdr->LDS", 2: "act", 3: "resets+scalars", 4: "pass0: grid->LDS", 5: "pass0: patch+front cells", 6: "pass0: overlay",
         7: "pass0: observe", 8: "pass0: copy-out", 9: "remaining passes", 10: "hdr->HBM"}

def run(cfg, agents, e, **kw):
    spec, gen = envs.make_spec(cfg, num_agents=agents, agent_view_size=7, **kw)
    env = BatchedMultiGrid(spec, e, device=0, autoreset=1, seed=1)
    pool = layouts.make_pool(gen, 256)
    env.upload_layouts(*pool[:3]); env.reset()
    L = _lib.lib()
    nt = ctypes.c_int()
    buf = torch.zeros((1 << 16, 32), dtype=torch.int64, device=env.device)
    acts = torch.randint(0, 7, (8, e, agents), device=env.device, dtype=torch.uint8)
    for t in range(30): env.step(acts[t % 8])
    L.mg_dbg_buffer(env._h, ctypes.c_void_p(buf.data_ptr()), ctypes.byref(nt))
    if os.environ.get('GENOBS'): env.gen_obs()
    else: env.step(acts[0])
    torch.cuda.synchronize()
    b = buf[:nt.value].cpu().numpy().astype(np.float64)
    total = (b[:, 10] - b[:, 0]).mean()
    if total > 0:  # when do wavefronts start and end, relative to the first start (one clock domain assumed)
        t0 = b[:, 0].min()
        st, en, life = b[:, 0] - t0, b[:, 10] - t0, b[:, 10] - b[:, 0]
        q = lambda x: " / ".join(f"{np.percentile(x, p):.0f}" for p in (0, 50, 90, 99, 100))
        print(f"   start  (min / median / p90 / p99 / max): {q(st)} clk")
        print(f"   end    (min / median / p90 / p99 / max): {q(en)} clk")
        print(f"   life   (min / median / p90 / p99 / max): {q(life)} clk")
    print(f"{cfg} A={agents} E={e} tiles={nt.value} tile={os.environ.get('MGRID_TILE','auto')}: mean wave lifetime {total:.0f} clk")
    prev = 0
    for k in range(1, 11):
        d = (b[:, k] - b[:, prev]).mean()
        print(f"   {NAMES[k]:28s} {d:9.0f} clk  {100*d/total:5.1f}%")
        prev = k
    if int(os.environ.get("MGRID_DBG", "0")) & 16:
        n = b[:, 19].sum()
        print(f"   light ticks per wave: staging {b[:,20].mean():.0f}  act {b[:,21].mean():.0f}  same-step restarts {b[:,22].mean():.0f}  whole {b[:,23].mean():.0f} clk")
        print(f"   start -> scalars written {b[:,31].mean():.0f}  then to the end {b[:,30].mean():.0f} clk")
        print(f"   light ticks per pass: grids+patch+log+front+draw+observe {b[:,16].sum()/n:.0f}  commit {b[:,17].sum()/n:.0f}  copy-out {b[:,18].sum()/n:.0f}  act inside the pass {b[:,20+0].sum()*0 + b[:,16+4].sum()*0:.0f} clk ({n:.0f} passes)")
    env.close()

if __name__ == "__main__":
    cases = (("MultiGrid-Empty-16x16-v0", 8, 131072), ("MultiGrid-BlockedUnlockPickup-v0", 2, 65536), ("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536), ("MultiGrid-RedBlueDoors-8x8-v0", 2, 4096),
             ("MultiGrid-Empty-16x16-v0", 8, 131072, dict(allow_agent_overlap=False)))
    for i in [int(x) for x in os.environ.get("CASES", "0").split(",")]:
        run(*cases[i][:3], **(cases[i][3] if len(cases[i]) > 3 else {}))
