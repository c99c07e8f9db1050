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
    life = b[:, 10] - b[:, 0]
    slow = life >= np.percentile(life, 97)  # the slowest wavefronts: where do they lose their time?
    for k in range(1, 11):
        ph = b[:, k] - b[:, prev]
        d = ph.mean()
        print(f"   {NAMES[k]:28s} {d:9.0f} clk  {100*d/total:5.1f}%   median {np.median(ph):6.0f}  p99 {np.percentile(ph, 99):6.0f}"
              f"  max {ph.max():6.0f}   slowest 3 % of wavefronts: {ph[slow].mean():6.0f}")
        prev = k
    if not int(os.environ.get("MGRID_DBG", "0")) & 16:
        act = b[:, 2] - b[:, 1]
        print("   act phase by what the wavefront met (slots 24, 26-29 of the diagnostic buffer):")
        rep = b[:, 26] > 0
        for nm, col in (("rows in, first evaluation, marking", 11), ("finish, log, rows out", 12), ("order and places", 13), ("rounds", 14)):
            print(f"     {nm:36s} no marked agents {b[~rep, col].mean():6.0f} clk   with marked agents {b[rep, col].mean() if rep.any() else 0:6.0f} clk")
        tail = b[:, 2] - b[:, 15]
        print(f"     {'hooks, flags, write-back (last call)':36s} no marked agents {tail[~rep].mean():6.0f} clk   with marked agents {tail[rep].mean() if rep.any() else 0:6.0f} clk")
        for name, col in (("replay rounds", 26), ("marked agents", 24), ("cells written", 27), ("rewards", 28), ("unstable", 29)):
            vals = np.unique(b[:, col])
            print(f"     {name:14s}: " + "  ".join(f"{int(v)}: {act[b[:, col] == v].mean():.0f} clk ({(b[:, col] == v).mean()*100:.0f} %)" for v in vals[:10]))
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
