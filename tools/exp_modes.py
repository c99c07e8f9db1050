"""Launch times of the tile kernel's other modes (gen_obs, reset) and of the adapters' small kernels, with the bytes
they move, on the BASELINE per-GPU batches.    python tools/exp_modes.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from multigrid_amd import _lib, envs, layouts
from multigrid_amd.batched import BatchedMultiGrid
from multigrid_amd.vector import MultiGridVectorEnv
from multigrid_amd.views import BatchedParallelView

CASES = (("MultiGrid-BlockedUnlockPickup-v0", 2, 65536), ("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536),
         ("MultiGrid-Empty-16x16-v0", 8, 131072))


def timed(fn, n=200, warm=30):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


for name, agents, e in CASES:
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7)
    env = BatchedMultiGrid(spec, e, device=0, autoreset=1)
    pool = layouts.make_pool(gen, 256)
    env.upload_layouts(*pool[:3])
    env.reset()
    w, h, a = spec["width"], spec["height"], agents
    hs = (16 + 8 * (a + a % 2) + 2 * (a + 2) + 15) // 16 * 16
    gs = (2 * w * h + 15) // 16 * 16
    img = 147 * a
    idx = torch.randint(0, 256, (e,), device=env.device, dtype=torch.int32)
    t_obs = timed(lambda: env.gen_obs())
    t_rst = timed(lambda: env.reset(layout_idx=idx))
    b_obs, b_rst = hs + gs + img + a, 2 * (hs + gs) + img + a
    print(f"{name} A={a} E={e}: gen_obs {t_obs:6.1f} us ({e * b_obs / t_obs / 1e3:6.0f} GB/s, {e * b_obs / t_obs / 8e6:.3f} of 8 TB/s)   "
          f"reset of every environment {t_rst:6.1f} us ({e * b_rst / t_rst / 1e3:6.0f} GB/s, {e * b_rst / t_rst / 8e6:.3f})")
    view = BatchedParallelView(env)
    t_mask = timed(lambda: view.agents_mask)
    rew = torch.zeros((e, a), dtype=torch.float32, device=env.device)
    flags = torch.zeros((e, a), dtype=torch.uint8, device=env.device)
    ret = torch.zeros((e, a), dtype=torch.float64, device=env.device)
    ln = torch.zeros((e,), dtype=torch.int64, device=env.device)
    done = torch.zeros((e,), dtype=torch.uint8, device=env.device)
    epr, epl = torch.zeros_like(ret), torch.zeros_like(ln)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    L = _lib.lib()
    t_eps = timed(lambda: L.mg_episode_stats(e, a, p(rew), p(flags), p(flags), p(ret), p(ln), p(done), p(epr), p(epl), env._stream()))
    b_eps, b_mask = 6 * a + 2 * (8 * a + 8) + 8 * a + 9, 8 * a + 4 + a + 1
    print(f"    mg_episode_stats {t_eps:5.1f} us ({e * b_eps / t_eps / 1e3:5.0f} GB/s)   mg_agents_mask {t_mask:5.1f} us ({e * b_mask / t_mask / 1e3:5.0f} GB/s)")
    env.close()
