"""Phase-cost experiment: time mg_step with (a) random actions, (b) all 'done' actions, (c) no image output."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multigrid_amd import _lib, envs, layouts
from multigrid_amd.batched import BatchedMultiGrid, StepOutput

def run(cfg, agents, e, variant, steps=200):
    spec, gen = envs.make_spec(cfg, num_agents=agents, agent_view_size=7)
    env = BatchedMultiGrid(spec, e, device=0, autoreset=1, seed=1)
    pool = layouts.make_pool(gen, 256)
    env.upload_layouts(*pool[:3]); env.reset()
    dev = env.device
    acts = torch.randint(0, 7, (8, e, agents), device=dev, dtype=torch.uint8)
    if variant == "done": acts[:] = 6
    if variant == "turn": acts[:] = torch.randint(0, 2, acts.shape, device=dev, dtype=torch.uint8)
    out = None
    if variant == "noimg":
        out = StepOutput(image=None, direction=env.out.direction, reward=env.out.reward, terminated=env.out.terminated, truncated=env.out.truncated)
        class O: pass
    def step(t):
        if variant == "noimg":
            import ctypes
            from multigrid_amd._lib import MgOut, check
            mo = MgOut(None, env.out.direction.data_ptr(), env.out.reward.data_ptr(), env.out.terminated.data_ptr(), env.out.truncated.data_ptr())
            check(_lib.lib().mg_step(env._h, ctypes.c_void_p(acts[t % 8].data_ptr()), None, ctypes.byref(mo), env._stream()), env._h)
        else:
            env.step(acts[t % 8])
    for t in range(20): step(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): step(t)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    print(f"{cfg:36s} A={agents} E={e:7d} {variant:6s} {us:8.1f} us/launch  {e*agents/us/1e3:7.2f} G agent-steps/s", flush=True)
    env.close()

if __name__ == "__main__":
    cases = (("MultiGrid-Empty-16x16-v0", 8, 131072), ("MultiGrid-BlockedUnlockPickup-v0", 2, 65536), ("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536), ("MultiGrid-RedBlueDoors-8x8-v0", 2, 4096))
    sel = [int(x) for x in os.environ.get("CASES", "0,1,2,3").split(",")]
    variants = os.environ.get("VARIANTS", "random,done,turn,noimg").split(",")
    for i in sel:
        cfg, a, e = cases[i]
        for v in variants:
            run(cfg, a, e, v)
