"""Long randomized parity run against the C oracle (test infrastructure, not a product path): many steps of mid-sized
batches with same-step auto-reset, biased action mixes that make pickups, drops, door toggles and hook events frequent.
    python tools/soak_parity.py [seconds per case]        (SOAK_SHARDS=2: through StreamShardedMultiGrid;
                                                            SOAK_FUSED=T: T steps per launch through mg_step_n)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_batch as tb
from multigrid_amd.batched import StepOutput

SHARDS = int(os.environ.get("SOAK_SHARDS", "1"))
FUSED = int(os.environ.get("SOAK_FUSED", "1"))
CASES = ["cfg2", "cfg3", "cfg4", "cfg5", "cfg5_nooverlap", "crowd8_bup", "crowd6_lh", "crowd8_rbd", "odd3", "odd5", "single", "lh8",
         "view9_bup", "view11_lh", "view9_see", "view3_rbd", "view5_lh"]
# the views other than 7 x 7 (the 9 x 9 and 11 x 11 ones run the kernel's general forms)
TABLE = dict(tb.CASES)
TABLE.update({
    "view9_bup": ("MultiGrid-BlockedUnlockPickup-v0", 2, dict(max_steps=60, agent_view_size=9)),
    "view11_lh": ("MultiGrid-LockedHallway-2Rooms-v0", 3, dict(max_steps=40, agent_view_size=11)),
    "view9_see": ("MultiGrid-Empty-8x8-v0", 4, dict(max_steps=40, agent_view_size=9, see_through_walls=True)),
    "view3_rbd": ("MultiGrid-RedBlueDoors-8x8-v0", 2, dict(max_steps=40, agent_view_size=3)),
    "view5_lh": ("MultiGrid-LockedHallway-4Rooms-v0", 4, dict(max_steps=50, agent_view_size=5)),
})
MIXES = [None, [1, 1, 3, 2, 2, 3, 1], [1, 1, 2, 4, 1, 4, 0]]  # weights over left right forward pickup drop toggle done


def run(case, seconds, e=3000, seed=0):
    spec, pool, env, ora = tb.make(case, e, table=TABLE, autoreset=2)
    if SHARDS > 1:  # the same batch as slices on streams of their own (mg_step_shards)
        from multigrid_amd.sharded import StreamShardedMultiGrid
        env.close()
        env = StreamShardedMultiGrid(spec, e, shards=SHARDS, device=0, autoreset=2)
        env.upload_layouts(pool[0], pool[1], pool[2])
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(seed)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    out = env.reset()
    idx = tb.expected_layout(ids_all, episode, n_l)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    t0, t, events = time.time(), 0, np.zeros(3, np.int64)
    while time.time() - t0 < seconds or t % FUSED:
        mix = MIXES[(t // 50) % len(MIXES)]
        if mix is None:
            actions = rng.integers(0, 7, size=(e, a)).astype(np.uint8)
        else:
            w = np.asarray(mix, float) / sum(mix)
            actions = rng.choice(7, size=(e, a), p=w).astype(np.uint8)
        order = np.argsort(rng.random(size=(e, a)), axis=-1).astype(np.uint8)
        if FUSED > 1:
            # T steps by one library call: this call draws the inputs of all of them, the oracle then takes them one by one
            if t % FUSED == 0:
                mixes = [MIXES[((t + k) // 50) % len(MIXES)] for k in range(FUSED)]
                acts_n = np.stack([rng.integers(0, 7, size=(e, a)) if m is None else
                                   rng.choice(7, size=(e, a), p=np.asarray(m, float) / sum(m)) for m in mixes]).astype(np.uint8)
                order_n = np.argsort(rng.random(size=(FUSED, e, a)), axis=-1).astype(np.uint8)
                out_n = env.step_n(acts_n, order_n)
                if isinstance(out_n, list):  # stream shards: one output per slice, [T, E_k, ...] each
                    import torch
                    out_n = StepOutput(*[torch.cat([getattr(o, f) for o in out_n], dim=1) for f in
                                         ("image", "direction", "reward", "terminated", "truncated")])
            k = t % FUSED
            actions, order = acts_n[k], order_n[k]
            out = StepOutput(*[None if x is None else x[k] for x in
                                  (out_n.image, out_n.direction, out_n.reward, out_n.terminated, out_n.truncated)])
        else:
            out = env.step(actions, order)
        res = ora.step(actions, order)
        tb.compare(out, res, f"{case} step {t}")
        events += [int((res[2] != 0).sum()), int(res[3].sum()), int(res[4].all(axis=1).sum())]
        done = res[3].all(axis=1) | res[4].all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = tb.expected_layout(ids, episode[ids], n_l)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
        if (t + 1) % FUSED == 0 and (t // FUSED) % max(1, 200 // FUSED) == 0:  # (the device is T steps ahead inside a fused call)
            tb.compare_state(env, ora, f"{case} state at step {t}")
        t += 1
    tb.compare_state(env, ora, f"{case} final")
    env.close()
    print(f"{case}: {t} steps x {e} envs bit-exact; rewards paid {events[0]}, terminations {events[1]}, truncated episodes {events[2]}", flush=True)


COMP_CASES = {
    "comp_1v1": ("DTDE-1v1", dict(max_steps=40)),
    "comp_2v2_dm": ("DTDE-2v2-Death-Match", dict(max_steps=50)),
    "comp_ctce_v3": ("CTCE-2v2", dict(max_steps=40, agent_view_size=3)),
    "comp_ctde_all": ("CTDE-2v2", dict(max_steps=40, success_termination_mode="all", allow_agent_overlap=True, death_match=True)),
    "comp_obstacle": ("DTDE-Red-Single-with-Obstacle", dict(max_steps=30)),
}


def run_comp(case, seconds, e=3000, seed=0):
    """CompetativeRedBlueDoor environments: as run(), with the step's info bytes compared too; every 40 steps the agents of
    half of the environments are moved next to each other, to the keys and in front of the doors, so that eliminations,
    pickups and door openings keep happening."""
    import test_gpu_competitive as tc
    name, kw = COMP_CASES[case]
    spec, pool, env, ora = tc.make(name, e, autoreset=2, **kw)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(seed)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    env.reset()
    idx = tb.expected_layout(ids_all, episode, n_l)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    t0, t, events = time.time(), 0, np.zeros(4, np.int64)
    while time.time() - t0 < seconds:
        if t % 40 == 0:
            tc.crowd(ora, rng)
            env.set_state(grid=ora.grid, agents=ora.agents)
        actions = rng.choice(7, size=(e, a), p=tc.PICKY).astype(np.uint8)
        order = np.argsort(rng.random((e, a)), axis=1).astype(np.uint8)
        out = env.step(actions, order)
        res = ora.step(actions, order)
        tb.compare(out, res, f"{case} step {t}")
        tb.compare_info(out, ora.info, f"{case} step {t}")
        events += [int((res[2] > 0.6).sum()), int((res[2] < -0.5).sum()), int(res[3].sum()), int((ora.info >> 6 != 0).sum())]
        done = res[3].all(axis=1) | res[4].all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = tb.expected_layout(ids, episode[ids], n_l)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
        if t % 100 == 0:
            tb.compare_state(env, ora, f"{case} state at step {t}")
        t += 1
    tb.compare_state(env, ora, f"{case} final")
    env.close()
    print(f"{case}: {t} steps x {e} envs bit-exact (info bytes too); doors opened for a win {events[0]}, eliminations "
          f"{events[1]}, terminations {events[2]}, rewarded pickups {events[3]}", flush=True)


if __name__ == "__main__":
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    only = [c for c in os.environ.get("SOAK_CASES", "").split(",") if c]  # a subset, by name
    for i, c in enumerate(CASES):
        if not only or c in only:
            run(c, secs, seed=i)
    for i, c in enumerate(COMP_CASES):
        if (not only or c in only) and SHARDS == 1 and FUSED == 1:
            run_comp(c, secs, seed=100 + i)
    print("soak ok")
