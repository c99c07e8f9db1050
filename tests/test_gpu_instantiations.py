"""Every instantiation of the tile kernel that the library was built with is launched here, on purpose and against the
oracle: view sizes 3 / 5 / 7 / 9 / 11, with and without sight through walls, step / gen_obs / reset, grids known to be walled or
not, one-hot output, several steps per launch, CompetativeRedBlueDoor steps, the cell-table act phase and the three
staging widths.  The library's own launch census (mg_launch_census) then has to name every kernel of the build's register
report (multigrid_amd/kernel_registers.json): an instantiation nothing here reaches is one nobody has checked, and
fails the test - remove it from the dispatch or add its case."""
import numpy as np
import pytest

from multigrid_amd import _lib, envs, layouts

pytestmark = pytest.mark.gpu

V3 = "MultiGrid-CompetativeRedBlueDoor-v3-"


def stepper(spec, gen, e, seed, one_hot=False, n_layouts=24, **kw):
    from multigrid_amd.batched import BatchedMultiGrid
    from oracle.coracle import BatchOracle
    pool = layouts.make_pool(gen, n_layouts, seed0=seed)
    env = BatchedMultiGrid(spec, e, device=0, one_hot=one_hot, **kw)
    env.upload_layouts(pool[0], pool[1], pool[2])
    return pool, env, BatchOracle(spec, e)


def check_step(env, ora, rng, where, order=True, one_hot=False):
    import torch
    from oracle import pyref
    e, a = env.E, env.A
    actions = rng.choice(7, size=(e, a), p=[0.1, 0.1, 0.34, 0.14, 0.1, 0.2, 0.02]).astype(np.uint8)
    od = np.argsort(rng.random((e, a)), axis=1).astype(np.uint8) if order else None
    out = env.step(actions, od)
    img, direction, rew, term, trunc = ora.step(actions, od)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.image.cpu().numpy(), img, err_msg=where + " image")
    np.testing.assert_array_equal(out.direction.cpu().numpy(), direction, err_msg=where + " direction")
    np.testing.assert_allclose(out.reward.cpu().numpy(), rew, rtol=0, atol=1e-6, err_msg=where + " reward")
    np.testing.assert_array_equal(out.terminated.cpu().numpy(), term, err_msg=where + " terminated")
    np.testing.assert_array_equal(out.truncated.cpu().numpy(), trunc, err_msg=where + " truncated")
    if one_hot:
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), pyref.one_hot_batch(img), err_msg=where + " one-hot")


def check_obs(env, ora, where, one_hot=False):
    import torch
    from oracle import pyref
    out = env.gen_obs()
    torch.cuda.synchronize()
    want = ora.gen_obs()
    np.testing.assert_array_equal(out.image.cpu().numpy(), want, err_msg=where + " gen_obs")
    if one_hot:
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), pyref.one_hot_batch(want), err_msg=where + " gen_obs one-hot")


def check_reset(env, ora, pool, rng, where, one_hot=False):
    import torch
    from oracle import pyref
    idx = rng.integers(0, len(pool[0]), size=env.E)
    out = env.reset(layout_idx=idx)
    ora.load(np.arange(env.E), pool[0][idx], pool[1][idx], pool[2][idx])
    torch.cuda.synchronize()
    want = ora.gen_obs()
    np.testing.assert_array_equal(out.image.cpu().numpy(), want, err_msg=where + " reset")
    if one_hot:
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), pyref.one_hot_batch(want), err_msg=where + " reset one-hot")


def check_step_n(env, ora, rng, where, t_steps=3):
    import torch
    e, a = env.E, env.A
    actions = rng.integers(0, 7, size=(t_steps, e, a)).astype(np.uint8)
    od = np.argsort(rng.random((t_steps, e, a)), axis=2).astype(np.uint8)
    out = env.step_n(actions, od)
    torch.cuda.synchronize()
    for t in range(t_steps):
        img, direction, rew, term, trunc = ora.step(actions[t], od[t])
        np.testing.assert_array_equal(out.image[t].cpu().numpy(), img, err_msg=f"{where} fused step {t} image")
        np.testing.assert_allclose(out.reward[t].cpu().numpy(), rew, rtol=0, atol=1e-6, err_msg=f"{where} fused step {t} reward")
        np.testing.assert_array_equal(out.terminated[t].cpu().numpy(), term, err_msg=f"{where} fused step {t} terminated")


def unwall(env):
    """Hands the library its own grids back through mg_set_state: the state is the same, but the library no longer
    knows every grid to be ringed by walls and launches the general-border kernels."""
    st = env.get_state()
    env.set_state(grid=st["grid"])


def sweep(view, see, monkeypatch):
    rng = np.random.default_rng(100 * view + see)
    monkeypatch.setenv("MGRID_TUNING", "1")
    e = 330
    # plain steps at the three staging widths, walled and general grids; gen_obs; reset; fused steps
    monkeypatch.setenv("MGRID_TILE", "16,1")  # full sub-tiles of 64 agent lanes: every staging width takes its fast path
    for stage in ("16", "32", "64"):
        monkeypatch.setenv("MGRID_STAGE", stage)
        spec, gen = envs.make_spec("MultiGrid-BlockedUnlockPickup-v0", num_agents=4, agent_view_size=view, max_steps=40)
        spec["see_through_walls"] = bool(see)
        pool, env, ora = stepper(spec, gen, e, seed=view)
        where = f"view {view} see {see} stage {stage}"
        check_reset(env, ora, pool, rng, where)           # reset, grids not yet known to be walled
        for _ in range(3):
            check_step(env, ora, rng, where + " walled")  # step, walled
        check_obs(env, ora, where + " walled")            # gen_obs, walled
        check_reset(env, ora, pool, rng, where)           # reset, walled
        check_step_n(env, ora, rng, where + " walled")    # fused, walled
        unwall(env)
        for _ in range(3):
            check_step(env, ora, rng, where + " general")
        check_obs(env, ora, where + " general")
        check_step_n(env, ora, rng, where + " general")
        env.close()
    monkeypatch.delenv("MGRID_STAGE")
    monkeypatch.delenv("MGRID_TILE")
    # one-hot output: step, gen_obs, reset
    spec, gen = envs.make_spec("MultiGrid-LockedHallway-2Rooms-v0", num_agents=3, agent_view_size=view, max_steps=40)
    spec["see_through_walls"] = bool(see)
    pool, env, ora = stepper(spec, gen, 150, seed=view + 1, one_hot=True)
    where = f"view {view} see {see} one-hot"
    check_reset(env, ora, pool, rng, where, one_hot=True)
    for _ in range(2):
        check_step(env, ora, rng, where, one_hot=True)
    check_obs(env, ora, where, one_hot=True)
    env.close()
    # the cell-table act phase (6 to 8 agents that may not overlap): one and several steps per launch
    spec, gen = envs.make_spec("MultiGrid-Empty-Random-6x6-v0", num_agents=7, agent_view_size=view, max_steps=40,
                               allow_agent_overlap=False, success_termination_mode="all")
    spec["see_through_walls"] = bool(see)
    pool, env, ora = stepper(spec, gen, e, seed=view + 2)
    where = f"view {view} see {see} cell table"
    check_reset(env, ora, pool, rng, where)
    for _ in range(3):
        check_step(env, ora, rng, where + " walled")
    check_step_n(env, ora, rng, where + " walled")
    unwall(env)
    for _ in range(3):
        check_step(env, ora, rng, where + " general")
    check_step_n(env, ora, rng, where + " general")
    env.close()
    # CompetativeRedBlueDoor steps, one and several per launch
    spec, gen = envs.make_spec(V3 + "DTDE-2v2", agent_view_size=view)
    spec["see_through_walls"] = bool(see)
    pool, env, ora = stepper(spec, gen, e, seed=view + 3)
    where = f"view {view} see {see} competitive"
    check_reset(env, ora, pool, rng, where)
    for _ in range(3):
        check_step(env, ora, rng, where)
    check_step_n(env, ora, rng, where)
    env.close()


def sweep_wide(view, see):
    """9 x 9 and 11 x 11 views run the kernel's general forms only: plain steps, gen_obs and reset, the same with
    one-hot images, CompetativeRedBlueDoor steps; several steps per call are queued as single launches."""
    rng = np.random.default_rng(100 * view + see)
    spec, gen = envs.make_spec("MultiGrid-BlockedUnlockPickup-v0", num_agents=4, agent_view_size=view, max_steps=40)
    spec["see_through_walls"] = bool(see)
    pool, env, ora = stepper(spec, gen, 330, seed=view)
    where = f"view {view} see {see}"
    check_reset(env, ora, pool, rng, where)
    for _ in range(4):
        check_step(env, ora, rng, where)
    check_obs(env, ora, where)
    check_step_n(env, ora, rng, where)
    env.close()
    spec, gen = envs.make_spec("MultiGrid-Empty-Random-6x6-v0", num_agents=7, agent_view_size=view, max_steps=40,
                               allow_agent_overlap=False, success_termination_mode="all")
    spec["see_through_walls"] = bool(see)
    pool, env, ora = stepper(spec, gen, 200, seed=view + 2)
    check_reset(env, ora, pool, rng, where + " 7 agents")
    for _ in range(4):
        check_step(env, ora, rng, where + " 7 agents")
    env.close()
    spec, gen = envs.make_spec("MultiGrid-LockedHallway-2Rooms-v0", num_agents=3, agent_view_size=view, max_steps=40)
    spec["see_through_walls"] = bool(see)
    pool, env, ora = stepper(spec, gen, 150, seed=view + 1, one_hot=True)
    check_reset(env, ora, pool, rng, where + " one-hot", one_hot=True)
    for _ in range(2):
        check_step(env, ora, rng, where + " one-hot", one_hot=True)
    check_obs(env, ora, where + " one-hot", one_hot=True)
    env.close()
    spec, gen = envs.make_spec(V3 + "DTDE-2v2", agent_view_size=view)
    spec["see_through_walls"] = bool(see)
    pool, env, ora = stepper(spec, gen, 330, seed=view + 3)
    check_reset(env, ora, pool, rng, where + " competitive")
    for _ in range(3):
        check_step(env, ora, rng, where + " competitive")
    check_step_n(env, ora, rng, where + " competitive")
    env.close()


@pytest.mark.parametrize("see", [0, 1])
@pytest.mark.parametrize("view", [3, 5, 7, 9, 11])
def test_every_dispatch_path_against_the_oracle(view, see, monkeypatch):
    if view > 7:
        sweep_wide(view, see)
    else:
        sweep(view, see, monkeypatch)


def test_zz_every_compiled_instantiation_was_launched(monkeypatch):
    """Runs after the sweeps above (and repeats them when run alone): the census names every compiled instantiation."""
    compiled = _lib.compiled_instantiations()
    assert len(compiled) > 50, "the build's register report lists no tile kernels"
    if not compiled <= set(_lib.launch_census()):
        for view in (3, 5, 7):
            for see in (0, 1):
                sweep(view, see, monkeypatch)
        for view in (9, 11):
            for see in (0, 1):
                sweep_wide(view, see)
    launched = set(_lib.launch_census())
    names = "V,SEE,MODE,WALLED,OH,MULTI,COMP,CROWD,STG"
    never = sorted(compiled - launched)
    assert not never, f"{len(never)} compiled instantiation(s) ({names}) were never launched: {never[:12]}"
    stray = sorted(launched - compiled)
    assert not stray, f"launched but not in the build's register report (stale kernel_registers.json?): {stray[:12]}"
