"""HIP path vs the C oracle on large, multi-tile batches: resets, auto-reset modes, on-device agent
order, odd alignments, statistics.  Everything goes through the C ABI (multigrid_amd.batched)."""
import numpy as np
import pytest

from multigrid_amd import envs, layouts

pytestmark = pytest.mark.gpu

BIASED = np.array([0.12, 0.12, 0.36, 0.12, 0.08, 0.18, 0.02])
LAYOUT_STRIDE = 7919

CASES = {
    "cfg1": ("MultiGrid-Empty-8x8-v0", 2, dict(max_steps=48)),
    "cfg2": ("MultiGrid-RedBlueDoors-8x8-v0", 2, dict(max_steps=40)),
    "cfg3": ("MultiGrid-BlockedUnlockPickup-v0", 2, dict(max_steps=60)),
    "cfg4": ("MultiGrid-LockedHallway-4Rooms-v0", 4, dict(max_steps=50)),
    "cfg5": ("MultiGrid-Empty-16x16-v0", 8, dict(max_steps=64)),
    "cfg5_nooverlap": ("MultiGrid-Empty-16x16-v0", 8, dict(max_steps=64, allow_agent_overlap=False)),
    "odd3": ("MultiGrid-Empty-Random-6x6-v0", 3, dict(max_steps=30, success_termination_mode="all")),
    "odd5": ("MultiGrid-LockedHallway-2Rooms-v0", 5, dict(max_steps=30)),
    "single": ("MultiGrid-BlockedUnlockPickup-v0", 1, dict(max_steps=30)),
    # more than 8 agents take the kernel's generic (LDS-array) act phase
    "big12": ("MultiGrid-Empty-16x16-v0", 12, dict(max_steps=40, allow_agent_overlap=False, success_termination_mode="all")),
    "big10": ("MultiGrid-LockedHallway-6Rooms-v0", 10, dict(max_steps=40)),
    # crowded rooms: agents block each other and reach for the same objects within one step, which is
    # what separates the lane-per-agent act phase from the in-order replay
    "crowd8_empty": ("MultiGrid-Empty-5x5-v0", 8, dict(max_steps=30, allow_agent_overlap=False, success_termination_mode="all")),
    "crowd8_rbd": ("MultiGrid-RedBlueDoors-6x6-v0", 8, dict(max_steps=40, allow_agent_overlap=False)),
    "crowd8_bup": ("MultiGrid-BlockedUnlockPickup-v0", 8, dict(max_steps=60, joint_reward=True)),
    "crowd6_lh": ("MultiGrid-LockedHallway-2Rooms-v0", 6, dict(max_steps=40)),
    # more rooms than door colours: doors of one colour are separate doors (locked_hallway.py:150-151)
    "lh8": ("MultiGrid-LockedHallway-6Rooms-v0", 5, dict(num_rooms=8, room_size=5, max_steps=60)),
    "crowd16_lh": ("MultiGrid-LockedHallway-4Rooms-v0", 16, dict(max_steps=40, allow_agent_overlap=False)),
    # hook-free environments finished inside the lane-per-agent pass: joint rewards, per-agent and joint termination
    "joint8_empty": ("MultiGrid-Empty-5x5-v0", 8, dict(max_steps=30, joint_reward=True, success_termination_mode="all")),
    "joint7_empty_any": ("MultiGrid-Empty-Random-5x5-v0", 7, dict(max_steps=25, joint_reward=True)),
    "solo6_empty": ("MultiGrid-Empty-Random-6x6-v0", 6, dict(max_steps=25, success_termination_mode="all", failure_termination_mode="any")),
}


BASELINE = {
    "cfg2": ("MultiGrid-RedBlueDoors-8x8-v0", 2, {}),
    "cfg3": ("MultiGrid-BlockedUnlockPickup-v0", 2, {}),
    "cfg4": ("MultiGrid-LockedHallway-4Rooms-v0", 4, {}),
    "cfg5": ("MultiGrid-Empty-16x16-v0", 8, {}),
}


def make(case, num_envs, n_layouts=64, table=None, **kw):
    from multigrid_amd.batched import BatchedMultiGrid
    from oracle.coracle import BatchOracle

    name, agents, extra = (table or CASES)[case]
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(dict(agent_view_size=7), **extra))
    pool = layouts.make_pool(gen, n_layouts, seed0=7)
    env = BatchedMultiGrid(spec, num_envs, device=0, **kw)
    env.upload_layouts(pool[0], pool[1], pool[2])
    return spec, pool, env, BatchOracle(spec, num_envs)


def compare(out, ora_out, where, image=True):
    import torch
    torch.cuda.synchronize()
    img, direction, rew, term, trunc = ora_out
    if image:
        np.testing.assert_array_equal(out.image.cpu().numpy(), img, err_msg=where + " image")
        np.testing.assert_array_equal(out.direction.cpu().numpy(), direction, err_msg=where + " direction")
    np.testing.assert_allclose(out.reward.cpu().numpy(), rew, rtol=0, atol=1e-6, err_msg=where + " reward")
    np.testing.assert_array_equal(out.terminated.cpu().numpy(), term, err_msg=where + " terminated")
    np.testing.assert_array_equal(out.truncated.cpu().numpy(), trunc, err_msg=where + " truncated")


def compare_info(out, info, where):
    """CompetativeRedBlueDoor: the step's info dict as a byte per agent (bit 0 door_open_done, bit 1 got_eliminated_done,
    bit 2 eliminated_opponents_done, bits 4-7 eliminated_opponent_num)."""
    np.testing.assert_array_equal(out.info.cpu().numpy(), info, err_msg=where + " info")


def compare_state(env, ora, where):
    st = env.get_state()
    np.testing.assert_array_equal(st["grid"].cpu().numpy(), ora.grid, err_msg=where + " grid")
    np.testing.assert_array_equal(st["agents"].cpu().numpy(), ora.agents, err_msg=where + " agents")
    np.testing.assert_array_equal(st["steps"].cpu().numpy(), ora.steps, err_msg=where + " steps")
    np.testing.assert_array_equal(st["flags"].cpu().numpy(), ora.flags, err_msg=where + " flags")
    np.testing.assert_array_equal(st["meta"].cpu().numpy(), ora.meta, err_msg=where + " meta")


def rand_inputs(rng, e, a):
    actions = rng.choice(7, size=(e, a), p=BIASED).astype(np.uint8)
    order = np.argsort(rng.random((e, a)), axis=1).astype(np.uint8)
    return actions, order


@pytest.mark.parametrize("case", list(CASES))
def test_manual_reset_loop(case):
    """Reference usage: step until done, then reset() that environment (here: masked mg_reset)."""
    e = 3000 if case != "single" else 1500
    spec, pool, env, ora = make(case, e)
    a = spec["num_agents"]
    rng = np.random.default_rng(1)
    idx = rng.integers(0, len(pool[0]), size=e)
    out = env.reset(layout_idx=idx)
    ora.load(np.arange(e), pool[0][idx], pool[1][idx], pool[2][idx])
    np.testing.assert_array_equal(out.image.cpu().numpy(), ora.gen_obs(), err_msg="reset image")
    for t in range(70):
        actions, order = rand_inputs(rng, e, a)
        out = env.step(actions, order)
        res = ora.step(actions, order)
        compare(out, res, f"{case} step {t}")
        if t % 10 == 9:
            compare_state(env, ora, f"{case} step {t}")
        done = res[3].all(axis=1) | res[4].all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            new = rng.integers(0, len(pool[0]), size=e)
            ora.load(ids, pool[0][new[ids]], pool[1][new[ids]], pool[2][new[ids]])
            expect_img = out.image.cpu().numpy().copy()
            expect_img[ids] = ora.gen_obs()[ids]
            out = env.reset(layout_idx=new, mask=done.astype(np.uint8))
            # only the reset environments' observations may change
            np.testing.assert_array_equal(out.image.cpu().numpy(), expect_img, err_msg=f"{case} masked reset {t}")
    compare_state(env, ora, f"{case} final")
    env.close()


def expected_layout(env_ids, episode, n_layouts, offset=0):
    return (offset + env_ids + LAYOUT_STRIDE * episode) % n_layouts


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("case", ["cfg2", "cfg3", "cfg4", "cfg5", "odd3", "big10", "joint8_empty", "solo6_empty"])
def test_autoreset_modes(case, mode):
    e, offset = 2500, 1000
    spec, pool, env, ora = make(case, e, autoreset=mode, env_id_offset=offset)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(2)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    out = env.reset()
    idx = expected_layout(ids_all, episode, n_l, offset)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    np.testing.assert_array_equal(out.image.cpu().numpy(), ora.gen_obs())
    n_done = n_term = n_trunc = 0
    len_sum = 0
    rew_sum = 0.0
    for t in range(90):
        actions, order = rand_inputs(rng, e, a)
        out = env.step(actions, order)
        img, direction, rew, term, trunc = ora.step(actions, order)
        img, direction = img.copy(), direction.copy()
        done = term.all(axis=1) | trunc.all(axis=1)
        rew_sum += np.floor(rew.sum(axis=1) * 2 ** 24).sum() / 2 ** 24
        if done.any():
            ids = np.nonzero(done)[0]
            n_done += len(ids)
            n_term += int(term.all(axis=1)[ids].sum())
            n_trunc += int((~term.all(axis=1))[ids].sum())
            len_sum += int(ora.steps[ids].sum())
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], n_l, offset)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
            if mode == 1:  # the caller sees the next episode's first observation
                fresh = ora.gen_obs()
                img[ids] = fresh[ids]
                direction[ids] = ora.agents[ids, :, 2]
        compare(out, (img, direction, rew, term, trunc), f"{case} mode {mode} step {t}")
        if t % 15 == 14:
            compare_state(env, ora, f"{case} mode {mode} step {t}")
    assert n_done > 0
    st = env.stats().cpu().numpy()
    assert st[0] == n_done and st[1] == len_sum and st[2] == n_term and st[3] == n_trunc and st[5] == 90
    assert abs(st[4] / 2 ** 24 - rew_sum) < 1e-3 * max(1.0, rew_sum)
    env.close()


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("rooms,room_size,agents,joint", [(8, 4, 4, True), (10, 5, 3, False), (14, 5, 6, True)])
def test_lockedhallway_with_more_rooms_than_door_colours(rooms, room_size, agents, joint, mode):
    """Beyond six rooms door colours repeat (locked_hallway.py:150-151): doors of one colour are separate doors that
    each pay once, and the all-doors ending is reached at the number of DISTINCT colours (:160-166, :213).  Half of
    the environments start with every door unlocked and the agents toggle a lot, so that doors pay all the time."""
    from multigrid_amd.batched import BatchedMultiGrid
    from oracle.coracle import BatchOracle
    e, n_l = 2000, 48
    spec, gen = envs.make_spec("MultiGrid-LockedHallway-6Rooms-v0", num_agents=agents, num_rooms=rooms, room_size=room_size,
                               max_steps=45, joint_reward=joint, agent_view_size=5)  # (rooms of 2 x 2 cells cannot hold the keys
    # of three doors of one colour: the reference's place_obj would then never return, locked_hallway.py:176-181)
    pool = list(layouts.make_pool(gen, n_l, seed0=3)[:3])
    assert (pool[2][:, 3] >= 4).all() and (pool[2][:, 3] <= 6).all()
    doors = pool[0][..., 0] == 4
    pool[0][::2][doors[::2], 2] = 1  # closed instead of locked: a toggle opens them
    env = BatchedMultiGrid(spec, e, device=0, autoreset=mode)
    env.upload_layouts(pool[0], pool[1], pool[2])
    ora = BatchOracle(spec, e)
    rng = np.random.default_rng(rooms)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    env.reset()
    idx = expected_layout(ids_all, episode, n_l)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    paid = ended = 0
    for t in range(100):
        actions = rng.choice(7, size=(e, agents), p=[0.14, 0.14, 0.3, 0.02, 0.02, 0.38, 0.0]).astype(np.uint8)
        order = np.argsort(rng.random((e, agents)), axis=1).astype(np.uint8)
        out = env.step(actions, order)
        img, direction, rew, term, trunc = ora.step(actions, order)
        img, direction = img.copy(), direction.copy()
        paid += int((rew > 0).any(axis=1).sum())
        ended += int((term.all(axis=1) & ~trunc.all(axis=1)).sum())
        done = term.all(axis=1) | trunc.all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            if mode == 0:
                compare(out, (img, direction, rew, term, trunc), f"{rooms} rooms step {t}")
                new = rng.integers(0, n_l, size=e)
                ora.load(ids, pool[0][new[ids]], pool[1][new[ids]], pool[2][new[ids]])
                env.reset(layout_idx=new, mask=done.astype(np.uint8))
                continue
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], n_l)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
            img[ids] = ora.gen_obs()[ids]
            direction[ids] = ora.agents[ids, :, 2]
        compare(out, (img, direction, rew, term, trunc), f"{rooms} rooms mode {mode} step {t}")
        if t % 10 == 9:
            compare_state(env, ora, f"{rooms} rooms mode {mode} step {t}")  # (meta[0..1]: the doors paid so far)
    compare_state(env, ora, f"{rooms} rooms mode {mode}")
    assert paid > 200 and (ended > 0 or rooms > 8), (paid, ended)  # (the ending needs 4 to 6 doors within 45 steps)
    env.close()


def fmix32(h):
    h = h ^ (h >> np.uint32(16))
    h = h * np.uint32(0x85EBCA6B)
    h = h ^ (h >> np.uint32(13))
    h = h * np.uint32(0xC2B2AE35)
    return h ^ (h >> np.uint32(16))


def device_order(seed, gid, episode, step, a):
    """The documented on-device order: Fisher-Yates over 0..a-1, for k = a-1 .. 1 swapping entries k and
    j = (field * (k + 1)) >> 16 with 16-bit fields taken low half first from fmix32(key + (q + 1) * 0x9E3779B9),
    the key mixing (seed, global environment index, episode, step) through the murmur3 finalizer."""
    with np.errstate(over="ignore"):
        gid, episode, step = (np.asarray(x).astype(np.uint32) for x in (gid, episode, step))
        h = fmix32(np.uint32(seed & 0xFFFFFFFF) ^ gid)
        h = fmix32(h ^ np.uint32(seed >> 32) ^ episode)
        key = fmix32(h ^ step)
        order = np.tile(np.arange(a, dtype=np.uint8), (len(gid), 1))
        rows = np.arange(len(gid))
        word = None
        for f, k in enumerate(range(a - 1, 0, -1)):
            if f % 2 == 0:
                word = fmix32(key + np.uint32(f // 2 + 1) * np.uint32(0x9E3779B9))
            field = (word >> np.uint32(16 * (f % 2))) & np.uint32(0xFFFF)
            j = ((field * np.uint32(k + 1)) >> np.uint32(16)).astype(np.int64)
            tmp = order[rows, k].copy()
            order[rows, k] = order[rows, j]
            order[rows, j] = tmp
    return order


@pytest.mark.parametrize("case", ["cfg4", "cfg5_nooverlap", "big12"])
def test_on_device_agent_order(case):
    e, offset, seed = 2048, 77, 99
    spec, pool, env, ora = make(case, e, autoreset=1, env_id_offset=offset, seed=seed)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(3)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    env.reset()
    idx = expected_layout(ids_all, episode, n_l, offset)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    seen = set()
    for t in range(80):
        actions, _ = rand_inputs(rng, e, a)
        order = device_order(seed, offset + ids_all, episode, ora.steps + 1, a)
        seen.update(map(bytes, order[:64]))
        out = env.step(actions, None)
        img, direction, rew, term, trunc = ora.step(actions, order)
        img, direction = img.copy(), direction.copy()
        done = term.all(axis=1) | trunc.all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], n_l, offset)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
            img[ids] = ora.gen_obs()[ids]
            direction[ids] = ora.agents[ids, :, 2]
        compare(out, (img, direction, rew, term, trunc), f"{case} device order step {t}")
    assert len(seen) > 20  # the permutations do vary
    env.close()


def test_gen_obs_entry_point():
    spec, pool, env, ora = make("cfg4", 777)
    idx = np.arange(777) % len(pool[0])
    env.reset(layout_idx=idx)
    ora.load(np.arange(777), pool[0][idx], pool[1][idx], pool[2][idx])
    rng = np.random.default_rng(4)
    for t in range(5):
        actions, order = rand_inputs(rng, 777, spec["num_agents"])
        env.step(actions, order)
        ora.step(actions, order)
    import torch
    from multigrid_amd.batched import StepOutput
    o = StepOutput(image=torch.full_like(env.out.image, 255), direction=torch.full_like(env.out.direction, 255),
                   reward=env.out.reward, terminated=env.out.terminated, truncated=env.out.truncated)
    env.gen_obs(out=o)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(o.image.cpu().numpy(), ora.gen_obs())
    np.testing.assert_array_equal(o.direction.cpu().numpy(), ora.agents[:, :, 2])
    env.close()


def test_set_state_roundtrip_and_errors():
    from multigrid_amd._lib import MgError
    spec, pool, env, ora = make("cfg3", 500)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, len(pool[0]), 500)
    env.set_state(grid=pool[0][idx], agents=pool[1][idx], meta=pool[2][idx],
                  steps=np.full(500, 7, np.int32), flags=np.zeros(500, np.uint8))
    ora.load(np.arange(500), pool[0][idx], pool[1][idx], pool[2][idx])
    ora.steps[:] = 7
    compare_state(env, ora, "after set_state")
    actions, order = rand_inputs(rng, 500, 2)
    compare(env.step(actions, order), ora.step(actions, order), "step after set_state")
    with pytest.raises(ValueError):
        env.step(np.zeros((3, 2), np.uint8))
    bad = pool[0][:4].copy()
    bad[0, 0, 0, 0] = 99
    with pytest.raises(MgError):
        env.upload_layouts(bad, pool[1][:4], pool[2][:4])
    # a state the packed cells cannot hold is refused as a whole, not truncated
    bad_grid = pool[0][idx].copy()
    bad_grid[3, 2, 2, 1] = 16
    with pytest.raises(MgError, match="nothing was changed"):
        env.set_state(grid=bad_grid)
    bad_agents = st_agents = env.get_state()["agents"].cpu().numpy().copy()
    bad_agents[7, 1, 5] = 200
    with pytest.raises(MgError):
        env.set_state(agents=bad_agents)
    compare_state(env, ora, "after refused set_state")
    env.close()


def test_redbluedoors_layouts_need_door_positions():
    from multigrid_amd._lib import MgError
    spec, pool, env, ora = make("cfg2", 8)
    with pytest.raises(ValueError):
        env.upload_layouts(pool[0], pool[1])
    meta = pool[2].copy()
    meta[5] = 0  # the corner wall is neither door
    with pytest.raises(MgError, match="red door"):
        env.upload_layouts(pool[0], pool[1], meta)
    env.close()


@pytest.mark.parametrize("case,e", [("cfg2", 4096), ("cfg3", 65536), ("cfg4", 65536), ("cfg5", 131072)])
def test_baseline_sizes(case, e):
    """Full BASELINE batch sizes (per GPU), auto-reset on, checked against the C oracle."""
    offset = 0
    spec, pool, env, ora = make(case, e, table=BASELINE, autoreset=1)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(6)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    env.reset()
    idx = expected_layout(ids_all, episode, n_l, offset)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    # start episodes at scattered step counts so that truncations and resets occur immediately
    steps0 = rng.integers(0, spec["max_steps"], size=e).astype(np.int32)
    env.set_state(steps=steps0)
    ora.steps[:] = steps0
    for t in range(12):
        actions, order = rand_inputs(rng, e, a)
        out = env.step(actions, order)
        img, direction, rew, term, trunc = ora.step(actions, order)
        img, direction = img.copy(), direction.copy()
        done = term.all(axis=1) | trunc.all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], n_l, offset)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
            img[ids] = ora.gen_obs()[ids]
            direction[ids] = ora.agents[ids, :, 2]
        compare(out, (img, direction, rew, term, trunc), f"{case} E={e} step {t}")
    compare_state(env, ora, f"{case} E={e} final")
    env.close()


@pytest.mark.parametrize("case,tile", [("cfg5", "2,3"), ("wide2", "32,1"), ("wide2", "20,2"), ("cfg4", "16,4"),
                                       ("single", "64,1"), ("odd3", "7,2")])
def test_explicit_tile_shapes(case, tile, monkeypatch):
    """Tile shapes the size heuristic does not pick for small batches, among them sub-tiles whose grids
    exceed the 8 KiB the prefetch registers hold (wide2)."""
    table = dict(CASES, wide2=("MultiGrid-Empty-16x16-v0", 2, dict(max_steps=40)))
    monkeypatch.setenv("MGRID_TUNING", "1")
    monkeypatch.setenv("MGRID_TILE", tile)
    e = 1111
    spec, pool, env, ora = make(case, e, table=table, autoreset=2)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(5)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    out = env.reset()
    idx = expected_layout(ids_all, episode, n_l)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    np.testing.assert_array_equal(out.image.cpu().numpy(), ora.gen_obs())
    for t in range(60):
        actions, order = rand_inputs(rng, e, a)
        out = env.step(actions, order)
        res = ora.step(actions, order)
        compare(out, res, f"{case} tile {tile} step {t}")
        done = res[3].all(axis=1) | res[4].all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], n_l)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
    compare_state(env, ora, f"{case} tile {tile} final")
    env.close()


@pytest.mark.parametrize("stage", ["16", "32", "64"])
@pytest.mark.parametrize("case", ["cfg2", "cfg3", "cfg4", "cfg5", "odd3", "single"])
def test_staging_widths(case, stage, monkeypatch):
    """The step kernel's variants that the library otherwise picks by batch size: images staged 16, 32 or 64 agents at
    a time (MGRID_STAGE).  Same contract as everywhere: bit-exact against the C oracle, next-step auto-reset."""
    lanes = None
    monkeypatch.setenv("MGRID_TUNING", "1")
    monkeypatch.setenv("MGRID_STAGE", stage)
    e = 777
    spec, pool, env, ora = make(case, e, autoreset=2)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(int(stage) + 3)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    out = env.reset()
    idx = expected_layout(ids_all, episode, n_l)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    np.testing.assert_array_equal(out.image.cpu().numpy(), ora.gen_obs())
    for t in range(70):
        actions, order = rand_inputs(rng, e, a)
        out = env.step(actions, order)
        res = ora.step(actions, order)
        compare(out, res, f"{case} stage {stage} lanes {lanes} step {t}")
        done = res[3].all(axis=1) | res[4].all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], n_l)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
    compare_state(env, ora, f"{case} stage {stage} lanes {lanes} final")
    env.close()


@pytest.mark.parametrize("case", ["crowd16_lh", "crowd8_rbd", "crowd8_bup", "crowd12_bup", "cfg5_nooverlap", "crowd8_empty"])
def test_crowded_rooms_at_scale(case):
    """Many environments in which agents block each other, reach for the same objects and walk through doors
    that others open in the same step: the cases in which only some agents of an environment are replayed in order."""
    table = dict(CASES, crowd12_bup=("MultiGrid-BlockedUnlockPickup-v0", 12, dict(max_steps=60, allow_agent_overlap=False)))
    e = 20000
    spec, pool, env, ora = make(case, e, table=table, n_layouts=128, autoreset=2)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(9)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    env.reset()
    idx = expected_layout(ids_all, episode, n_l)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    for t in range(45):
        actions, order = rand_inputs(rng, e, a)
        out = env.step(actions, order)
        res = ora.step(actions, order)
        compare(out, res, f"{case} step {t}", image=(t % 5 == 4))
        done = res[3].all(axis=1) | res[4].all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], n_l)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
    compare_state(env, ora, f"{case} final")
    env.close()


@pytest.mark.parametrize("case", ["cfg4", "cfg5_nooverlap"])
def test_two_shards_equal_one_batch(case):
    """Sharding as bench.py --gpus N does it: two handles holding the halves of a batch (env_id_offset 0 and E / 2)
    step exactly like one handle holding all of it - layout choice and the on-device agent order are functions of
    the GLOBAL environment index - and their statistics add up."""
    import torch
    e, seed = 3000, 5
    spec, pool, whole, _ = make(case, e, autoreset=1, seed=seed)
    _, _, lo, _ = make(case, e // 2, autoreset=1, seed=seed, env_id_offset=0)
    _, _, hi, _ = make(case, e // 2, autoreset=1, seed=seed, env_id_offset=e // 2)
    a = spec["num_agents"]
    outs = [x.reset() for x in (whole, lo, hi)]
    torch.cuda.synchronize()
    assert torch.equal(outs[0].image, torch.cat([outs[1].image, outs[2].image]))
    g = torch.Generator(device="cuda:0").manual_seed(3)
    steps0 = torch.randint(0, spec["max_steps"], (e,), generator=g, device="cuda:0", dtype=torch.int32)
    whole.set_state(steps=steps0)
    lo.set_state(steps=steps0[:e // 2])
    hi.set_state(steps=steps0[e // 2:])
    for t in range(70):
        acts = torch.randint(0, 7, (e, a), generator=g, device="cuda:0", dtype=torch.uint8)
        w, l, h = whole.step(acts), lo.step(acts[:e // 2].contiguous()), hi.step(acts[e // 2:].contiguous())
        for f in ("image", "direction", "reward", "terminated", "truncated"):
            assert torch.equal(getattr(w, f), torch.cat([getattr(l, f), getattr(h, f)])), f"{case} step {t} {f}"
    sw, sl, sh = (x.stats().cpu().numpy() for x in (whole, lo, hi))
    assert sw[0] > 0 and (sw[:5] == sl[:5] + sh[:5]).all() and sw[5] == sl[5] == sh[5] == 70
    for x in (whole, lo, hi):
        x.close()


@pytest.mark.parametrize("case", ["cfg3", "cfg4", "joint8_empty", "big10", "odd5"])
def test_agent_rows_and_double_rewards_from_the_step_launch(case):
    """MgOut.rows_dev / reward64_dev: the agent rows of the state a step (reset, gen_obs) leaves and the rewards as
    doubles, written by the same launch - rows against mg_get_state and the oracle's agent array, rewards EQUAL to the
    oracle's doubles (multigrid/base.py:655-659 in the reference's order of operations), not merely close."""
    import torch
    from multigrid_amd.batched import StepOutput
    e = 2100
    spec, pool, env, ora = make(case, e)
    a = spec["num_agents"]
    out = StepOutput(image=env.out.image, direction=env.out.direction, reward=env.out.reward, terminated=env.out.terminated,
                     truncated=env.out.truncated, rows=torch.full((e, a, 8), 77, dtype=torch.uint8, device="cuda:0"),
                     reward64=torch.full((e, a), -5.0, dtype=torch.float64, device="cuda:0"))
    rng = np.random.default_rng(2)
    idx = rng.integers(0, len(pool[0]), size=e)
    env.reset(layout_idx=idx, out=out)
    ora.load(np.arange(e), pool[0][idx], pool[1][idx], pool[2][idx])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.rows.cpu().numpy(), ora.agents, err_msg="rows after reset")
    assert (out.reward64 == -5.0).all()  # a reset writes no rewards
    paid = 0
    for t in range(60):
        actions, order = rand_inputs(rng, e, a)
        env.step(actions, order, out=out)
        res = ora.step(actions, order)
        compare(out, res, f"{case} step {t}")
        np.testing.assert_array_equal(out.rows.cpu().numpy(), ora.agents, err_msg=f"{case} rows step {t}")
        np.testing.assert_array_equal(out.reward64.cpu().numpy(), res[2], err_msg=f"{case} double rewards step {t}")
        paid += int((res[2] != 0).sum())
    assert paid > 0 or case != "joint8_empty"  # (the small Empty rooms pay often)
    out.rows.fill_(0)
    env.gen_obs(out=out)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.rows.cpu().numpy(), env.get_state()["agents"].cpu().numpy())
    # with these outputs mg_step_n takes one launch per step and still fills every slice
    T = 3
    acts = rng.integers(0, 7, size=(T, e, a)).astype(np.uint8)
    many = env.step_n(acts, out=StepOutput(
        image=None, direction=None, reward=torch.zeros((T, e, a), dtype=torch.float32, device="cuda:0"),
        terminated=None, truncated=None, rows=torch.zeros((T, e, a, 8), dtype=torch.uint8, device="cuda:0"),
        reward64=torch.zeros((T, e, a), dtype=torch.float64, device="cuda:0")))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(many.rows[-1].cpu().numpy(), env.get_state()["agents"].cpu().numpy())
    np.testing.assert_allclose(many.reward64.cpu().numpy(), many.reward.cpu().numpy().astype(np.float64), rtol=0, atol=1e-6)
    env.close()
