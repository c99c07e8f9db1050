"""Sizes at the edges of what the library accepts: one environment, 64 agents, grids of several thousand cells,
every view size, batch sizes that leave partial tiles."""
import numpy as np
import pytest

from multigrid_amd import envs, layouts

pytestmark = pytest.mark.gpu

EXTREMES = {
    "one_env": ("MultiGrid-LockedHallway-4Rooms-v0", 4, dict(max_steps=30), 1),
    "one_env_single_agent": ("MultiGrid-Empty-8x8-v0", 1, dict(max_steps=20), 1),
    "agents64": ("MultiGrid-Empty-16x16-v0", 64, dict(max_steps=40, allow_agent_overlap=False, success_termination_mode="all"), 130),
    "agents33": ("MultiGrid-Empty-16x16-v0", 33, dict(max_steps=40), 257),
    "agents17_bup": ("MultiGrid-BlockedUnlockPickup-v0", 17, dict(max_steps=40, room_size=8), 300),
    "grid60_view7": ("MultiGrid-Empty-Random-6x6-v0", 3, dict(size=60, max_steps=40), 150),
    "grid63_view3": ("MultiGrid-Empty-Random-6x6-v0", 5, dict(size=63, max_steps=40, agent_view_size=3), 97),
    "grid40_view5_walls": ("MultiGrid-RedBlueDoors-8x8-v0", 6, dict(size=40, max_steps=40, agent_view_size=5), 211),
    "rooms6_big": ("MultiGrid-LockedHallway-6Rooms-v0", 9, dict(room_size=9, max_steps=40), 180),
}


@pytest.mark.parametrize("case", list(EXTREMES))
@pytest.mark.parametrize("mode", [0, 1])
def test_extreme_sizes(case, mode):
    from multigrid_amd.batched import BatchedMultiGrid
    from oracle.coracle import BatchOracle
    from test_gpu_batch import compare, compare_state, expected_layout, rand_inputs

    name, agents, kw, e = EXTREMES[case]
    kw = dict(kw)
    view = kw.pop("agent_view_size", 7)
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=view, **kw)
    pool = layouts.make_pool(gen, 16, seed0=3)
    env = BatchedMultiGrid(spec, e, device=0, autoreset=mode)
    env.upload_layouts(pool[0], pool[1], pool[2])
    ora = BatchOracle(spec, e)
    rng = np.random.default_rng(4)
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    out = env.reset()
    idx = expected_layout(ids_all, episode, len(pool[0]))
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    np.testing.assert_array_equal(out.image.cpu().numpy(), ora.gen_obs(), err_msg=f"{case} reset image")
    for t in range(50):
        actions, order = rand_inputs(rng, e, agents)
        out = env.step(actions, order)
        img, direction, rew, term, trunc = ora.step(actions, order)
        img, direction = img.copy(), direction.copy()
        done = term.all(axis=1) | trunc.all(axis=1)
        if mode == 1 and done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], len(pool[0]))
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
            img[ids] = ora.gen_obs()[ids]
            direction[ids] = ora.agents[ids, :, 2]
        compare(out, (img, direction, rew, term, trunc), f"{case} mode {mode} step {t}")
    compare_state(env, ora, f"{case} final")
    env.close()


def test_limits_are_reported():
    from multigrid_amd._lib import MgError
    from multigrid_amd.batched import BatchedMultiGrid
    spec, _ = envs.make_spec("MultiGrid-Empty-8x8-v0", num_agents=2, size=70, agent_view_size=7)  # 4900 cells
    with pytest.raises(MgError, match="not supported"):
        BatchedMultiGrid(spec, 4, device=0)
    spec, _ = envs.make_spec("MultiGrid-Empty-8x8-v0", num_agents=65, agent_view_size=7)
    with pytest.raises(MgError, match="num_agents"):
        BatchedMultiGrid(spec, 4, device=0)


@pytest.mark.parametrize("view", [9, 11])
def test_wide_views_through_the_other_entry_points(view):
    """9 x 9 and 11 x 11 views through the entry points the parity replays do not use: a caller-held state
    (mg_gen_obs_of_state), two stream shards (mg_step_shards), several steps per call (mg_step_n, queued as single
    launches for these views) and the one-hot output - each against the C oracle."""
    import torch
    from multigrid_amd.batched import BatchedMultiGrid, gen_obs_of_state
    from multigrid_amd.sharded import StreamShardedMultiGrid
    from oracle import pyref
    from oracle.coracle import BatchOracle
    e, agents = 700, 3
    spec, gen = envs.make_spec("MultiGrid-LockedHallway-6Rooms-v0", num_agents=agents, num_rooms=8, room_size=4,
                               max_steps=30, agent_view_size=view)
    pool = layouts.make_pool(gen, 32, seed0=view)
    idx = np.arange(e) % 32
    ora = BatchOracle(spec, e)
    ora.load(np.arange(e), pool[0][idx], pool[1][idx], pool[2][idx])
    dev = torch.device("cuda:0")
    # a state the caller holds
    img = torch.zeros((e, agents, view, view, 3), dtype=torch.uint8, device=dev)
    oh = torch.zeros((e, agents, view, view, 24), dtype=torch.uint8, device=dev)
    gen_obs_of_state(spec, torch.as_tensor(pool[0][idx]).to(dev).contiguous(), torch.as_tensor(pool[1][idx]).to(dev).contiguous(),
                     image=img, onehot=oh)
    torch.cuda.synchronize()
    want = ora.gen_obs()
    np.testing.assert_array_equal(img.cpu().numpy(), want)
    np.testing.assert_array_equal(oh.cpu().numpy(), pyref.one_hot_batch(want))
    # two stream shards, then several steps per call on one handle
    sh = StreamShardedMultiGrid(spec, e, shards=2, device=0)
    sh.upload_layouts(pool[0], pool[1], pool[2])
    sh.reset(layout_idx=idx)
    one = BatchedMultiGrid(spec, e, device=0)
    one.upload_layouts(pool[0], pool[1], pool[2])
    one.reset(layout_idx=idx)
    rng = np.random.default_rng(view)
    acts = rng.choice(7, size=(6, e, agents), p=[0.15, 0.15, 0.3, 0.1, 0.05, 0.25, 0.0]).astype(np.uint8)
    order = np.argsort(rng.random((6, e, agents)), axis=2).astype(np.uint8)
    outn = one.step_n(acts, order)
    torch.cuda.synchronize()
    for t in range(6):
        out = sh.step(acts[t], order[t])
        img_t, direction, rew, term, trunc = ora.step(acts[t], order[t])
        torch.cuda.synchronize()
        for got in (out.image, outn.image[t]):
            np.testing.assert_array_equal(got.cpu().numpy(), img_t, err_msg=f"view {view} step {t}")
        np.testing.assert_allclose(outn.reward[t].cpu().numpy(), rew, rtol=0, atol=1e-6)
        np.testing.assert_array_equal(out.terminated.cpu().numpy(), term)
    sh.close()
    one.close()
