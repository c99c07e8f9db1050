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
