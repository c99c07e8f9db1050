"""The SyncVectorEnv-style adapter: same-step auto-reset, final_info episode statistics, against the C oracle."""
import numpy as np
import pytest

from multigrid_amd import envs, layouts

pytestmark = pytest.mark.gpu

LAYOUT_STRIDE = 7919


def device_order_for(seed, gid, episode, step, a):
    from test_gpu_batch import device_order
    return device_order(seed, gid, episode, step, a)


@pytest.mark.parametrize("name,agents,single,kw", [
    ("MultiGrid-BlockedUnlockPickup-v0", 1, True, dict(max_steps=40)),
    ("MultiGrid-LockedHallway-2Rooms-v0", 3, False, dict(max_steps=35)),
    ("MultiGrid-Empty-6x6-v0", 2, False, dict(max_steps=24)),
])
@pytest.mark.parametrize("final_obs", [False, True])
def test_vector_env_contract(name, agents, single, kw, final_obs):
    import torch
    from multigrid_amd.vector import MultiGridVectorEnv
    from oracle.coracle import BatchOracle
    from oracle import pyref

    n, n_layouts, seed = 700, 48, 21
    venv = MultiGridVectorEnv(name, n, agents=agents, device=0, num_layouts=n_layouts, seed=seed,
                              single_agent=single, numpy=True, one_hot=True, agent_view_size=7,
                              final_observation=final_obs, **kw)
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(dict(agent_view_size=7), **kw))
    pool = layouts.make_pool(gen, n_layouts, seed0=seed)
    ora = BatchOracle(spec, n)
    ids_all = np.arange(n)
    episode = np.zeros(n, np.int64)
    obs, infos = venv.reset()
    assert infos == {}
    idx = (ids_all + LAYOUT_STRIDE * episode) % n_layouts
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    squeeze = (lambda x: x[:, 0]) if single else (lambda x: x)
    np.testing.assert_array_equal(obs["image"], squeeze(pyref.one_hot_batch(ora.gen_obs())))
    rng = np.random.default_rng(3)
    ret = np.zeros((n, agents))
    length = np.zeros(n, np.int64)
    finished = 0
    for t in range(90):
        actions = rng.integers(0, 7, size=(n, agents)).astype(np.uint8)
        order = device_order_for(seed, ids_all, episode, ora.steps + 1, agents)
        obs, rew, term, trunc, infos = venv.step(actions[:, 0] if single else actions)
        img, direction, r, tm, tr = ora.step(actions, order)
        img, direction = img.copy(), direction.copy()
        ret += r
        length += 1
        done = tm.all(axis=1) | tr.all(axis=1)
        if done.any():
            ids = np.nonzero(done)[0]
            finished += len(ids)
            assert np.array_equal(infos["_final_info"], done)
            for i in range(n):
                if done[i]:
                    ep = infos["final_info"][i]["episode"]
                    assert ep["l"] == length[i]
                    np.testing.assert_allclose(np.atleast_1d(ep["r"]), ret[i], rtol=0, atol=1e-5)
                else:
                    assert infos["final_info"][i] is None
            np.testing.assert_allclose(infos["episode_return"], squeeze(np.where(done[:, None], ret, 0.0)), atol=1e-5)
            np.testing.assert_array_equal(infos["episode_length"], np.where(done, length, 0))
            if final_obs:  # the observation the finished episode ended on (SyncVectorEnv's info["final_observation"])
                assert np.array_equal(infos["_final_observation"], done)
                last = squeeze(pyref.one_hot_batch(img))
                for i in range(n):
                    if done[i]:
                        np.testing.assert_array_equal(infos["final_observation"][i]["image"], last[i])
                        np.testing.assert_array_equal(infos["final_observation"][i]["direction"], squeeze(direction)[i])
                    else:
                        assert infos["final_observation"][i] is None
            else:
                assert "final_observation" not in infos
            ret[ids] = 0
            length[ids] = 0
            episode[ids] += 1
            idx = (ids + LAYOUT_STRIDE * episode[ids]) % n_layouts
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
            img[ids] = ora.gen_obs()[ids]
            direction[ids] = ora.agents[ids, :, 2]
        else:
            assert infos == {}
        where = f"{name} step {t}"
        np.testing.assert_array_equal(obs["image"], squeeze(pyref.one_hot_batch(img)), err_msg=where)
        np.testing.assert_array_equal(obs["direction"], squeeze(direction), err_msg=where)
        np.testing.assert_allclose(rew, squeeze(r), rtol=0, atol=1e-6, err_msg=where)
        np.testing.assert_array_equal(term, squeeze(tm.astype(bool)), err_msg=where)
        np.testing.assert_array_equal(trunc, squeeze(tr.astype(bool)), err_msg=where)
    assert finished >= n  # every environment was truncated at least once
    assert obs["image"].shape == ((n, 7, 7, 24) if single else (n, agents, 7, 7, 24))
    venv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("e,a", [(1, 1), (1000, 1), (777, 5), (4096, 8), (3, 64), (5, 200), (2, 256)])
def test_episode_stats_launch_matches_the_per_field_arithmetic(e, a):
    """mg_episode_stats against the same bookkeeping written with tensor operations (RecordEpisodeStatistics:
    running return and length, episode over when all agents are terminated or the environment is truncated)."""
    import ctypes

    import torch

    from multigrid_amd import _lib
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(e + a)
    ret = torch.zeros((e, a), dtype=torch.float64, device=dev)
    length = torch.zeros((e,), dtype=torch.int64, device=dev)
    ret_ref, len_ref = ret.clone(), length.clone()
    done = torch.zeros((e,), dtype=torch.uint8, device=dev)
    ep_ret = torch.zeros((e, a), dtype=torch.float64, device=dev)
    ep_len = torch.zeros((e,), dtype=torch.int64, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for t in range(40):
        reward = torch.rand((e, a), generator=g, device=dev, dtype=torch.float32) * (torch.rand((e, a), generator=g, device=dev) < 0.3)
        term = (torch.rand((e, a), generator=g, device=dev) < 0.6).to(torch.uint8)
        trunc = (torch.rand((e, 1), generator=g, device=dev) < 0.05).to(torch.uint8).expand(e, a).contiguous()
        _lib.check(_lib.lib().mg_episode_stats(e, a, p(reward), p(term), p(trunc), p(ret), p(length), p(done), p(ep_ret),
                                               p(ep_len), None))
        ret_ref += reward
        len_ref += 1
        over = term.bool().all(dim=1) | trunc.bool()[:, 0]
        assert torch.equal(done.bool(), over)
        assert torch.equal(ep_ret, torch.where(over[:, None], ret_ref, torch.zeros_like(ret_ref)))
        assert torch.equal(ep_len, torch.where(over, len_ref, torch.zeros_like(len_ref)))
        ret_ref.masked_fill_(over[:, None], 0.0)
        len_ref.masked_fill_(over, 0)
        assert torch.equal(ret, ret_ref) and torch.equal(length, len_ref)
    assert _lib.lib().mg_episode_stats(0, a, None, None, None, None, None, None, None, None, None) == -1
    assert _lib.lib().mg_episode_stats(e, 257, p(reward), p(term), p(trunc), p(ret), p(length), p(done), p(ep_ret),
                                       p(ep_len), None) == -1
