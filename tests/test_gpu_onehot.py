"""One-hot observations written by the step / reset / gen_obs launches, and mg_step_n, through the C ABI."""
import numpy as np
import pytest

from golden_util import load_group, load_onehot_fixture
from test_gpu_batch import CASES, compare, compare_state, expected_layout, make, rand_inputs

pytestmark = pytest.mark.gpu


def oh(images):
    from oracle import pyref
    return pyref.one_hot_batch(images)


@pytest.mark.parametrize("name,key", [("chaos_blockedunlockpickup", "v7a"), ("chaos_lockedhallway4", "v7b"),
                                      ("chaos_view5", "v5"), ("chaos_view3", "v3")])
def test_one_hot_on_reference_trajectories(name, key):
    """Replays recorded reference trajectories; every image the reference's OneHotObsWrapper encoded for the
    fixture must come out of the launch with exactly that encoding."""
    import torch
    from multigrid_amd.batched import BatchedMultiGrid

    spec, d = load_group(name)
    fx = load_onehot_fixture()
    want = {im.tobytes(): enc for im, enc in zip(fx[f"images_{key}"], fx[f"images_{key}_onehot"])}
    n = len(d["length"])
    env = BatchedMultiGrid(spec, n, device=0, one_hot=True)
    env.upload_layouts(d["layout_grid"], d["layout_agents"], d["layout_meta"])
    out = env.reset(layout_idx=np.arange(n))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.onehot.cpu().numpy(), oh(d["reset_image"]))
    hits = 0
    for t in range(int(d["length"].max())):
        live = d["length"] > t
        out = env.step(d["actions"][:, t], d["order"][:, t])
        got, img = out.onehot.cpu().numpy(), out.image.cpu().numpy()
        np.testing.assert_array_equal(img[live], d["image"][live, t], err_msg=f"{name} step {t} image")
        np.testing.assert_array_equal(got[live], oh(d["image"][live, t]), err_msg=f"{name} step {t} one-hot")
        for e in np.nonzero(live)[0]:
            for a in range(img.shape[1]):
                ref = want.get(img[e, a].tobytes())
                if ref is not None:
                    np.testing.assert_array_equal(got[e, a], ref)
                    hits += 1
    assert hits >= len(fx[f"images_{key}"])
    env.close()


@pytest.mark.parametrize("case", ["cfg2", "cfg4", "cfg5", "odd3", "odd5", "single", "big12"])
@pytest.mark.parametrize("compact", [True, False])
def test_one_hot_large_batches(case, compact):
    """Multi-tile batches with masked resets (the byte-exact fallback of the image writer) and odd agent counts."""
    from multigrid_amd.batched import BatchedMultiGrid
    from oracle.coracle import BatchOracle
    from multigrid_amd import envs, layouts
    import torch

    name, agents, extra = CASES[case]
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(dict(agent_view_size=7), **extra))
    pool = layouts.make_pool(gen, 32, seed0=11)
    e = 2222
    env = BatchedMultiGrid(spec, e, device=0, one_hot=True, compact_image=compact)
    env.upload_layouts(pool[0], pool[1], pool[2])
    ora = BatchOracle(spec, e)
    rng = np.random.default_rng(8)
    idx = rng.integers(0, len(pool[0]), size=e)
    out = env.reset(layout_idx=idx)
    ora.load(np.arange(e), pool[0][idx], pool[1][idx], pool[2][idx])
    torch.cuda.synchronize()
    assert (out.image is None) == (not compact)
    np.testing.assert_array_equal(out.onehot.cpu().numpy(), oh(ora.gen_obs()), err_msg="reset one-hot")
    for t in range(40):
        actions, order = rand_inputs(rng, e, agents)
        out = env.step(actions, order)
        res = ora.step(actions, order)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), oh(res[0]), err_msg=f"{case} step {t} one-hot")
        if compact:
            compare(out, res, f"{case} step {t}")
        done = res[3].all(axis=1) | res[4].all(axis=1)
        if t % 7 == 3:
            done[rng.integers(0, e, size=50)] = True  # scattered partial resets
        if done.any():
            ids = np.nonzero(done)[0]
            new = rng.integers(0, len(pool[0]), size=e)
            ora.load(ids, pool[0][new[ids]], pool[1][new[ids]], pool[2][new[ids]])
            expect = out.onehot.cpu().numpy().copy()
            expect[ids] = oh(ora.gen_obs()[ids])
            out = env.reset(layout_idx=new, mask=done.astype(np.uint8))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(out.onehot.cpu().numpy(), expect, err_msg=f"{case} masked reset {t}")
    o = env.gen_obs()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(o.onehot.cpu().numpy(), oh(ora.gen_obs()), err_msg="gen_obs one-hot")
    compare_state(env, ora, f"{case} final")
    env.close()


def test_one_hot_pointer_alignment_is_checked():
    import ctypes
    from multigrid_amd import _lib
    spec, pool, env, ora = make("cfg2", 64)
    env.reset()
    mo = _lib.MgOut(None, None, None, None, None, 4)
    rc = _lib.lib().mg_gen_obs(env._h, ctypes.byref(mo), None)
    assert rc == -1 and b"8-byte" in _lib.lib().mg_last_error(env._h)
    env.close()


@pytest.mark.parametrize("case,mode", [("cfg3", 0), ("cfg5", 1), ("cfg4", 2)])
def test_step_n_equals_steps(case, mode):
    """mg_step_n queues T launches; the slices must equal T single steps (checked against the oracle)."""
    import torch
    e, t_steps = 1500, 9
    spec, pool, env, ora = make(case, e, autoreset=mode)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(12)
    env.reset()
    episode = np.zeros(e, np.int64)
    idx = expected_layout(np.arange(e), episode, n_l)
    ora.load(np.arange(e), pool[0][idx], pool[1][idx], pool[2][idx])
    steps0 = rng.integers(0, spec["max_steps"], size=e).astype(np.int32)
    env.set_state(steps=steps0)
    ora.steps[:] = steps0
    acts = np.stack([rand_inputs(rng, e, a)[0] for _ in range(t_steps)])
    order = np.stack([rand_inputs(rng, e, a)[1] for _ in range(t_steps)])
    out = env.step_n(acts, order, one_hot=True)
    torch.cuda.synchronize()
    for t in range(t_steps):
        img, direction, rew, term, trunc = ora.step(acts[t], order[t])
        img, direction = img.copy(), direction.copy()
        done = term.all(axis=1) | trunc.all(axis=1)
        if mode != 0 and done.any():
            ids = np.nonzero(done)[0]
            episode[ids] += 1
            idx = expected_layout(ids, episode[ids], n_l)
            ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
            if mode == 1:
                img[ids] = ora.gen_obs()[ids]
                direction[ids] = ora.agents[ids, :, 2]
        where = f"{case} step_n slice {t}"
        np.testing.assert_array_equal(out.image[t].cpu().numpy(), img, err_msg=where + " image")
        np.testing.assert_array_equal(out.onehot[t].cpu().numpy(), oh(img), err_msg=where + " one-hot")
        np.testing.assert_array_equal(out.direction[t].cpu().numpy(), direction, err_msg=where)
        np.testing.assert_allclose(out.reward[t].cpu().numpy(), rew, rtol=0, atol=1e-6, err_msg=where)
        np.testing.assert_array_equal(out.terminated[t].cpu().numpy(), term, err_msg=where)
        np.testing.assert_array_equal(out.truncated[t].cpu().numpy(), trunc, err_msg=where)
    compare_state(env, ora, f"{case} after step_n")
    env.close()


@pytest.mark.parametrize("stream", ["0", "1"])
@pytest.mark.parametrize("case", ["cfg3", "cfg5", "odd5"])
def test_streaming_stores_either_way(case, stream, monkeypatch):
    """Image and one-hot stores take their non-temporal form above a size; both forms forced here."""
    import torch
    from multigrid_amd import envs, layouts
    from multigrid_amd.batched import BatchedMultiGrid
    from oracle.coracle import BatchOracle

    monkeypatch.setenv("MGRID_TUNING", "1")
    monkeypatch.setenv("MGRID_STREAM", stream)
    name, agents, extra = CASES[case]
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(dict(agent_view_size=7), **extra))
    pool = layouts.make_pool(gen, 32, seed0=13)
    e = 4096
    env = BatchedMultiGrid(spec, e, device=0, one_hot=True, autoreset=2)
    env.upload_layouts(pool[0], pool[1], pool[2])
    ora = BatchOracle(spec, e)
    rng = np.random.default_rng(14)
    idx = rng.integers(0, len(pool[0]), size=e)
    env.reset(layout_idx=idx)
    ora.load(np.arange(e), pool[0][idx], pool[1][idx], pool[2][idx])
    for t in range(12):
        actions, order = rand_inputs(rng, e, agents)
        out = env.step(actions, order)
        res = ora.step(actions, order)
        torch.cuda.synchronize()
        compare(out, res, f"{case} stream={stream} step {t}")
        np.testing.assert_array_equal(out.onehot.cpu().numpy(), oh(res[0]), err_msg=f"{case} stream={stream} step {t} one-hot")
        done = res[3].all(axis=1) | res[4].all(axis=1)
        if done.any():
            break
    env.close()


def test_competitive_v2_observation_on_device():
    """views.competitive_v2_observation against the reference wrapper's recorded output."""
    import os
    import torch
    from golden_util import GOLDEN_DIR
    from multigrid_amd.views import competitive_v2_observation
    with np.load(os.path.join(GOLDEN_DIR, "onehot_v2.npz")) as z:
        d = {k: z[k] for k in z.files}
    for v in (7, 5):
        got = competitive_v2_observation(torch.as_tensor(d[f"images_v{v}"], device="cuda:0"),
                                         torch.as_tensor(d[f"directions_v{v}"], device="cuda:0"))
        np.testing.assert_array_equal(got.cpu().numpy(), d[f"observations_v{v}"])
