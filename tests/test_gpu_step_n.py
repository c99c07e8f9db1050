"""mg_step_n as ONE launch for T steps (header blocks and grids stay in LDS between the steps): every slice of
every output and the final state against the C oracle, for all three auto-reset modes."""
import numpy as np
import pytest

from test_gpu_batch import CASES, compare_state, device_order, expected_layout, make, rand_inputs

pytestmark = pytest.mark.gpu


def run_case(case, mode, t_steps, e, calls=1, device_order_seed=None, n_layouts=64):
    import torch
    kw = dict(autoreset=mode)
    if device_order_seed is not None:
        kw.update(seed=device_order_seed, env_id_offset=5)
    offset = kw.get("env_id_offset", 0)
    spec, pool, env, ora = make(case, e, n_layouts=n_layouts, **kw)
    a, n_l = spec["num_agents"], len(pool[0])
    rng = np.random.default_rng(21)
    env.reset()
    ids_all = np.arange(e)
    episode = np.zeros(e, np.int64)
    idx = expected_layout(ids_all, episode, n_l, offset)
    ora.load(ids_all, pool[0][idx], pool[1][idx], pool[2][idx])
    steps0 = rng.integers(0, spec["max_steps"], size=e).astype(np.int32)
    env.set_state(steps=steps0)
    ora.steps[:] = steps0
    for call in range(calls):
        acts = np.stack([rand_inputs(rng, e, a)[0] for _ in range(t_steps)])
        order = np.stack([rand_inputs(rng, e, a)[1] for _ in range(t_steps)])
        out = env.step_n(acts, None if device_order_seed is not None else order)
        torch.cuda.synchronize()
        for t in range(t_steps):
            od = order[t]
            if device_order_seed is not None:
                od = device_order(device_order_seed, offset + ids_all, episode, ora.steps + 1, a)
            img, direction, rew, term, trunc = ora.step(acts[t], od)
            img, direction = img.copy(), direction.copy()
            done = term.all(axis=1) | trunc.all(axis=1)
            if mode != 0 and done.any():
                ids = np.nonzero(done)[0]
                episode[ids] += 1
                idx = expected_layout(ids, episode[ids], n_l, offset)
                ora.load(ids, pool[0][idx], pool[1][idx], pool[2][idx])
                if mode == 1:
                    img[ids] = ora.gen_obs()[ids]
                    direction[ids] = ora.agents[ids, :, 2]
            where = f"{case} mode {mode} T={t_steps} call {call} slice {t}"
            np.testing.assert_array_equal(out.image[t].cpu().numpy(), img, err_msg=where + " image")
            np.testing.assert_array_equal(out.direction[t].cpu().numpy(), direction, err_msg=where + " direction")
            np.testing.assert_allclose(out.reward[t].cpu().numpy(), rew, rtol=0, atol=1e-6, err_msg=where + " reward")
            np.testing.assert_array_equal(out.terminated[t].cpu().numpy(), term, err_msg=where + " terminated")
            np.testing.assert_array_equal(out.truncated[t].cpu().numpy(), trunc, err_msg=where + " truncated")
        compare_state(env, ora, f"{case} mode {mode} T={t_steps} after call {call}")
    st = env.stats().cpu().numpy()
    assert st[5] == calls * t_steps
    env.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("case", ["cfg2", "cfg3", "cfg4", "cfg5", "cfg5_nooverlap", "odd3", "single", "big10", "crowd8_rbd",
                                  "crowd8_bup"])
def test_one_launch_for_nine_steps(case, mode):
    run_case(case, mode, 9, 1500, calls=2)


@pytest.mark.parametrize("case,mode", [("cfg3", 1), ("cfg4", 2), ("cfg5", 1), ("crowd6_lh", 0)])
def test_one_launch_for_sixty_four_steps(case, mode):
    run_case(case, mode, 64, 1100)


@pytest.mark.parametrize("case,mode", [("cfg4", 1), ("cfg5_nooverlap", 2), ("big12", 1)])
def test_multi_step_launch_with_device_order(case, mode):
    run_case(case, mode, 9, 1300, calls=2, device_order_seed=77)


@pytest.mark.parametrize("case,tile", [("cfg4", "8,4"), ("cfg2", "20,2"), ("cfg5", "2,3"), ("odd3", "7,3")])
def test_multi_step_launch_with_several_sub_tiles(case, tile, monkeypatch):
    """Tiles of several sub-tiles keep ALL their grids in LDS in the multi-step kernel."""
    monkeypatch.setenv("MGRID_TUNING", "1")
    monkeypatch.setenv("MGRID_TILE", tile)
    run_case(case, 1, 9, 1111)


def test_step_n_without_the_multi_step_kernel(monkeypatch):
    """The per-step fallback (forced here; also taken for one-hot outputs) gives the same slices."""
    monkeypatch.setenv("MGRID_TUNING", "1")
    monkeypatch.setenv("MGRID_MULTI_STEP", "0")
    run_case("cfg4", 1, 9, 900)


def test_step_n_of_one_step_is_step():
    run_case("cfg3", 2, 1, 700, calls=3)
