"""One batch held as stream shards (mg_step_shards, multigrid_amd/sharded.py) steps exactly like one handle holding
all of it, and like the C oracle - whether the slices are joined after every step or run ahead of each other."""
import ctypes

import numpy as np
import pytest

from multigrid_amd import envs, layouts
from test_gpu_batch import CASES, compare, compare_state, rand_inputs

pytestmark = pytest.mark.gpu

FIELDS = ("image", "direction", "reward", "terminated", "truncated")


def make_pair(case, e, shards, **kw):
    from multigrid_amd.batched import BatchedMultiGrid
    from multigrid_amd.sharded import StreamShardedMultiGrid
    name, agents, extra = CASES[case]
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(dict(agent_view_size=7), **extra))
    pool = layouts.make_pool(gen, 48, seed0=11)
    whole = BatchedMultiGrid(spec, e, device=0, **kw)
    cut = StreamShardedMultiGrid(spec, e, shards=shards, device=0, **kw)
    for x in (whole, cut):
        x.upload_layouts(pool[0], pool[1], pool[2])
    return spec, pool, whole, cut


@pytest.mark.parametrize("case,e,shards", [("cfg5", 3000, 2), ("cfg4", 2500, 3), ("cfg3", 1111, 2), ("cfg5_nooverlap", 1500, 4),
                                           ("cfg2", 100, 2)])
def test_shards_equal_one_batch(case, e, shards):
    """Ragged slices (the last one shorter, sizes that are no multiple of a tile), on-device agent order, same-step
    auto-reset with scattered start steps: every output of every step and the statistics match one handle's."""
    import torch
    spec, pool, whole, cut = make_pair(case, e, shards, autoreset=1, seed=5, env_id_offset=640)
    assert sum(hi - lo for lo, hi in cut.bounds) == e and len(cut.shards) <= shards
    a = spec["num_agents"]
    ow, oc = whole.reset(), cut.reset()
    torch.cuda.synchronize()
    assert torch.equal(ow.image, oc.image) and torch.equal(ow.direction, oc.direction)
    g = torch.Generator(device="cuda:0").manual_seed(3)
    steps0 = torch.randint(0, spec["max_steps"], (e,), generator=g, device="cuda:0", dtype=torch.int32)
    whole.set_state(steps=steps0)
    cut.set_state(steps=steps0)
    for t in range(60):
        acts = torch.randint(0, 7, (e, a), generator=g, device="cuda:0", dtype=torch.uint8)
        w, c = whole.step(acts), cut.step(acts)
        torch.cuda.synchronize()
        for f in FIELDS:
            assert torch.equal(getattr(w, f), getattr(c, f)), f"{case} step {t} {f}"
    sw, sc = whole.stats().cpu().numpy(), cut.stats().cpu().numpy()
    assert sw[0] > 0 and (sw == sc).all(), (sw, sc)
    st_w, st_c = whole.get_state(), cut.get_state()
    for k in st_w:
        assert torch.equal(st_w[k], st_c[k]), k
    whole.close()
    cut.close()


def test_free_running_shards_against_oracle():
    """The slices step without waiting for each other (no fork, no join: the bench's mode) through inputs that are
    complete beforehand; outputs go to per-step tensors and are compared with the C oracle afterwards."""
    import torch
    from multigrid_amd.sharded import StreamShardedMultiGrid
    from oracle.coracle import BatchOracle
    from multigrid_amd.batched import StepOutput
    name, agents, extra = CASES["cfg4"]
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(dict(agent_view_size=7), **extra))
    pool = layouts.make_pool(gen, 32, seed0=3)
    e, a, v, n = 2000, agents, 7, 40
    env = StreamShardedMultiGrid(spec, e, shards=2, device=0)
    env.upload_layouts(pool[0], pool[1], pool[2])
    ora = BatchOracle(spec, e)
    rng = np.random.default_rng(8)
    idx = rng.integers(0, 32, size=e).astype(np.int32)
    out = env.reset(layout_idx=idx)
    ora.load(np.arange(e), pool[0][idx], pool[1][idx], pool[2][idx])
    ins = [rand_inputs(rng, e, a) for _ in range(n)]
    acts = torch.as_tensor(np.stack([i[0] for i in ins])).cuda()
    order = torch.as_tensor(np.stack([i[1] for i in ins])).cuda()
    torch.cuda.synchronize()
    # every step writes its own slice of [n, ...] tensors: point the shards' output block there step by step
    keep = StepOutput(image=torch.zeros((n, e, a, v, v, 3), dtype=torch.uint8, device="cuda:0"),
                      direction=torch.zeros((n, e, a), dtype=torch.uint8, device="cuda:0"),
                      reward=torch.zeros((n, e, a), dtype=torch.float32, device="cuda:0"),
                      terminated=torch.zeros((n, e, a), dtype=torch.uint8, device="cuda:0"),
                      truncated=torch.zeros((n, e, a), dtype=torch.uint8, device="cuda:0"))
    torch.cuda.synchronize()
    from multigrid_amd.batched import BatchedMultiGrid
    for t in range(n):
        env._mgout = BatchedMultiGrid._make_out(StepOutput(keep.image[t], keep.direction[t], keep.reward[t],
                                                           keep.terminated[t], keep.truncated[t]))
        env.step(acts[t], order[t], fork=False, join=False)
    env.synchronize()
    for t in range(n):
        o = ora.step(*ins[t])
        compare(StepOutput(keep.image[t], keep.direction[t], keep.reward[t], keep.terminated[t], keep.truncated[t]),
                o, f"free-running step {t}")
    compare_state(env, ora, "free-running end")
    env.close()


def test_shard_list_is_checked():
    """mg_step_shards refuses handles that are not consecutive slices of one batch."""
    import torch
    from multigrid_amd import _lib
    from multigrid_amd.sharded import StreamShardedMultiGrid
    name, agents, extra = CASES["cfg1"]
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(dict(agent_view_size=7), **extra))
    pool = layouts.make_pool(gen, 4, seed0=0)
    env = StreamShardedMultiGrid(spec, 256, shards=2, device=0)
    env.upload_layouts(pool[0], pool[1], pool[2])
    env.reset()
    acts = torch.zeros((256, agents), dtype=torch.uint8, device="cuda:0")
    env.step(acts)
    swapped = (ctypes.c_void_p * 2)(env._handles[1], env._handles[0])
    rc = _lib.lib().mg_step_shards(swapped, 2, ctypes.c_void_p(acts.data_ptr()), None, ctypes.byref(env._mgout), None)
    assert rc == -1 and b"env_id_offset" in _lib.lib().mg_last_error(ctypes.c_void_p(env._handles[1]))
    twice = (ctypes.c_void_p * 2)(env._handles[0], env._handles[0])
    assert _lib.lib().mg_step_shards(twice, 2, ctypes.c_void_p(acts.data_ptr()), None, ctypes.byref(env._mgout), None) == -1
    assert _lib.lib().mg_step_shards(None, 0, None, None, None, None) == -1
    torch.cuda.synchronize()
    env.close()
    # a slice that cannot step keeps every slice from stepping
    env = StreamShardedMultiGrid(spec, 256, shards=2, device=0)
    env.upload_layouts(pool[0], pool[1], pool[2])
    env.shards[0].reset()
    with pytest.raises(_lib.MgError, match="shard 1"):
        env.step(acts)
    torch.cuda.synchronize()
    assert int(env.shards[0].stats()[5]) == 0
    env.close()


@pytest.mark.parametrize("case,e", [("cfg5", 1500), ("cfg3", 2100)])
def test_fused_steps_per_shard(case, e):
    """step_n on shards: every slice takes T steps in one fused launch on its stream; together they equal T steps of
    one handle holding the whole batch (on-device agent order, same-step auto-reset)."""
    import torch
    spec, pool, whole, cut = make_pair(case, e, 2, autoreset=1, seed=9)
    a, T = spec["num_agents"], 9
    whole.reset()
    cut.reset()
    g = torch.Generator(device="cuda:0").manual_seed(4)
    steps0 = torch.randint(0, spec["max_steps"], (e,), generator=g, device="cuda:0", dtype=torch.int32)
    whole.set_state(steps=steps0)
    cut.set_state(steps=steps0)
    for rnd in range(3):
        acts = torch.randint(0, 7, (T, e, a), generator=g, device="cuda:0", dtype=torch.uint8)
        w = whole.step_n(acts)
        parts = cut.step_n(acts)
        torch.cuda.synchronize()
        for f in FIELDS:
            got = torch.cat([getattr(p, f) for p in parts], dim=1)
            assert torch.equal(getattr(w, f), got), f"{case} round {rnd} {f}"
    st_w, st_c = whole.get_state(), cut.get_state()
    for k in st_w:
        assert torch.equal(st_w[k], st_c[k]), k
    assert (whole.stats().cpu() == cut.stats().cpu()).all()
    whole.close()
    cut.close()


def test_batch_views_over_shards():
    """The RLlib / PettingZoo batch views take a sharded batch in place of a single handle."""
    import torch
    from multigrid_amd.views import BatchedParallelView, BatchedRLlibView
    spec, pool, whole, cut = make_pair("odd3", 700, 2, autoreset=0)
    a = spec["num_agents"]
    vw, vc = BatchedRLlibView(whole), BatchedRLlibView(cut)
    pw, pc = BatchedParallelView(whole), BatchedParallelView(cut)
    vw.reset()
    vc.reset()
    g = torch.Generator(device="cuda:0").manual_seed(2)
    for t in range(35):
        acts = torch.randint(0, 7, (700, a), generator=g, device="cuda:0", dtype=torch.uint8)
        order = torch.argsort(torch.rand((700, a), generator=g, device="cuda:0"), dim=1).to(torch.uint8)
        rw, rc = vw.step(acts, order), vc.step(acts, order)
        torch.cuda.synchronize()
        assert torch.equal(rw[0]["image"], rc[0]["image"]) and torch.equal(rw[1], rc[1])
        for k in ("agents", "__all__"):
            assert torch.equal(rw[2][k], rc[2][k]) and torch.equal(rw[3][k], rc[3][k])
        assert torch.equal(pw.agents_mask, pc.agents_mask) and torch.equal(pw.is_done(), pc.is_done())
    whole.close()
    cut.close()


def test_closed_loop_per_slice_equals_whole_batch():
    """Observations -> policy -> actions -> step with each slice's policy and step on the slice's own stream (no
    fork, no join: nothing outside a slice is read) ends in the state of the same loop over one handle."""
    import torch
    spec, pool, whole, cut = make_pair("cfg4", 1800, 2, autoreset=1, seed=3)
    a = spec["num_agents"]

    def policy(image, direction, t):
        ahead = image[:, :, 3, 5, 0]
        mix = image[:, :, 3, 6, 1].to(torch.int32) + image[:, :, 2, 6, 0] + direction + t
        return torch.where(ahead == 1, 2, 3 + mix % 4).to(torch.uint8)  # forward, or pickup / drop / toggle / done

    ow, oc = whole.reset(), cut.reset()
    torch.cuda.synchronize()
    for t in range(80):
        whole.step(policy(ow.image, ow.direction, t))
    for k, (sh, (lo, hi)) in enumerate(zip(cut.shards, cut.bounds)):
        with torch.cuda.stream(cut.streams[k]):  # the whole run of slice k is queued before slice k + 1 starts
            for t in range(80):
                sh.step(policy(oc.image[lo:hi], oc.direction[lo:hi], t))
    torch.cuda.synchronize()
    for f in FIELDS:
        assert torch.equal(getattr(ow, f), getattr(oc, f)), f
    st_w, st_c = whole.get_state(), cut.get_state()
    for key in st_w:
        assert torch.equal(st_w[key], st_c[key]), key
    assert (whole.stats().cpu() == cut.stats().cpu()).all()
    whole.close()
    cut.close()


@pytest.mark.parametrize("case,e,mode", [("cfg5", 900, 2), ("cfg3", 1300, 2), ("odd3", 640, 0)])
def test_shards_with_one_hot_images_and_other_reset_modes(case, e, mode):
    """One-hot images written by the slices' launches (every slice's part of the array starts 8-byte aligned), with
    next-step auto-reset and without auto-reset, injected agent orders."""
    import torch
    spec, pool, whole, cut = make_pair(case, e, 3, autoreset=mode, seed=1, one_hot=True)
    a = spec["num_agents"]
    ow, oc = whole.reset(), cut.reset()
    torch.cuda.synchronize()
    assert torch.equal(ow.onehot, oc.onehot)
    g = torch.Generator(device="cuda:0").manual_seed(6)
    for t in range(45):
        acts = torch.randint(0, 7, (e, a), generator=g, device="cuda:0", dtype=torch.uint8)
        order = torch.argsort(torch.rand((e, a), generator=g, device="cuda:0"), dim=1).to(torch.uint8)
        w, c = whole.step(acts, order), cut.step(acts, order)
        torch.cuda.synchronize()
        for f in FIELDS + ("onehot",):
            assert torch.equal(getattr(w, f), getattr(c, f)), f"{case} step {t} {f}"
    whole.close()
    cut.close()


def test_shards_of_competitive_environments():
    """CompetativeRedBlueDoorEnvV3 (death match, per-agent reward schemes set on every slice's handle) as shards."""
    import torch
    from multigrid_amd.batched import BatchedMultiGrid
    from multigrid_amd.sharded import StreamShardedMultiGrid
    spec, gen = envs.make_spec("MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2-Death-Match", max_steps=40)
    pool = layouts.make_pool(gen, 32, seed0=5)
    e, a = 1200, spec["num_agents"]
    whole = BatchedMultiGrid(spec, e, device=0, autoreset=1)
    cut = StreamShardedMultiGrid(spec, e, shards=2, device=0, autoreset=1)
    for x in (whole, cut):
        x.upload_layouts(pool[0], pool[1], pool[2])
        x.reset()
    g = torch.Generator(device="cuda:0").manual_seed(12)
    paid = 0
    for t in range(120):
        # forward, pickup and toggle often: keys, balls, doors and opponents are reached
        acts = torch.multinomial(torch.tensor([0.1, 0.1, 0.3, 0.2, 0.05, 0.23, 0.02], device="cuda:0"), e * a,
                                 replacement=True, generator=g).view(e, a).to(torch.uint8)
        order = torch.argsort(torch.rand((e, a), generator=g, device="cuda:0"), dim=1).to(torch.uint8)
        w, c = whole.step(acts, order), cut.step(acts, order)
        torch.cuda.synchronize()
        for f in FIELDS:
            assert torch.equal(getattr(w, f), getattr(c, f)), f"step {t} {f}"
        paid += int((w.reward != 0).sum())
    assert paid > 0
    st_w, st_c = whole.get_state(), cut.get_state()
    for k in st_w:
        assert torch.equal(st_w[k], st_c[k]), k
    whole.close()
    cut.close()


def test_unforked_steps_with_inputs_that_need_a_copy():
    """fork=False is only honoured for inputs used as they are.  Host arrays (copied to the device on the current
    stream) and a whole-batch [T, E, A] tensor for step_n (cut into per-slice copies there) make the call wait for
    those copies whatever the caller said; results equal one handle's."""
    import torch
    spec, pool, whole, cut = make_pair("cfg4", 1900, 2, autoreset=1, seed=9)
    a, e = spec["num_agents"], 1900
    whole.reset()
    cut.reset()
    rng = np.random.default_rng(4)
    for t in range(25):
        acts = rng.integers(0, 7, size=(e, a)).astype(np.uint8)       # host array: uploaded inside step()
        order = np.argsort(rng.random((e, a)), axis=1).astype(np.uint8)
        w = whole.step(acts, order)
        c = cut.step(acts, order, fork=False, join=False)
        cut.join()
        torch.cuda.synchronize()
        for f in FIELDS:
            assert torch.equal(getattr(w, f), getattr(c, f)), f"step {t} {f}"
    T = 6
    acts = torch.from_numpy(rng.integers(0, 7, size=(T, e, a)).astype(np.uint8)).cuda()
    wn = whole.step_n(acts)
    parts = cut.step_n(acts, fork=False, join=False)  # whole-batch tensor: per-slice copies are made inside
    cut.join()
    torch.cuda.synchronize()
    for (lo, hi), o in zip(cut.bounds, parts):
        for f in FIELDS:
            assert torch.equal(getattr(wn, f)[:, lo:hi], getattr(o, f)), f"step_n {f}"
    whole.close()
    cut.close()
