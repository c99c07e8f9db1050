"""mg_gen_obs_of_state - the entry point that allocates nothing - through the C ABI: against the C oracle's gen_obs of
the same state and against mg_gen_obs of a handle holding it, on two calls in a row (the second with a smaller batch);
that it really allocates nothing (device-wide free memory, and a call recorded under HIP graph capture, where an
allocation fails the capture); the shared-buffer contract on a second stream; the size limit."""
import numpy as np
import pytest

from multigrid_amd import envs, layouts

pytestmark = pytest.mark.gpu

CASES = [
    ("MultiGrid-LockedHallway-4Rooms-v0", 4, 7, 3000, 700),
    ("MultiGrid-Empty-16x16-v0", 8, 7, 4000, 129),
    ("MultiGrid-BlockedUnlockPickup-v0", 2, 5, 9000, 1),
    ("MultiGrid-RedBlueDoors-8x8-v0", 3, 3, 2500, 2499),
]


def state_after_some_steps(name, agents, view, e, seed):
    """A handle stepped for a while: its state in the reference's layouts, its own mg_gen_obs output, and the C
    oracle's gen_obs (multigrid/utils/obs.py:64-99 restated in oracle/mg_oracle.c) of that state."""
    import torch
    from multigrid_amd.batched import BatchedMultiGrid, StepOutput
    from oracle.coracle import BatchOracle
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=view, max_steps=50)
    pool = layouts.make_pool(gen, 32, seed0=seed)
    env = BatchedMultiGrid(spec, e, device=0, one_hot=True)
    env.upload_layouts(pool[0], pool[1], pool[2])
    rng = np.random.default_rng(seed)
    env.reset(layout_idx=rng.integers(0, 32, size=e))
    for _ in range(12):
        env.step(rng.integers(0, 7, size=(e, agents)).astype(np.uint8),
                 np.argsort(rng.random((e, agents)), axis=1).astype(np.uint8))
    st = env.get_state()
    ref = StepOutput(image=torch.full_like(env.out.image, 201), direction=torch.full_like(env.out.direction, 202),
                     reward=env.out.reward, terminated=env.out.terminated, truncated=env.out.truncated,
                     onehot=torch.full_like(env.out.onehot, 203))
    env.gen_obs(out=ref)
    torch.cuda.synchronize()
    env.close()
    ora = BatchOracle(spec, e)
    ora.load(np.arange(e), st["grid"].cpu().numpy(), st["agents"].cpu().numpy(), st["meta"].cpu().numpy())
    return spec, st["grid"].clone(), st["agents"].clone(), ref, ora.gen_obs().copy()


@pytest.mark.parametrize("name,agents,view,e_first,e_second", CASES)
def test_no_alloc_gen_obs_matches_oracle_and_handle(name, agents, view, e_first, e_second):
    import torch
    from multigrid_amd.batched import gen_obs_of_state
    for e, seed in ((e_first, 5), (e_second, 6)):  # the second call reuses the library's buffer for a smaller batch
        spec, grid, ag, ref, want = state_after_some_steps(name, agents, view, e, seed)
        image = torch.full_like(ref.image, 99)
        direction = torch.full_like(ref.direction, 98)
        onehot = torch.full_like(ref.onehot, 97)
        gen_obs_of_state(spec, grid, ag, image=image, direction=direction, onehot=onehot)
        torch.cuda.synchronize()
        assert np.array_equal(image.cpu().numpy(), want), f"{name} E={e} image against the oracle"
        assert np.array_equal(direction.cpu().numpy(), ag[:, :, 2].cpu().numpy()), f"{name} E={e} direction"
        assert torch.equal(image, ref.image), f"{name} E={e} image"
        assert torch.equal(direction, ref.direction), f"{name} E={e} direction"
        assert torch.equal(onehot, ref.onehot), f"{name} E={e} one-hot image"
        # image alone, twice on the same stream
        image.fill_(0)
        gen_obs_of_state(spec, grid, ag, image=image)
        gen_obs_of_state(spec, grid, ag, image=image)
        torch.cuda.synchronize()
        assert torch.equal(image, ref.image)


def test_no_alloc_gen_obs_leaves_device_memory_alone():
    """Free device memory as the driver reports it (not the framework's allocator, which a hipMalloc inside the
    library would bypass) is the same before and after calls that follow the first one on this device."""
    import torch
    from multigrid_amd.batched import gen_obs_of_state
    spec, grid, ag, ref, want = state_after_some_steps("MultiGrid-LockedHallway-4Rooms-v0", 4, 7, 2000, 11)
    image = torch.full_like(ref.image, 9)
    onehot = torch.full_like(ref.onehot, 9)
    gen_obs_of_state(spec, grid, ag, image=image, onehot=onehot)  # (the buffer's address is looked up once per device)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(20):
        gen_obs_of_state(spec, grid, ag, image=image, onehot=onehot)
        gen_obs_of_state(spec, grid[:301], ag[:301], image=image[:301])
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info(0)[0] == free0, "a call changed the device's free memory"
    gen_obs_of_state(spec, grid, ag, image=image)
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), want)


def test_no_alloc_gen_obs_under_graph_capture():
    """The use the entry point exists for: recorded into a HIP graph (an allocation inside the call would fail the
    capture), replayed on fresh inputs, checked against the oracle."""
    import torch
    from multigrid_amd.batched import gen_obs_of_state
    spec, grid, ag, ref, want = state_after_some_steps("MultiGrid-Empty-16x16-v0", 8, 7, 1500, 21)
    spec2, grid2, ag2, ref2, want2 = state_after_some_steps("MultiGrid-Empty-16x16-v0", 8, 7, 1500, 22)
    g_in, a_in = grid.clone(), ag.clone()
    image = torch.zeros_like(ref.image)
    direction = torch.zeros_like(ref.direction)
    gen_obs_of_state(spec, g_in, a_in, image=image, direction=direction)  # warm: per-device set-up happens here
    torch.cuda.synchronize()
    image.zero_()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(graph, stream=side):
        gen_obs_of_state(spec, g_in, a_in, image=image, direction=direction)
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), want)
    g_in.copy_(grid2)
    a_in.copy_(ag2)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(image.cpu().numpy(), want2)
    assert torch.equal(direction, ref2.direction)


def test_no_alloc_gen_obs_on_a_second_stream_after_an_event():
    """All calls on one GPU share the library's buffer: a call on another stream is safe once that stream has waited
    for the earlier call (the documented contract)."""
    import torch
    from multigrid_amd.batched import gen_obs_of_state
    spec, grid, ag, ref, want = state_after_some_steps("MultiGrid-BlockedUnlockPickup-v0", 2, 7, 6000, 31)
    spec2, grid2, ag2, ref2, want2 = state_after_some_steps("MultiGrid-BlockedUnlockPickup-v0", 2, 7, 500, 32)
    img1, img2 = torch.zeros_like(ref.image), torch.zeros_like(ref2.image)
    other = torch.cuda.Stream()
    gen_obs_of_state(spec, grid, ag, image=img1)
    done = torch.cuda.Event()
    done.record()
    other.wait_event(done)
    with torch.cuda.stream(other):
        gen_obs_of_state(spec2, grid2, ag2, image=img2)
    torch.cuda.synchronize()
    assert np.array_equal(img1.cpu().numpy(), want)
    assert np.array_equal(img2.cpu().numpy(), want2)


def test_no_alloc_gen_obs_rejects_what_does_not_fit():
    import torch
    from multigrid_amd._lib import MgError
    from multigrid_amd.batched import gen_obs_of_state
    spec, _ = envs.make_spec("MultiGrid-Empty-16x16-v0", num_agents=8, agent_view_size=7)
    e = 14000  # 14000 * (112 + 512) bytes > 8 MiB
    grid = torch.zeros((e, 16, 16, 3), dtype=torch.uint8, device="cuda:0")
    ag = torch.zeros((e, 8, 8), dtype=torch.uint8, device="cuda:0")
    image = torch.zeros((e, 8, 7, 7, 3), dtype=torch.uint8, device="cuda:0")
    with pytest.raises(MgError, match="at most"):
        gen_obs_of_state(spec, grid, ag, image=image)
