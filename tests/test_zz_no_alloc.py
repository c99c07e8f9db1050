"""mg_gen_obs_of_state - the entry point that allocates nothing - against mg_gen_obs of a handle holding the
same state, through the C ABI: two calls in a row, the second with a smaller batch, and the size limit."""
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
    """A handle stepped for a while: its state in the reference's layouts and its own mg_gen_obs output."""
    import torch
    from multigrid_amd.batched import BatchedMultiGrid, StepOutput
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
    return spec, st["grid"].clone(), st["agents"].clone(), ref


@pytest.mark.parametrize("name,agents,view,e_first,e_second", CASES)
def test_no_alloc_gen_obs_matches_handle_gen_obs(name, agents, view, e_first, e_second):
    import torch
    from multigrid_amd.batched import gen_obs_of_state
    for e, seed in ((e_first, 5), (e_second, 6)):  # the second call reuses the library's buffer for a smaller batch
        spec, grid, ag, ref = state_after_some_steps(name, agents, view, e, seed)
        image = torch.full_like(ref.image, 99)
        direction = torch.full_like(ref.direction, 98)
        onehot = torch.full_like(ref.onehot, 97)
        before = torch.cuda.memory_allocated()
        gen_obs_of_state(spec, grid, ag, image=image, direction=direction, onehot=onehot)
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        assert torch.equal(image, ref.image), f"{name} E={e} image"
        assert torch.equal(direction, ref.direction), f"{name} E={e} direction"
        assert torch.equal(onehot, ref.onehot), f"{name} E={e} one-hot image"
        # image alone, twice on the same stream
        image.fill_(0)
        gen_obs_of_state(spec, grid, ag, image=image)
        gen_obs_of_state(spec, grid, ag, image=image)
        torch.cuda.synchronize()
        assert torch.equal(image, ref.image)


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
