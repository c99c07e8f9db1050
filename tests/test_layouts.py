"""Host-side layout generators reproduce the reference's reset() layouts for the same generators."""
import numpy as np
import pytest

from golden_util import load_layout_fixture
from multigrid_amd import envs

CASES = ["MultiGrid-Empty-8x8-v0", "MultiGrid-RedBlueDoors-8x8-v0", "MultiGrid-BlockedUnlockPickup-v0",
         "MultiGrid-LockedHallway-4Rooms-v0", "MultiGrid-LockedHallway-2Rooms-v0",
         "MultiGrid-LockedHallway-6Rooms-v0", "MultiGrid-Empty-16x16-v0", "MultiGrid-RedBlueDoors-6x6-v0",
         "MultiGrid-Empty-Random-6x6-v0",
         "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-1v1", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2",
         "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2-Death-Match",
         "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-Red-Single-with-Obstacle"]


@pytest.mark.parametrize("name", CASES)
def test_layout_generator_matches_reference(name):
    fx = load_layout_fixture()
    key = name.replace("MultiGrid-", "").replace("-", "_")
    a = int(fx[key + "__num_agents"])
    spec, gen = envs.make_spec(name, num_agents=a)
    assert spec["num_agents"] == a
    for seed in range(fx[key + "__grid"].shape[0]):
        # the fixture generator seeds RandomMixin with `seed` and calls reset(seed=seed + 100)
        grid, agents, meta, mission = gen(mixin_rng=np.random.default_rng(seed),
                                          env_rng=np.random.default_rng(seed + 100))
        np.testing.assert_array_equal(grid, fx[key + "__grid"][seed], err_msg=f"{name} seed {seed} grid")
        np.testing.assert_array_equal(agents, fx[key + "__agents"][seed], err_msg=f"{name} seed {seed} agents")
        np.testing.assert_array_equal(meta, fx[key + "__meta"][seed], err_msg=f"{name} seed {seed} meta")
        if "Competative" not in name:  # there the text is drawn by the mission space's own generator (parity unpinned)
            assert mission == str(fx[key + "__mission"][seed])
        assert grid.shape == (spec["width"], spec["height"], 3)


def test_lockedhallway_with_more_rooms_than_colours():
    """Eight rooms: door colours repeat (locked_hallway.py:150-151).  The start states of the `solve_lockedhallway8`
    trajectories are untouched `reset()` layouts of the reference (RandomMixin seeded 55 * 1000 + n, reset(seed=55 * 7919
    + n), tests/golden/make_golden.py); meta[3] is len(env.rooms), the number of distinct door colours."""
    from golden_util import load_group
    ref_spec, d = load_group("solve_lockedhallway8")
    spec, gen = envs.make_spec("MultiGrid-LockedHallway-6Rooms-v0", num_agents=2, num_rooms=8, room_size=4,
                               max_steps=2000, agent_view_size=7)
    for dct in (spec, ref_spec):
        dct.pop("mission")
        dct.pop("training_scheme", None)
    assert spec == ref_spec and spec["num_rooms"] == 8 and (spec["width"], spec["height"]) == (10, 13)
    for n in range(len(d["length"])):
        grid, agents, meta, _ = gen(mixin_rng=np.random.default_rng(55 * 1000 + n), env_rng=np.random.default_rng(55 * 7919 + n))
        np.testing.assert_array_equal(grid, d["layout_grid"][n])
        np.testing.assert_array_equal(agents, d["layout_agents"][n])
        np.testing.assert_array_equal(meta, d["layout_meta"][n])
        doors = grid[..., 0] == 4
        assert doors.sum() == 8 and len(set(grid[..., 1][doors].tolist())) == meta[3] < 8


@pytest.mark.parametrize("group,name,agents,kw", [
    ("cfg1_empty8", "MultiGrid-Empty-8x8-v0", 2, {}),
    ("cfg2_redbluedoors8", "MultiGrid-RedBlueDoors-8x8-v0", 2, {}),
    ("cfg3_blockedunlockpickup", "MultiGrid-BlockedUnlockPickup-v0", 2, {}),
    ("cfg4_lockedhallway4", "MultiGrid-LockedHallway-4Rooms-v0", 4, {}),
    ("cfg5_empty16", "MultiGrid-Empty-16x16-v0", 8, {}),
    ("cfg5_empty16_nooverlap", "MultiGrid-Empty-16x16-v0", 8, {"allow_agent_overlap": False}),
    ("short_lockedhallway2", "MultiGrid-LockedHallway-2Rooms-v0", 2, {"max_steps": 30}),
    ("default_empty8", "MultiGrid-Empty-8x8-v0", 2, {"agent_view_size": None}),  # the reference's own default view
    ("comp_1v1", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-1v1", None, {"agent_view_size": None}),
    ("comp_2v2", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2", None, {"max_steps": 120}),
    ("comp_2v2_deathmatch", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2-Death-Match", None,
     {"agent_view_size": None, "max_steps": 90}),
    ("comp_2v2_all", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2", None,
     {"max_steps": 100, "success_termination_mode": "all", "allow_agent_overlap": True}),
    ("comp_single", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-Red-Single-with-Obstacle", None, {"max_steps": 60}),
])
def test_registry_spec_matches_reference_env(group, name, agents, kw):
    """make_spec() derives the same geometry and flags the reference classes report about themselves."""
    from golden_util import load_group
    ref_spec, _ = load_group(group)
    kw = dict(kw)
    if kw.setdefault("agent_view_size", 7) is None:  # the recorded BASELINE groups pass 7 explicitly
        kw.pop("agent_view_size")
    spec, _ = envs.make_spec(name, num_agents=agents, **kw)
    ref_spec.pop("mission")
    spec.pop("mission")
    spec.pop("training_scheme", None)
    assert spec == ref_spec
