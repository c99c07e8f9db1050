"""The drop-in MultiGridEnv facade reproduces the reference's trajectories from SEEDS alone:
layouts come from the host generators, agent orders from np_random, exactly as in the reference."""
import numpy as np
import pytest

from golden_util import INFO_KEYS, MISSION_TEXTS, load_group, pack_info, pack_info_dict

pytestmark = pytest.mark.gpu

# (fixture group, registry name, agents, seed0 used by tests/golden/make_golden.py, kwargs)
CASES = [
    ("cfg1_empty8", "MultiGrid-Empty-8x8-v0", 2, 11, {}),
    ("cfg2_redbluedoors8", "MultiGrid-RedBlueDoors-8x8-v0", 2, 12, {}),
    ("cfg3_blockedunlockpickup", "MultiGrid-BlockedUnlockPickup-v0", 2, 13, {}),
    ("cfg4_lockedhallway4", "MultiGrid-LockedHallway-4Rooms-v0", 4, 15, {}),
    ("cfg5_empty16", "MultiGrid-Empty-16x16-v0", 8, 17, {}),
    # constructed with the reference's own defaults: 5 x 5 views (multigrid/base.py:96)
    ("default_empty8", "MultiGrid-Empty-8x8-v0", 2, 21, {"agent_view_size": None}),
    # registered configurations through 9 x 9 and 11 x 11 views (the reference takes any odd size, base.py:96)
    ("seeded_view9_lockedhallway2", "MultiGrid-LockedHallway-2Rooms-v0", 3, 57, {"agent_view_size": 9}),
    ("seeded_view11_blockedunlockpickup", "MultiGrid-BlockedUnlockPickup-v0", 2, 58, {"agent_view_size": 11}),
    # eight rooms: more rooms than door colours (locked_hallway.py:150-151), played by the recording's planner
    ("solve_lockedhallway8", "MultiGrid-LockedHallway-6Rooms-v0", 2, 55, {"num_rooms": 8, "room_size": 4, "max_steps": 2000}),
]


@pytest.mark.parametrize("pinned_blocks", [True, False])
@pytest.mark.parametrize("group,name,agents,seed0,kw", CASES)
def test_drop_in_env_matches_reference_from_seed(group, name, agents, seed0, kw, pinned_blocks, monkeypatch):
    """``pinned_blocks``: the kernel reads actions from and writes its outputs to pinned host memory (one launch and one
    wait per step); otherwise the blocks are device memory and travel by one upload and one download."""
    from multigrid_amd import gym_api
    from multigrid_amd.gym_api import MultiGridEnv, PettingZooWrapper, RLlibWrapper
    if not pinned_blocks:
        monkeypatch.setattr(gym_api, "_gpu_addresses", lambda t: False)

    spec, d = load_group(group)
    kw = dict(kw)
    if kw.get("agent_view_size", 7) is None:
        kw.pop("agent_view_size")  # the facade's default must be the reference's
    else:
        kw.setdefault("agent_view_size", 7)
    view = spec["view_size"]
    for n in range(min(3, len(d["length"]))):
        env = MultiGridEnv(name, agents=agents, layout_rng=np.random.default_rng(seed0 * 1000 + n), **kw)
        assert env._zero_copy == pinned_blocks  # (this GPU maps pinned memory at its host address)
        obs, infos = env.reset(seed=seed0 * 7919 + n)
        assert set(obs) == set(range(agents)) and obs[0]["image"].dtype == np.int64
        np.testing.assert_array_equal(np.stack([obs[i]["image"] for i in range(agents)]), d["reset_image"][n])
        np.testing.assert_array_equal(env.grid_state, d["layout_grid"][n])
        wrapped = RLlibWrapper(env)
        for t in range(min(500 if group.startswith("solve") else 60, int(d["length"][n]))):  # (doors open late)
            acts = {i: int(d["actions"][n, t, i]) for i in range(agents)}
            o, r, term, trunc, info = wrapped.step(acts)
            np.testing.assert_array_equal(np.stack([o[i]["image"] for i in range(agents)]), d["image"][n, t],
                                          err_msg=f"{group}[{n}] step {t}")
            assert [int(o[i]["direction"]) for i in range(agents)] == list(d["direction"][n, t])
            assert [float(r[i]) for i in range(agents)] == list(d["reward"][n, t])  # the reference's doubles, bit for bit
            assert [bool(term[i]) for i in range(agents)] == list(d["terminated"][n, t])
            assert [trunc[i] for i in range(agents)] == list(d["truncated"][n, t])
            assert term["__all__"] == all(d["terminated"][n, t]) and trunc["__all__"] == all(d["truncated"][n, t])
            if "BlockedUnlockPickup" not in name:  # there the reference hands agents a random mission (parity unpinned)
                assert str(o[0]["mission"]) == spec["mission"] and o[0]["mission"].string == spec["mission"]
        pz = PettingZooWrapper(env)
        # spaces: names "{color}_{team_index}" (base.py:262-265), image Box(0, 255, (V, V, 3)), 7 actions
        assert list(env.observation_space.keys()) == list(env.action_space.keys()) == env.agent_names
        assert env.agent_names[0] == "red_0" and len(env.agent_names) == agents
        sp = pz.observation_space(0)
        assert sp["image"].shape == (view, view, 3) == o[0]["image"].shape and sp["image"].contains(o[0]["image"]) and sp["direction"].n == 4
        assert pz.action_space(agents - 1).n == 7 and set(pz.action_spaces) == set(range(agents))
        assert pz.possible_agents == list(range(agents))
        assert pz.agents == ([] if env.is_done() else [i for i in range(agents) if not d["post_agents"][n, t, i, 3]])
        env.close()


def test_competitive_env_from_seed_dtde():
    """CompetativeRedBlueDoorEnvV3 (DTDE 1v1 as registered): name-keyed dicts, agents acting in dict order."""
    from multigrid_amd.gym_api import CompetativeRedBlueDoorEnv, make

    spec, d = load_group("comp_1v1_plain")
    names = spec["agent_names"]
    a = len(names)
    for n in range(3):
        env = make("MultiGrid-CompetativeRedBlueDoor-v3-DTDE-1v1", layout_rng=np.random.default_rng(61 * 1000 + n))
        assert isinstance(env, CompetativeRedBlueDoorEnv) and env.agent_names == names and env.teams == {"blue": 1, "red": 1}
        obs, infos = env.reset(seed=61 * 7919 + n)
        assert list(obs) == names and obs[names[0]]["image"].shape == (5, 5, 3)
        # (the first text is a sample of the mission space drawn with the reset seed: parity unpinned, see gym_api)
        assert [obs[k]["mission"].string for k in names] == [MISSION_TEXTS[j] for j in d["reset_mission"][n]]
        np.testing.assert_array_equal(np.stack([obs[k]["image"] for k in names]), d["reset_image"][n])
        np.testing.assert_array_equal(env.grid_state, d["layout_grid"][n])
        for t in range(min(150, int(d["length"][n]))):
            acts = {names[i]: int(d["actions"][n, t, i]) for i in d["order"][n, t]}
            o, r, term, trunc, info = env.step(acts)
            where = f"comp_1v1_plain[{n}] step {t}"
            np.testing.assert_array_equal(np.stack([o[k]["image"] for k in names]), d["image"][n, t], err_msg=where)
            assert [int(o[k]["direction"]) for k in names] == list(d["direction"][n, t]), where
            np.testing.assert_allclose([r[k] for k in names], d["reward"][n, t], atol=1e-6, rtol=0, err_msg=where)
            assert [term[k] for k in names] == list(d["terminated"][n, t]), where
            assert [trunc[k] for k in names] == list(d["truncated"][n, t]), where
            # the mission texts, rewritten after rewarded key and ball pickups (:438-468)
            assert [o[k]["mission"].string for k in names] == [MISSION_TEXTS[j] for j in d["mission"][n, t]], where
            assert list(info) == names and list(info[names[0]]) == list(INFO_KEYS), where
            np.testing.assert_array_equal(pack_info_dict(info, names), pack_info(d["info"][n, t]), err_msg=where + " info")
        env.close()


def test_competitive_env_ctce_team_outputs():
    """training_scheme "CTCE": actions, observations, rewards and flags per team (ctce_step)."""
    from multigrid_amd.gym_api import make

    spec, d = load_group("ctce_2v2")
    members = spec["team_members"]
    for n in range(len(d["length"])):
        env = make("MultiGrid-CompetativeRedBlueDoor-v3-CTCE-2v2", agent_view_size=7, max_steps=80)
        teams = list(env.team_members)
        assert [env.team_members[t] for t in teams] == members
        env.reset(seed=n)
        env._sim.set_state(grid=d["layout_grid"][n][None], agents=d["layout_agents"][n][None])  # the harness-edited start
        env._missions = d["reset_mission"][n].tolist()  # (... and the texts its episode began with)
        obs = env.gen_obs()
        np.testing.assert_array_equal(np.stack([obs[t][j]["image"] for t in teams for j in range(len(members[0]))]),
                                      d["reset_image"][n])
        for t in range(int(d["length"][n])):
            order = list(d["order"][n, t])
            team_order = []
            for k in order:
                tm = next(tt for tt in teams if k in env.team_members[tt])
                if tm not in team_order:
                    team_order.append(tm)
            acts = {tm: [int(d["actions"][n, t, k]) for k in env.team_members[tm]] for tm in team_order}
            o, r, term, trunc, info = env.step(acts)
            where = f"ctce_2v2[{n}] step {t}"
            np.testing.assert_array_equal(pack_info_dict(info, teams), pack_info(d["info"][n, t]), err_msg=where + " info")
            img = np.zeros_like(d["image"][n, t])
            for tm in teams:
                for j, k in enumerate(env.team_members[tm]):
                    assert o[tm][j]["agent_id"] == j
                    img[k] = o[tm][j]["image"]
                    assert o[tm][j]["mission"].string == MISSION_TEXTS[d["mission"][n, t, k]], where
            np.testing.assert_array_equal(img, d["image"][n, t], err_msg=where)
            np.testing.assert_allclose([r[tm] for tm in teams], d["reward"][n, t], atol=2e-6, rtol=0, err_msg=where)
            assert [term[tm] for tm in teams] == list(d["terminated"][n, t]), where
            assert [trunc[tm] for tm in teams] == list(d["truncated"][n, t]), where
        np.testing.assert_array_equal(env.grid_state, d["post_grid"][n, int(d["length"][n]) - 1])
        env.close()


def test_blockedunlockpickup_agents_get_a_sampled_mission():
    """multigrid/base.py:329-332: ``reset`` seeds the mission space with the reset seed and gives every agent ONE sample
    of it; BlockedUnlockPickup's space is colours x (box, key), so the observations' mission is a random one while
    ``env.mission`` names the box that was placed.  (Parity unpinned: gymnasium's sampler is not importable here.)"""
    from multigrid_amd.constants import COLOR_NAMES
    from multigrid_amd.gym_api import MultiGridEnv
    env = MultiGridEnv("MultiGrid-BlockedUnlockPickup-v0", agents=2, layout_rng=np.random.default_rng(1))
    space = {f"pick up the {c} {t}" for c in COLOR_NAMES for t in ("box", "key")}
    seen = set()
    for seed in range(40):
        obs, _ = env.reset(seed=seed)
        first = obs[0]["mission"].string
        assert first in space and obs[1]["mission"] is obs[0]["mission"]
        assert env.mission.string.endswith(" box") and env.mission.string in space
        rng = np.random.default_rng(seed)
        idx = (rng.random(2) * np.array([6, 2])).astype(np.int64)
        assert first == f"pick up the {COLOR_NAMES[idx[0]]} {('box', 'key')[idx[1]]}"
        o, *_ = env.step({0: 0, 1: 1})
        assert o[0]["mission"].string == first  # kept for the episode
        seen.add(first)
    assert len(seen) >= 8
    env.close()
    other = MultiGridEnv("MultiGrid-Empty-8x8-v0", agents=2)
    obs, _ = other.reset(seed=3)
    assert obs[0]["mission"] is other.mission and other.mission.string == "get to the green goal square"
    other.close()
