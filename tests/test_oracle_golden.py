"""The Python restatement (oracle/pyref.py) replays every reference trajectory bit for bit."""
import numpy as np
import pytest

from golden_util import INFO_MASK, MISSION_TEXTS as MISSIONS, group_names, load_group, pack_info, pack_info_dict, team_info
from multigrid_amd.gym_api import competitive_missions_after_step
from oracle import pyref


def make_spec(spec):
    return pyref.EnvSpec(**spec)


@pytest.mark.parametrize("name", group_names())
@pytest.mark.parametrize("fast_obs", [False, True])
def test_pyref_replays_reference(name, fast_obs):
    spec, d = load_group(name)
    a = spec["num_agents"]
    for n in range(len(d["length"])):
        env = pyref.Env(make_spec(spec), fast_obs=fast_obs)
        obs = env.load_layout(d["layout_grid"][n], d["layout_agents"][n], d["layout_meta"][n])
        img0 = np.stack([obs[i]["image"] for i in range(a)])
        np.testing.assert_array_equal(img0, d["reset_image"][n], err_msg=f"{name}[{n}] reset obs")
        if "mission" in d:  # CompetativeRedBlueDoorEnvV3: the texts the agents start with (a sample of the mission space)
            env.missions = [MISSIONS[k] for k in d["reset_mission"][n]]
        for t in range(int(d["length"][n])):
            acts = d["actions"][n, t]
            o, r, term, trunc, info = env.step({i: int(acts[i]) for i in range(a)}, d["order"][n, t])
            where = f"{name}[{n}] step {t}"
            if "info" in d:  # CompetativeRedBlueDoorEnvV3: the step's info dict
                np.testing.assert_array_equal(pack_info_dict(info, range(a)), pack_info(d["info"][n, t]), err_msg=where + " info")
            if "mission" in d:  # ... and the mission texts its observations carry, rewritten after rewarded pickups
                assert env.obs_missions == [MISSIONS[k] for k in d["mission"][n, t]], where + " missions"
            img = np.stack([o[i]["image"] for i in range(a)])
            np.testing.assert_array_equal(img, d["image"][n, t], err_msg=where + " image")
            np.testing.assert_array_equal([int(o[i]["direction"]) for i in range(a)],
                                          d["direction"][n, t], err_msg=where + " direction")
            np.testing.assert_array_equal([float(r[i]) for i in range(a)], d["reward"][n, t],
                                          err_msg=where + " reward")
            np.testing.assert_array_equal([term[i] for i in range(a)], d["terminated"][n, t],
                                          err_msg=where + " terminated")
            np.testing.assert_array_equal([trunc[i] for i in range(a)], d["truncated"][n, t],
                                          err_msg=where + " truncated")
            np.testing.assert_array_equal(env.grid, d["post_grid"][n, t], err_msg=where + " grid")
            np.testing.assert_array_equal(env.agents, d["post_agents"][n, t], err_msg=where + " agents")
            if "post_flags" in d:  # CompetativeRedBlueDoorEnvV3: what the arrays do not show
                assert env.flags == d["post_flags"][n, t], where + " flags"
                np.testing.assert_array_equal(env.meta, d["post_meta"][n, t], err_msg=where + " ball discount counters")


@pytest.mark.parametrize("name", group_names())
def test_c_oracle_replays_reference(name):
    """oracle/mg_oracle.c: all trajectories of a group stepped as one batch."""
    from oracle.coracle import BatchOracle

    spec, d = load_group(name)
    n = len(d["length"])
    bo = BatchOracle(spec, n)
    bo.load(np.arange(n), d["layout_grid"], d["layout_agents"], d["layout_meta"])
    np.testing.assert_array_equal(bo.gen_obs(), d["reset_image"], err_msg=f"{name} reset obs")
    missions = d["reset_mission"].tolist() if "mission" in d else None
    for t in range(int(d["length"].max())):
        live = d["length"] > t
        img, direction, rew, term, trunc = bo.step(d["actions"][:, t], d["order"][:, t])
        where = f"{name} step {t}"
        np.testing.assert_array_equal(img[live], d["image"][live, t], err_msg=where + " image")
        np.testing.assert_array_equal(direction[live], d["direction"][live, t], err_msg=where + " dir")
        np.testing.assert_array_equal(rew[live], d["reward"][live, t], err_msg=where + " reward")
        np.testing.assert_array_equal(term[live].astype(bool), d["terminated"][live, t], err_msg=where + " term")
        np.testing.assert_array_equal(trunc[live].astype(bool), d["truncated"][live, t], err_msg=where + " trunc")
        np.testing.assert_array_equal(bo.grid[live], d["post_grid"][live, t], err_msg=where + " grid")
        np.testing.assert_array_equal(bo.agents[live], d["post_agents"][live, t], err_msg=where + " agents")
        if "post_flags" in d:
            np.testing.assert_array_equal(bo.flags[live], d["post_flags"][live, t], err_msg=where + " flags")
            np.testing.assert_array_equal(bo.meta[live], d["post_meta"][live, t], err_msg=where + " ball counters")
            np.testing.assert_array_equal(bo.info[live] & INFO_MASK, pack_info(d["info"][live, t]), err_msg=where + " info")
            # bits 6 / 7 of the info bytes (rewarded key / ball pickups) drive the mission texts of the next observations
            np.testing.assert_array_equal(np.asarray(missions)[live], d["mission"][live, t], err_msg=where + " missions")
            missions = [competitive_missions_after_step(missions[k], bo.info[k], d["order"][k, t], bo.agents[k]) for k in range(n)]


def team_outputs(members, reward, terminated, truncated):
    """ctce_step's reformatting (multigrid/envs/competative_red_blue_door.py:346-368): rewards summed, flags and-ed
    over each team's agents, in agent order."""
    r, tm, tr = [], [], []
    for team in members:
        total = reward[team[0]]
        for k in team[1:]:
            total = total + reward[k]
        r.append(float(total))
        tm.append(all(bool(terminated[k]) for k in team))
        tr.append(all(bool(truncated[k]) for k in team))
    return r, tm, tr


def test_ctce_outputs():
    """training_scheme "CTCE": no "some but not all terminated" reset (:375-378 belongs to dtde_step), outputs per team."""
    from oracle.coracle import BatchOracle
    spec, d = load_group("ctce_2v2")
    members = spec.pop("team_members")
    assert spec["per_agent_outputs"] is False
    a, n = spec["num_agents"], len(d["length"])
    bo = BatchOracle(spec, n)
    bo.load(np.arange(n), d["layout_grid"], d["layout_agents"], d["layout_meta"])
    envs_py = []
    for k in range(n):
        env = pyref.Env(make_spec(spec))
        env.load_layout(d["layout_grid"][k], d["layout_agents"][k], d["layout_meta"][k])
        env.missions = [MISSIONS[j] for j in d["reset_mission"][k]]
        envs_py.append(env)
    missions = d["reset_mission"].tolist()
    for t in range(int(d["length"].max())):
        img, direction, rew, term, trunc = bo.step(d["actions"][:, t], d["order"][:, t])
        for k in range(n):
            if d["length"][k] <= t:
                continue
            where = f"ctce_2v2[{k}] step {t}"
            o, r, tm, tr, info = envs_py[k].step({i: int(d["actions"][k, t, i]) for i in range(a)}, d["order"][k, t])
            want_info = pack_info(d["info"][k, t])
            np.testing.assert_array_equal(team_info(members, pack_info_dict(info, range(a))), want_info, err_msg=where + " info")
            np.testing.assert_array_equal(team_info(members, bo.info[k]), want_info, err_msg=where + " info (C)")
            assert envs_py[k].obs_missions == [MISSIONS[j] for j in d["mission"][k, t]], where + " missions"
            assert list(missions[k]) == d["mission"][k, t].tolist(), where + " missions (C)"
            missions[k] = competitive_missions_after_step(missions[k], bo.info[k], d["order"][k, t], bo.agents[k])
            for got in (team_outputs(members, r, tm, tr), team_outputs(members, rew[k], term[k], trunc[k])):
                assert got[0] == list(d["reward"][k, t]), where + " team rewards"
                assert got[1] == list(d["terminated"][k, t]), where + " team terminations"
                assert got[2] == list(d["truncated"][k, t]), where + " team truncations"
            np.testing.assert_array_equal(np.stack([o[i]["image"] for i in range(a)]), d["image"][k, t], err_msg=where)
            np.testing.assert_array_equal(img[k], d["image"][k, t], err_msg=where)
            np.testing.assert_array_equal(envs_py[k].grid, d["post_grid"][k, t], err_msg=where + " grid")
            np.testing.assert_array_equal(bo.grid[k], d["post_grid"][k, t], err_msg=where + " grid (C)")
            np.testing.assert_array_equal(envs_py[k].agents, d["post_agents"][k, t], err_msg=where + " agents")
            np.testing.assert_array_equal(bo.agents[k], d["post_agents"][k, t], err_msg=where + " agents (C)")
            assert envs_py[k].flags == bo.flags[k] == d["post_flags"][k, t], where + " flags"
