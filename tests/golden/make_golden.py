#!/usr/bin/env python3
"""Record golden trajectories from the UNMODIFIED reference implementation.

Runs only in the build container (needs ``/root/reference``); the GPU box and
the test-suite read the ``.npz`` files this script writes next to itself.

    python tests/golden/make_golden.py            # regenerate every group
    python tests/golden/make_golden.py NAME ...   # only the named groups ("layouts" for layouts.npz)

The reference needs numba / aenum / pygame / gymnasium / ray, none of which is
installed here; ``refshim/`` supplies name-only stand-ins (see its README) so
that the reference's own ``multigrid/{base,core,utils/obs,envs}`` code executes
as written.  What is recorded per trajectory:

* the start state (grid ``(W,H,3)``, agent rows, door metadata) - either the
  reference's own ``reset()`` layout or that layout after a documented harness
  edit (scattered objects, unlocked doors, carried keys ...), always followed
  by the reference's own ``gen_obs()``;
* per step: the actions, the agent order the reference drew
  (``np_random.random(A).argsort()``, multigrid/base.py:461), everything
  ``step()`` returned, and the post-step ``grid.state`` / ``agent_states``.
"""
from __future__ import annotations

import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, "/root/reference")

from multigrid.core.constants import Color, Type  # noqa: E402
from multigrid.core.world_object import (  # noqa: E402
    Ball, Box, Door, Floor, Goal, Key, Lava, Wall)
from multigrid.envs import CONFIGURATIONS  # noqa: E402
from multigrid.envs.blockedunlockpickup import BlockedUnlockPickupEnv  # noqa: E402
from multigrid.envs.competative_red_blue_door import CompetativeRedBlueDoorEnvV3  # noqa: E402
from multigrid.envs.empty import EmptyEnv  # noqa: E402
from multigrid.envs.locked_hallway import LockedHallwayEnv  # noqa: E402
from multigrid.envs.redbluedoors import RedBlueDoorsEnv  # noqa: E402

KIND = {EmptyEnv: 0, RedBlueDoorsEnv: 1, BlockedUnlockPickupEnv: 2, LockedHallwayEnv: 3}
TEAM_NAMES = ["red", "green", "blue", "purple", "yellow", "grey"]
COLORS = list(Color)


class OrderRecorder:
    """Wraps the env's ``np_random`` and remembers the last ``random(size)`` draw."""

    def __init__(self, gen):
        self._gen = gen
        self.last = None

    def random(self, size=None):
        self.last = self._gen.random(size=size)
        return self.last

    def __getattr__(self, name):
        return getattr(self._gen, name)


def teams_for(num_agents: int) -> dict:
    """Spread agents over teams the way a user of ``teams=`` would (base.py:185)."""
    if num_agents <= len(TEAM_NAMES):
        return {TEAM_NAMES[i]: 1 for i in range(num_agents)}
    base, extra = divmod(num_agents, len(TEAM_NAMES))
    return {TEAM_NAMES[i]: base + (1 if i < extra else 0) for i in range(len(TEAM_NAMES))}


def agent_rows(env) -> np.ndarray:
    s = np.asarray(env.agent_states).astype(np.int64)
    out = np.zeros((s.shape[0], 8), dtype=np.uint8)
    out[:, 0], out[:, 1] = s[:, 3], s[:, 4]
    out[:, 2], out[:, 3] = s[:, 2], s[:, 5]
    out[:, 4:7] = s[:, 6:9]
    out[:, 7] = s[:, 1]
    return out


def door_meta(env) -> np.ndarray:
    meta = np.zeros(4, dtype=np.uint8)
    if isinstance(env, RedBlueDoorsEnv):
        for (x, y), obj in env.grid.world_objects.items():
            if obj is env.red_door:
                meta[0], meta[1] = x, y
            if obj is env.blue_door:
                meta[2], meta[3] = x, y
    if isinstance(env, LockedHallwayEnv) and env.num_rooms > 6:
        # door colours repeat beyond six rooms (locked_hallway.py:150-151): the all-doors ending compares with
        # len(self.rooms), a dict keyed by colour (:160-166, :213)
        meta[3] = len(env.rooms)
    return meta


def spec_of(env) -> dict:
    return dict(
        kind=KIND[type(env)], width=env.width, height=env.height,
        num_agents=env.num_agents, max_steps=env.max_steps,
        view_size=env.agents[0].view_size,
        see_through_walls=bool(env.agents[0].see_through_walls),
        allow_agent_overlap=bool(env.allow_agent_overlap),
        joint_reward=bool(env.joint_reward),
        success_any=env.success_termination_mode == "any",
        failure_any=env.failure_termination_mode == "any",
        num_rooms=int(getattr(env, "num_rooms", 0)) if isinstance(env, LockedHallwayEnv) else 0,
        mission=str(env.mission) if isinstance(env.mission, str) else env.mission.string,
    )


# --- harness edits applied after reset() ("chaos" groups) ----------------------

def free_cells(env):
    pos = {tuple(p) for p in np.asarray(env.agent_states.pos)}
    return [(x, y) for x in range(1, env.width - 1) for y in range(1, env.height - 1)
            if env.grid.get(x, y) is None and (x, y) not in pos]


def scatter(env, rng, n, kinds):
    cells = free_cells(env)
    rng.shuffle(cells)
    for (x, y) in cells[:n]:
        k = kinds[rng.integers(len(kinds))]
        c = COLORS[rng.integers(len(COLORS))]
        if k == "goal":
            obj = Goal()
        elif k == "lava":
            obj = Lava()
        elif k == "floor":
            obj = Floor(c)
        elif k == "wall":
            obj = Wall()
        elif k == "key":
            obj = Key(c)
        elif k == "ball":
            obj = Ball(c)
        elif k == "box":
            obj = Box(c)
        elif k == "door_open":
            obj = Door(c, is_open=True)
        elif k == "door_closed":
            obj = Door(c)
        elif k == "door_locked":
            obj = Door(c, is_locked=True)
        env.grid.set(x, y, obj)


def scramble_agents(env, rng, carry=0.3, terminated=0.0):
    cells = free_cells(env)
    rng.shuffle(cells)
    for k, agent in enumerate(env.agents):
        if rng.random() < 0.85 and cells:
            agent.state.pos = cells.pop()
        agent.state.dir = int(rng.integers(4))
        if rng.random() < carry:
            kind = rng.integers(3)
            c = COLORS[rng.integers(len(COLORS))]
            agent.state.carrying = [Key, Ball, Box][kind](c)
        if rng.random() < terminated:
            agent.state.terminated = True


def unlock_doors(env, rng, p=1.0):
    for (x, y), obj in list(env.grid.world_objects.items()):
        if isinstance(obj, Door) and obj.is_locked and rng.random() < p:
            obj.is_locked = False
            if rng.random() < 0.3:
                obj.is_open = True
            env.grid.update(x, y)


def face_doors_with_keys(env, rng):
    """LockedHallway: stand agent k in front of a locked door holding its key."""
    doors = [((x, y), o) for (x, y), o in env.grid.world_objects.items() if isinstance(o, Door)]
    rng.shuffle(doors)
    for agent, ((x, y), door) in zip(env.agents, doors):
        for d, (dx, dy) in enumerate(((1, 0), (0, 1), (-1, 0), (0, -1))):
            sx, sy = x - dx, y - dy
            if 0 < sx < env.width - 1 and 0 < sy < env.height - 1 and env.grid.get(sx, sy) is None:
                agent.state.pos = (sx, sy)
                agent.state.dir = d
                if rng.random() < 0.8:
                    agent.state.carrying = Key(door.color)
                break


# --- a planner that plays BlockedUnlockPickup to the end -------------------------
#
# Agent 0 solves the task the way the mission reads, from an untouched reset(): take the ball away from the door,
# fetch the key, unlock the door, put the key down, walk through and pick up the box (blockedunlockpickup.py:136-167).
# Every step it looks at the live reference environment, decides the current sub-goal and takes the first action of
# a breadth-first path to it.  The other agents only turn or idle.  Nothing of the planner is reference code: it only
# reads the reference's objects.

DIRS = ((1, 0), (0, 1), (-1, 0), (0, -1))


def _walkable(env, x, y):
    if not (0 <= x < env.width and 0 <= y < env.height):
        return False
    obj = env.grid.get(x, y)
    return obj is None or obj.can_overlap()


def _path_to(env, start, goals):
    """First action of a shortest left / right / forward path from pose `start` to any pose in `goals`."""
    from collections import deque
    if start in goals:
        return None
    seen, queue = {start: None}, deque([start])
    while queue:
        x, y, d = cur = queue.popleft()
        nxt = [((x, y, (d - 1) % 4), 0), ((x, y, (d + 1) % 4), 1)]
        fx, fy = x + DIRS[d][0], y + DIRS[d][1]
        if _walkable(env, fx, fy):
            nxt.append(((fx, fy, d), 2))
        for pose, act in nxt:
            if pose in seen:
                continue
            seen[pose] = (cur, act)
            if pose in goals:
                while seen[pose][0] != start:
                    pose = seen[pose][0]
                return seen[pose][1]
            queue.append(pose)
    return 6  # unreachable: idle


def _poses_facing(env, cell):
    cx, cy = cell
    return {(cx - dx, cy - dy, d) for d, (dx, dy) in enumerate(DIRS) if _walkable(env, cx - dx, cy - dy)}


def solve_blockedunlockpickup(env, rng, t):
    ag = env.agents[0].state
    x, y = (int(v) for v in ag.pos)
    pose = (x, y, int(ag.dir))
    others = {tuple(int(v) for v in a.state.pos) for a in env.agents[1:]}
    objs = dict(env.grid.world_objects)
    door_pos, door = next((pos, o) for pos, o in objs.items() if isinstance(o, Door) and env.grid.get(*pos) is o)
    blocker = (door_pos[0] - 1, door_pos[1])  # blockedunlockpickup.py: the ball lies in front of the door
    carrying = ag.carrying

    def interact(cell, action):
        step = _path_to(env, pose, _poses_facing(env, cell))
        return action if step is None else step

    def put_down():
        # any free cell that is not the doorway's approach and holds no agent
        goals = set()
        for cx in range(1, env.width - 1):
            for cy in range(1, env.height - 1):
                if env.grid.get(cx, cy) is None and (cx, cy) != blocker and (cx, cy) not in others and (cx, cy) != (x, y):
                    goals |= _poses_facing(env, (cx, cy))
        step = _path_to(env, pose, goals)
        return 4 if step is None else step

    if door.is_locked:
        ball_there = isinstance(env.grid.get(*blocker), Ball)
        if isinstance(carrying, Ball) or (carrying is not None and ball_there):
            a0 = put_down()
        elif ball_there:
            a0 = interact(blocker, 3)
        elif isinstance(carrying, Key):
            a0 = interact(door_pos, 5)
        else:
            key_pos = next(pos for pos, o in objs.items() if isinstance(o, Key) and env.grid.get(*pos) is o)
            a0 = interact(key_pos, 3)
    elif not door.is_open:
        a0 = interact(door_pos, 5)
    elif carrying is not None and carrying is not env.obj:
        a0 = put_down()
    else:
        box_pos = next((pos for pos, o in objs.items() if o is env.obj and env.grid.get(*pos) is o), None)
        a0 = interact(box_pos, 3) if box_pos is not None else 6  # the box is in hand: the episode is over
    rest = [int(rng.choice([0, 1, 6])) for _ in env.agents[1:]]
    return [a0] + rest


def solve_lockedhallway(env, rng, t):
    """Agent 0 unlocks every door it can: fetch a reachable key whose door is still locked, carry it to that door,
    toggle (locked_hallway.py:200-215 pays once per door and ends the episode when all are unlocked); keys found
    behind the doors open the next ones.  A key without a use is put down."""
    ag = env.agents[0].state
    x, y = (int(v) for v in ag.pos)
    pose = (x, y, int(ag.dir))
    others = {tuple(int(v) for v in a.state.pos) for a in env.agents[1:]}
    objs = {pos: o for pos, o in env.grid.world_objects.items() if env.grid.get(*pos) is o}
    locked = {o.color: pos for pos, o in objs.items() if isinstance(o, Door) and o.is_locked}
    rest = [int(rng.choice([0, 1, 6])) for _ in env.agents[1:]]

    def towards(cell, action):  # (reachable, action to take now)
        step = _path_to(env, pose, _poses_facing(env, cell))
        if step is None:
            return True, action
        return step != 6, step

    carrying = ag.carrying
    if isinstance(carrying, Key) and carrying.color in locked:
        ok, a0 = towards(locked[carrying.color], 5)
        if ok:
            return [a0] + rest
    if carrying is not None:  # nothing to do with it (now): put it on a free cell
        goals = set()
        for cx in range(1, env.width - 1):
            for cy in range(1, env.height - 1):
                if env.grid.get(cx, cy) is None and (cx, cy) not in others and (cx, cy) != (x, y):
                    goals |= _poses_facing(env, (cx, cy))
        step = _path_to(env, pose, goals)
        return [4 if step is None else step] + rest
    for pos, o in sorted(objs.items()):
        if isinstance(o, Key) and o.color in locked:
            ok, a0 = towards(pos, 3)
            if ok:
                return [a0] + rest
    return [6] + rest


# --- recording ---------------------------------------------------------------

ONLY = set(sys.argv[1:])  # group names to (re)generate; empty = all


def record_group(name, build_env, n_traj, t_max, seed0, edit=None, action_p=None,
                 extra_after_done=3, script=None, policy=None):
    if ONLY and name not in ONLY:
        return
    rng = np.random.default_rng(seed0)
    trajs = []
    spec = None
    for n in range(n_traj):
        env = build_env()
        env._RandomMixin__np_random = np.random.default_rng(seed0 * 1000 + n)
        obs, _ = env.reset(seed=seed0 * 7919 + n)
        if edit is not None:
            edit(env, rng, n)
            obs = env.gen_obs()
        rec = OrderRecorder(env._np_random)
        env._np_random = rec
        if spec is None:
            spec = spec_of(env)
        a = env.num_agents
        v = env.agents[0].view_size
        tr = dict(
            layout_grid=env.grid.state.astype(np.uint8).copy(),
            layout_agents=agent_rows(env), layout_meta=door_meta(env),
            reset_image=np.stack([obs[i]["image"] for i in range(a)]).astype(np.uint8),
            actions=[], order=[], image=[], direction=[], reward=[], terminated=[],
            truncated=[], post_grid=[], post_agents=[])
        after = 0
        for t in range(t_max):
            if script is not None and t < len(script(n)):
                acts = np.asarray(script(n)[t][:a], dtype=np.int64)
            elif policy is not None:  # actions chosen by looking at the live reference environment
                acts = np.asarray(policy(env, rng, t)[:a], dtype=np.int64)
            elif action_p is None:
                acts = rng.integers(0, 7, size=a)
            else:
                acts = rng.choice(7, size=a, p=action_p)
            rec.last = None
            o, r, term, trunc, _ = env.step({i: int(acts[i]) for i in range(a)})
            order = np.arange(a) if rec.last is None else np.argsort(rec.last)
            tr["actions"].append(acts.astype(np.uint8))
            tr["order"].append(order.astype(np.uint8))
            img = np.stack([o[i]["image"] for i in range(a)])
            assert img.min() >= 0 and img.max() < 256 and img.shape == (a, v, v, 3)
            tr["image"].append(img.astype(np.uint8))
            tr["direction"].append(np.array([int(o[i]["direction"]) for i in range(a)], np.uint8))
            tr["reward"].append(np.array([float(r[i]) for i in range(a)], np.float64))
            tr["terminated"].append(np.array([bool(term[i]) for i in range(a)]))
            tr["truncated"].append(np.array([bool(trunc[i]) for i in range(a)]))
            tr["post_grid"].append(env.grid.state.astype(np.uint8).copy())
            tr["post_agents"].append(agent_rows(env))
            if all(term.values()) or all(trunc.values()):
                after += 1
                if after > extra_after_done:
                    break
        trajs.append(tr)

    n = len(trajs)
    lens = np.array([len(t["actions"]) for t in trajs], np.int32)
    tm = int(lens.max())
    out = dict(spec=np.array(json.dumps(spec)), length=lens)
    for key in ("layout_grid", "layout_agents", "layout_meta", "reset_image"):
        out[key] = np.stack([t[key] for t in trajs])
    for key in ("actions", "order", "image", "direction", "reward", "terminated",
                "truncated", "post_grid", "post_agents"):
        first = trajs[0][key][0]
        arr = np.zeros((n, tm) + first.shape, dtype=first.dtype)
        for i, t in enumerate(trajs):
            arr[i, :len(t[key])] = np.stack(t[key])
        out[key] = arr
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    nrew = int((out["reward"] != 0).sum())
    nterm = int(out["terminated"].any(axis=(1, 2)).sum())
    print(f"{name:28s} traj={n:3d} steps={int(lens.sum()):6d} nonzero_rewards={nrew:5d} "
          f"traj_with_termination={nterm:3d} size={os.path.getsize(path) / 1024:.0f} KiB")


def record_layouts():
    """Reference ``reset()`` layouts for fixed seeds (checks the host-side layout generators)."""
    out = {}
    for name, agents in (("MultiGrid-Empty-8x8-v0", 2), ("MultiGrid-RedBlueDoors-8x8-v0", 2),
                         ("MultiGrid-BlockedUnlockPickup-v0", 2),
                         ("MultiGrid-LockedHallway-4Rooms-v0", 4),
                         ("MultiGrid-LockedHallway-2Rooms-v0", 2),
                         ("MultiGrid-LockedHallway-6Rooms-v0", 3),
                         ("MultiGrid-Empty-16x16-v0", 8), ("MultiGrid-RedBlueDoors-6x6-v0", 3),
                         ("MultiGrid-Empty-Random-6x6-v0", 3),
                         ("MultiGrid-CompetativeRedBlueDoor-v3-DTDE-1v1", 2),
                         ("MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2", 4),
                         ("MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2-Death-Match", 4),
                         ("MultiGrid-CompetativeRedBlueDoor-v3-DTDE-Red-Single-with-Obstacle", 1)):
        cls, kw = CONFIGURATIONS[name]
        grids, ags, metas, missions = [], [], [], []
        for seed in range(24):
            env = cls(policies_map={}, **kw) if cls is CompetativeRedBlueDoorEnvV3 else make_env(cls, agents, **kw)
            env._RandomMixin__np_random = np.random.default_rng(seed)
            env.reset(seed=seed + 100)
            grids.append(env.grid.state.astype(np.uint8).copy())
            ags.append(agent_rows(env))
            metas.append(door_meta(env))
            m = env.mission
            missions.append(m if isinstance(m, str) else m.string)
        key = name.replace("MultiGrid-", "").replace("-", "_")
        out[key + "__grid"] = np.stack(grids)
        out[key + "__agents"] = np.stack(ags)
        out[key + "__meta"] = np.stack(metas)
        out[key + "__mission"] = np.array(missions)
        out[key + "__num_agents"] = np.array(agents)
    path = os.path.join(HERE, "layouts.npz")
    np.savez_compressed(path, **out)
    print(f"layouts.npz size={os.path.getsize(path) / 1024:.0f} KiB")


# --- CompetativeRedBlueDoorEnvV3 (multigrid/envs/competative_red_blue_door.py) ------------------------------

SCHEME_KEYS = ("eliminated_opponent_sparse_reward", "key_pickup_sparse_reward", "ball_pickup_dense_reward",
               "invalid_pickup_dense_penalty")


def comp_hidden(env, keys, balls):
    """State the arrays do not show: a key whose OBJECT is "picked up" while the agent row still carries the copy
    taken at pickup time (competative_red_blue_door.py:434-435), a door object whose cell was cleared under an
    eliminated agent (:503), and how often each ball's discount factor has been squared (:454)."""
    flags = 0
    for bit, color in enumerate(("red", "blue")):
        k = keys.get(color)
        if k is not None and int(k.state) == 3:
            flags |= 1 << bit
    for bit, door in enumerate((env.red_door, env.blue_door)):
        pos = [xy for xy, o in env.grid.world_objects.items() if o is door]
        if not pos:
            flags |= 4 << bit
    meta = np.zeros(4, np.uint8)
    for slot, color in enumerate(("red", "blue")):
        b = balls.get(color)
        if b is not None:
            n, d = 0, 0.9
            while d != b.discount_factor:
                d *= d
                n += 1
                assert n < 40
            meta[slot] = n
    return flags, meta


INFO_KEYS = ("door_open_done", "got_eliminated_done", "eliminated_opponents_done", "eliminated_opponent_num")


def record_competitive(name, kw, n_traj, t_max, seed0, edit=None, action_p=None, script=None, extra_after_done=3,
                       view=None, ctce=False):
    """Trajectories of CompetativeRedBlueDoorEnvV3 with its default step handlers (policies_map = {}).  The order in
    which agents act is the order of the keys of the actions dict (:633-641, :390); the harness shuffles it every step
    and records it."""
    if ONLY and name not in ONLY:
        return
    rng = np.random.default_rng(seed0)
    trajs, spec = [], None
    for n in range(n_traj):
        kwargs = dict(kw)
        if view is not None:
            kwargs["agent_view_size"] = view
        env = CompetativeRedBlueDoorEnvV3(policies_map={}, **kwargs)
        env._RandomMixin__np_random = np.random.default_rng(seed0 * 1000 + n)
        obs, _ = env.reset(seed=seed0 * 7919 + n)
        a = env.num_agents
        names = [ag.name for ag in env.agents]
        teams = list(env.teams)  # ctce_step keys everything by team (:319-370); kept here in the order of `teams`
        members = {tm: [env.team_index_dict[tm][j] for j in range(env.teams[tm])] for tm in teams}

        def per_agent(d):  # observations of a step, by agent index
            if ctce:
                flat = [None] * a
                for tm in teams:
                    for j, k in enumerate(members[tm]):
                        assert d[tm][j]["agent_id"] == j
                        flat[k] = d[tm][j]
                return flat
            return [d[nm] for nm in names]
        keys = {str(o.color.value if hasattr(o.color, "value") else o.color): o
                for o in env.grid.world_objects.values() if isinstance(o, Key)}
        balls = {str(o.color.value if hasattr(o.color, "value") else o.color): o
                 for o in env.grid.world_objects.values() if isinstance(o, Ball)}
        if edit is not None:
            edit(env, rng, n, keys, balls)
            obs = env.gen_obs()
        obs = per_agent(obs)
        if spec is None:
            spec = dict(
                kind=4, width=env.width, height=env.height, num_agents=a, max_steps=env.max_steps,
                view_size=env.agents[0].view_size, see_through_walls=bool(env.agents[0].see_through_walls),
                allow_agent_overlap=bool(env.allow_agent_overlap), joint_reward=bool(env.joint_reward),
                success_any=env.success_termination_mode == "any", failure_any=env.failure_termination_mode == "any",
                num_rooms=0, mission="", death_match=bool(env.death_match),
                per_agent_outputs=env.training_scheme != "CTCE",
                reward_schemes=[[float(env.reward_schemes[nm][k]) for k in SCHEME_KEYS] for nm in names],
                agent_names=names)
            if ctce:
                spec["team_members"] = [members[tm] for tm in teams]
        v = env.agents[0].view_size
        tr = dict(
            layout_grid=env.grid.state.astype(np.uint8).copy(), layout_agents=agent_rows(env),
            layout_meta=np.zeros(4, np.uint8),
            reset_image=np.stack([o["image"] for o in obs]).astype(np.uint8),
            reset_mission=np.array([mission_index(o["mission"]) for o in obs], np.uint8),
            actions=[], order=[], image=[], direction=[], reward=[], terminated=[], truncated=[], post_grid=[],
            post_agents=[], post_flags=[], post_meta=[], info=[], mission=[])
        after = 0
        for t in range(t_max):
            if script is not None and t < len(script(n)):
                acts = np.asarray(script(n)[t][:a], dtype=np.int64)
            elif action_p is None:
                acts = rng.integers(0, 7, size=a)
            else:
                acts = rng.choice(7, size=a, p=action_p)
            if ctce:  # one list of actions per team; teams, and so their agents, act in the order of the dict
                team_order = [teams[j] for j in rng.permutation(len(teams))]
                order = np.array([k for tm in team_order for k in members[tm]])
                o, r, term, trunc, inf = env.step({tm: [int(acts[k]) for k in members[tm]] for tm in team_order})
                out_keys = teams   # rewards summed, flags and-ed per team (:346-368)
            else:
                order = rng.permutation(a)
                o, r, term, trunc, inf = env.step({names[i]: int(acts[i]) for i in order})
                out_keys = names
            o = per_agent(o)
            tr["actions"].append(acts.astype(np.uint8))
            tr["order"].append(order.astype(np.uint8))
            img = np.stack([x["image"] for x in o])
            assert img.min() >= 0 and img.max() < 256 and img.shape == (a, v, v, 3)
            tr["image"].append(img.astype(np.uint8))
            tr["direction"].append(np.array([int(x["direction"]) for x in o], np.uint8))
            # the mission texts the observations carry (:307, :315; rewritten on pickups, :438-468), by MISSION_TEXTS index
            tr["mission"].append(np.array([mission_index(x["mission"]) for x in o], np.uint8))
            tr["reward"].append(np.array([float(r[nm]) for nm in out_keys], np.float64))
            tr["terminated"].append(np.array([bool(term[nm]) for nm in out_keys]))
            tr["truncated"].append(np.array([bool(trunc[nm]) for nm in out_keys]))
            # the info dict of the step (:320-328, :380-388): per agent name, or per team under CTCE
            assert set(inf) == set(out_keys) and all(set(inf[nm]) == set(INFO_KEYS) for nm in out_keys)
            tr["info"].append(np.array([[int(inf[nm][k]) for k in INFO_KEYS] for nm in out_keys], np.uint8))
            tr["post_grid"].append(env.grid.state.astype(np.uint8).copy())
            tr["post_agents"].append(agent_rows(env))
            fl, mt = comp_hidden(env, keys, balls)
            tr["post_flags"].append(np.uint8(fl))
            tr["post_meta"].append(mt)
            if all(term.values()) or all(trunc.values()):
                after += 1
                if after > extra_after_done:
                    break
        trajs.append(tr)
    n = len(trajs)
    lens = np.array([len(t["actions"]) for t in trajs], np.int32)
    tm = int(lens.max())
    out = dict(spec=np.array(json.dumps(spec)), length=lens)
    for key in ("layout_grid", "layout_agents", "layout_meta", "reset_image", "reset_mission"):
        out[key] = np.stack([t[key] for t in trajs])
    for key in ("actions", "order", "image", "direction", "reward", "terminated", "truncated", "post_grid",
                "post_agents", "post_flags", "post_meta", "info", "mission"):
        first = np.asarray(trajs[0][key][0])
        arr = np.zeros((n, tm) + first.shape, dtype=first.dtype)
        for i, t in enumerate(trajs):
            arr[i, :len(t[key])] = np.stack(t[key])
        out[key] = arr
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"info events {out['info'].reshape(-1, 4).sum(axis=0).tolist()} missions {np.bincount(out['mission'].ravel(), minlength=4).tolist()}", end="  ")
    print(f"{name:28s} traj={n:3d} steps={int(lens.sum()):6d} nonzero_rewards={int((out['reward'] != 0).sum()):5d} "
          f"positive={int((out['reward'] > 0).sum()):4d} below_minus_half={int((out['reward'] < -0.5).sum()):3d} "
          f"traj_with_termination={int(out['terminated'].any(axis=(1, 2)).sum()):3d} "
          f"flags_seen={sorted(set(out['post_flags'].ravel().tolist()))} ball_max={int(out['post_meta'].max())} "
          f"size={os.path.getsize(path) / 1024:.0f} KiB")


# the texts of the environment's mission space (:158-168), in its order; every text an observation carries is one of them
MISSION_TEXTS = ("Go pick up the key or the ball for opening the door", "Go pick up the key", "Go move away the ball",
                 "Go open the door with the key")


def mission_index(m) -> int:
    return MISSION_TEXTS.index(m if isinstance(m, str) else m.string)


def comp_free_cells(env):
    """Empty cells of the middle room that no agent stands on."""
    pos = {tuple(int(c) for c in ag.state.pos) for ag in env.agents}
    x0, x1 = env.width // 4 + 1, env.width // 4 + env.width // 2 - 1
    return [(x, y) for x in range(x0, x1) for y in range(1, env.height - 1)
            if env.grid.get(x, y) is None and (x, y) not in pos]


def comp_take(env, agent, obj):
    """Hand `obj` (one of the environment's own objects, wherever it lies) to `agent`."""
    for xy, o in list(env.grid.world_objects.items()):
        if o is obj:
            env.grid.set(*xy, None)
    agent.state.carrying = obj


def comp_edit(env, rng, n, keys, balls):
    cells = comp_free_cells(env)
    rng.shuffle(cells)
    by_color = {}
    for ag in env.agents:
        by_color.setdefault(str(ag.color.value if hasattr(ag.color, "value") else ag.color), []).append(ag)
    rx, ry, bx, by = env.width // 4, env.height - 2, env.width // 4 + env.width // 2 - 1, 1
    mode = n % 8
    if mode == 0 and "red" in by_color:       # red agent with the red key in front of the red door
        ag = by_color["red"][-1]
        comp_take(env, ag, keys["red"])
        if env.grid.get(rx + 1, ry) is not None:
            env.grid.set(rx + 1, ry, None)
        ag.state.pos, ag.state.dir = (rx + 1, ry), 2
    elif mode == 1 and "blue" in by_color:    # blue agent with the blue key in front of the blue door
        ag = by_color["blue"][0]
        comp_take(env, ag, keys["blue"])
        if env.grid.get(bx - 1, by) is not None:
            env.grid.set(bx - 1, by, None)
        ag.state.pos, ag.state.dir = (bx - 1, by), 0
    elif mode in (2, 3) and len(by_color) == 2:   # opponents next to each other, facing each other
        for r_ag, b_ag in zip(by_color["red"], by_color["blue"]):
            for (x, y) in cells:
                if (x + 1, y) in cells:
                    cells.remove((x, y))
                    cells.remove((x + 1, y))
                    r_ag.state.pos, r_ag.state.dir = (x, y), 0
                    b_ag.state.pos, b_ag.state.dir = (x + 1, y), 2
                    break
        if mode == 3:   # doors that can be walked through: an agent may be eliminated in a doorway
            for door, (dx, dy) in ((env.red_door, (rx, ry)), (env.blue_door, (bx, by))):
                door.is_locked = False
                door.is_open = True
                env.grid.update(dx, dy)
            ag = by_color["blue"][0]
            ag.state.pos, ag.state.dir = (rx, ry), 0
            opp = by_color["red"][0]
            if env.grid.get(rx + 1, ry) is not None:
                env.grid.set(rx + 1, ry, None)
            opp.state.pos, opp.state.dir = (rx + 1, ry), 2
    elif mode == 4:   # every agent next to its own key or the other team's, facing it
        for ag in env.agents:
            k = list(keys.values())[int(rng.integers(len(keys)))]
            kp = [xy for xy, o in env.grid.world_objects.items() if o is k]
            if kp and (kp[0][0] - 1, kp[0][1]) in cells:
                cells.remove((kp[0][0] - 1, kp[0][1]))
                ag.state.pos, ag.state.dir = (kp[0][0] - 1, kp[0][1]), 0
    elif mode == 5 and balls:   # agents in front of the balls
        for color, ball in balls.items():
            opp = "red" if color == "blue" else "blue"
            if opp in by_color:
                ag = by_color[opp][0]
                ix, iy = ball.init_pos
                ag.state.pos, ag.state.dir = ((ix + 1, iy), 2) if color == "blue" else ((ix - 1, iy), 0)
    elif mode == 6:   # scattered, some already holding things
        for ag in env.agents:
            if cells:
                ag.state.pos = cells.pop()
            ag.state.dir = int(rng.integers(4))
        for color, k in keys.items():
            if color in by_color and rng.random() < 0.7:
                comp_take(env, by_color[color][0], k)
    # mode 7: untouched


PICKY = np.array([0.10, 0.10, 0.22, 0.26, 0.10, 0.20, 0.02])


BIASED = np.array([0.12, 0.12, 0.36, 0.12, 0.08, 0.18, 0.02])
TOGGLY = np.array([0.10, 0.10, 0.25, 0.05, 0.05, 0.43, 0.02])


def main():
    cfg = CONFIGURATIONS
    # ---- the five BASELINE configurations, untouched reference layouts, uniform actions
    record_group("cfg1_empty8", lambda: make_env(*_ck(cfg["MultiGrid-Empty-8x8-v0"], 2)), 6, 262, 11)
    record_group("cfg2_redbluedoors8", lambda: make_env(*_ck(cfg["MultiGrid-RedBlueDoors-8x8-v0"], 2)), 8, 120, 12)
    record_group("cfg3_blockedunlockpickup", lambda: make_env(*_ck(cfg["MultiGrid-BlockedUnlockPickup-v0"], 2)), 10, 150, 13,
                 action_p=None)
    record_group("cfg3_blockedunlockpickup_biased", lambda: make_env(*_ck(cfg["MultiGrid-BlockedUnlockPickup-v0"], 2)), 10, 200, 14,
                 action_p=BIASED)
    record_group("cfg4_lockedhallway4", lambda: make_env(*_ck(cfg["MultiGrid-LockedHallway-4Rooms-v0"], 4)), 6, 150, 15)
    record_group("cfg4_lockedhallway4_biased", lambda: make_env(*_ck(cfg["MultiGrid-LockedHallway-4Rooms-v0"], 4)), 6, 250, 16,
                 action_p=BIASED)
    record_group("cfg5_empty16", lambda: make_env(*_ck(cfg["MultiGrid-Empty-16x16-v0"], 8)), 4, 120, 17)
    record_group("cfg5_empty16_nooverlap",
                 lambda: make_env(*_ck(cfg["MultiGrid-Empty-16x16-v0"], 8), allow_agent_overlap=False), 4, 120, 18,
                 action_p=BIASED)
    # short episodes: truncation, rewards late in the episode
    record_group("short_empty8", lambda: make_env(EmptyEnv, 2, size=8, max_steps=24), 4, 30, 19, action_p=BIASED)
    record_group("short_lockedhallway2", lambda: make_env(LockedHallwayEnv, 2, num_rooms=2, max_steps=30), 4, 36, 20,
                 action_p=BIASED)

    # the registry entry constructed with the reference's own defaults (agent_view_size 5, base.py:96)
    # registered configurations seen through 9 x 9 and 11 x 11 views, from untouched resets (replayed by the facade from seeds)
    record_group("seeded_view9_lockedhallway2",
                 lambda: make_env(*_ck(cfg["MultiGrid-LockedHallway-2Rooms-v0"], 3), view=9), 3, 60, 57, action_p=BIASED)
    record_group("seeded_view11_blockedunlockpickup",
                 lambda: make_env(*_ck(cfg["MultiGrid-BlockedUnlockPickup-v0"], 2), view=11), 3, 60, 58, action_p=BIASED)
    record_group("default_empty8",
                 lambda: EmptyEnv(agents=2, teams=teams_for(2), policies_map={}, **cfg["MultiGrid-Empty-8x8-v0"][1]),
                 3, 40, 21)

    # ---- harness-edited start states
    allk = ["goal", "lava", "floor", "wall", "key", "ball", "box", "door_open", "door_closed", "door_locked"]

    def chaos_empty(flags):
        def build():
            env = make_env(EmptyEnv, flags["agents"], view=flags.get("view", 7), size=flags.get("size", 9),
                           max_steps=flags.get("max_steps", 60),
                           allow_agent_overlap=flags["overlap"], joint_reward=flags["joint"],
                           success_termination_mode=flags["succ"], failure_termination_mode=flags["fail"])
            for ag in env.agents:
                ag.see_through_walls = flags["see"]
            return env

        def edit(env, rng, n):
            scatter(env, rng, flags.get("objects", 16), allk)
            scramble_agents(env, rng, carry=0.35, terminated=flags.get("terminated", 0.0))
        return build, edit

    for name, flags in (
        ("chaos_empty_a", dict(agents=3, overlap=False, joint=False, succ="all", fail="all", see=False)),
        ("chaos_empty_b", dict(agents=3, overlap=True, joint=True, succ="any", fail="any", see=False)),
        ("chaos_empty_c", dict(agents=4, overlap=False, joint=True, succ="all", fail="any", see=True,
                               terminated=0.25)),
        ("chaos_empty_d", dict(agents=2, overlap=True, joint=False, succ="any", fail="all", see=False,
                               size=7, objects=9)),
        ("chaos_single", dict(agents=1, overlap=True, joint=False, succ="all", fail="all", see=False)),
        ("chaos_view5", dict(agents=3, overlap=False, joint=False, succ="all", fail="all", see=False, view=5)),
        ("chaos_view3", dict(agents=2, overlap=True, joint=False, succ="any", fail="all", see=False, view=3)),
        # views wider than any registered configuration's (the reference takes any odd size, base.py:96)
        ("chaos_view9", dict(agents=3, overlap=False, joint=False, succ="all", fail="all", see=False, view=9,
                             size=12, objects=30)),
        ("chaos_view11", dict(agents=2, overlap=True, joint=True, succ="any", fail="any", see=False, view=11,
                              size=14, objects=40)),
        ("chaos_view9_see", dict(agents=4, overlap=False, joint=False, succ="all", fail="any", see=True, view=9,
                                 size=8, objects=12, terminated=0.2)),
        ("chaos_view11_see", dict(agents=7, overlap=False, joint=False, succ="all", fail="all", see=True, view=11,
                                  size=10, objects=20)),
        ("chaos_wide", dict(agents=6, overlap=False, joint=False, succ="all", fail="all", see=False,
                            size=12, objects=30, terminated=0.15)),
    ):
        build, edit = chaos_empty(flags)
        record_group(name, build, 8, 70, zlib.crc32(name.encode()) % 1000 + 30, edit=edit, action_p=BIASED)

    # RedBlueDoors with usable doors (the fork creates them locked with no key in the world)
    def rbd_edit(env, rng, n):
        unlock_doors(env, rng)
        scramble_agents(env, rng, carry=0.0)
        if n % 2 == 0:  # start next to a door, facing it
            door = env.blue_door if n % 4 == 0 else env.red_door
            for (x, y), o in env.grid.world_objects.items():
                if o is door:
                    env.agents[0].state.pos = (x - 1, y) if o is env.blue_door else (x + 1, y)
                    env.agents[0].state.dir = 0 if o is env.blue_door else 2
    record_group("chaos_redbluedoors", lambda: make_env(RedBlueDoorsEnv, 2, size=8, max_steps=80),
                 16, 60, 41, edit=rbd_edit, action_p=TOGGLY)
    record_group("chaos_redbluedoors_all",
                 lambda: make_env(RedBlueDoorsEnv, 3, size=6, max_steps=80,
                                  success_termination_mode="all", failure_termination_mode="all",
                                  joint_reward=False),
                 16, 70, 42, edit=rbd_edit, action_p=TOGGLY)

    def bup_edit(env, rng, n):
        scatter(env, rng, 3, ["key", "ball", "floor", "door_open"])
        if n % 3 == 0:
            unlock_doors(env, rng)
        if n % 2 == 0:
            for (x, y), o in env.grid.world_objects.items():
                if isinstance(o, Door):
                    env.agents[0].state.carrying = Key(o.color)
        scramble_agents(env, rng, carry=0.0)
        if n % 4 == 1:  # stand next to the target box
            for (x, y), o in env.grid.world_objects.items():
                if o is env.obj and env.grid.get(x - 1, y) is None:
                    env.agents[1].state.pos = (x - 1, y)
                    env.agents[1].state.dir = 0
    record_group("chaos_blockedunlockpickup", lambda: make_env(BlockedUnlockPickupEnv, 2, max_steps=90),
                 16, 80, 43, edit=bup_edit, action_p=BIASED)
    # (opaque walls seen through 9 x 9 and 11 x 11 views: the rooms' walls and doors shadow most of such a view)
    record_group("chaos_blockedunlockpickup_view9", lambda: make_env(BlockedUnlockPickupEnv, 2, max_steps=90, view=9),
                 6, 60, 51, edit=bup_edit, action_p=BIASED)
    # the task played to the end from untouched resets (2 and 3 agents, joint reward on and off)
    record_group("solve_blockedunlockpickup", lambda: make_env(*_ck(cfg["MultiGrid-BlockedUnlockPickup-v0"], 2)),
                 8, 120, 47, policy=solve_blockedunlockpickup)
    record_group("solve_blockedunlockpickup3", lambda: make_env(BlockedUnlockPickupEnv, 3, joint_reward=True),
                 4, 120, 48, policy=solve_blockedunlockpickup)
    # LockedHallway with every door unlocked in turn (rewards per door, the all-doors ending)
    record_group("solve_lockedhallway2", lambda: make_env(*_ck(cfg["MultiGrid-LockedHallway-2Rooms-v0"], 2)),
                 4, 200, 49, policy=solve_lockedhallway)
    record_group("solve_lockedhallway4", lambda: make_env(*_ck(cfg["MultiGrid-LockedHallway-4Rooms-v0"], 4)),
                 4, 400, 50, policy=solve_lockedhallway)

    def lh_edit(env, rng, n):
        if n % 2 == 0:
            face_doors_with_keys(env, rng)
        else:
            scramble_agents(env, rng, carry=0.0)
            for ag in env.agents:
                if rng.random() < 0.7:
                    ag.state.carrying = Key(COLORS[rng.integers(6)])

    def lh_script(n):
        return [[5] * 8] if n % 4 == 0 else []
    record_group("chaos_lockedhallway4", lambda: make_env(LockedHallwayEnv, 4, num_rooms=4, max_steps=120),
                 12, 80, 44, edit=lh_edit, action_p=TOGGLY, script=lh_script)
    record_group("chaos_lockedhallway2", lambda: make_env(LockedHallwayEnv, 2, num_rooms=2, max_steps=60,
                                                           joint_reward=False),
                 12, 50, 45, edit=lh_edit, action_p=TOGGLY, script=lh_script)
    record_group("chaos_lockedhallway2_view11", lambda: make_env(LockedHallwayEnv, 3, num_rooms=2, max_steps=60, view=11),
                 6, 50, 52, edit=lh_edit, action_p=TOGGLY, script=lh_script)
    # more rooms than door colours: doors of one colour are separate doors, the ending comes at len(self.rooms)
    record_group("chaos_lockedhallway8", lambda: make_env(LockedHallwayEnv, 4, num_rooms=8, room_size=4, max_steps=120),
                 10, 70, 53, edit=lh_edit, action_p=TOGGLY, script=lh_script)
    record_group("chaos_lockedhallway10", lambda: make_env(LockedHallwayEnv, 3, num_rooms=10, room_size=4, max_steps=90,
                                                            joint_reward=False),
                 6, 60, 54, edit=lh_edit, action_p=TOGGLY, script=lh_script)
    record_group("solve_lockedhallway8", lambda: make_env(LockedHallwayEnv, 2, num_rooms=8, room_size=4, max_steps=2000),
                 4, 700, 55, policy=solve_lockedhallway, extra_after_done=400)
    record_group("chaos_lockedhallway6", lambda: make_env(LockedHallwayEnv, 6, num_rooms=6, max_steps=100),
                 6, 60, 46, edit=lh_edit, action_p=TOGGLY, script=lh_script)
    # ---- CompetativeRedBlueDoorEnvV3, registry entries as they are and with harness-edited starts
    def comp_script(n):
        m = n % 8
        if m in (0, 1):
            return [[5] * 4, [5] * 4]     # toggle: the door opens and the team wins; then toggling goes on
        if m in (2, 3):
            return [[5] * 4]               # everybody toggles: eliminations
        if m in (4, 5):
            return [[3] * 4, [3] * 4, [3] * 4, [4] * 4, [3] * 4, [3] * 4]   # pick up, again, drop, pick up again
        return []
    for gname, reg, view, kwx, nt in (
        ("comp_1v1", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-1v1", None, {}, 16),
        ("comp_2v2", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2", 7, {"max_steps": 120}, 16),
        ("comp_2v2_deathmatch", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2-Death-Match", None, {"max_steps": 90}, 16),
        ("comp_2v2_all", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2", 7,
         {"max_steps": 100, "success_termination_mode": "all", "allow_agent_overlap": True}, 16),
        ("comp_single", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-Red-Single-with-Obstacle", 7, {"max_steps": 60}, 8),
        ("comp_2v2_ctce", "MultiGrid-CompetativeRedBlueDoor-v3-CTCE-2v2", 7, {"max_steps": 80}, 8),
        ("comp_2v2_view9", "MultiGrid-CompetativeRedBlueDoor-v3-DTDE-2v2-Death-Match", 9, {"max_steps": 90}, 8),
    ):
        cls, kw = cfg[reg]
        kw = dict(kw, **kwx)
        # (the CTCE group holds per-TEAM rewards and flags; its name keeps it out of the per-agent replay tests)
        record_competitive("ctce_2v2" if "ctce" in gname else gname, kw, nt, 70, zlib.crc32(gname.encode()) % 1000 + 50,
                           edit=comp_edit, action_p=PICKY, script=comp_script, view=view, ctce="ctce" in gname)
    record_competitive("comp_1v1_plain", dict(cfg["MultiGrid-CompetativeRedBlueDoor-v3-DTDE-1v1"][1]), 6, 320, 61)

    if not ONLY or "layouts" in ONLY:
        record_layouts()


def _ck(entry, agents):
    cls, kw = entry
    return (cls, agents), kw


def make_env(*args, **kw):
    """``make_env(cls, agents, **kw)`` or ``make_env((cls, agents), registry_kwargs, **kw)``."""
    if len(args) == 2 and isinstance(args[0], tuple):
        (cls, agents), base_kw = args
        kw = dict(base_kw, **kw)
    else:
        cls, agents = args[0], args[1]
    view = kw.pop("view", 7)
    return cls(agents=agents, teams=teams_for(agents), policies_map={}, agent_view_size=view, **kw)


if __name__ == "__main__":
    main()
