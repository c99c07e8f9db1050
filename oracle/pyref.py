"""CPU restatement of the reference ``MultiGridEnv.step / reset / gen_obs`` path.

THIS FILE IS TEST INFRASTRUCTURE.  It is the checker the HIP path is compared
against; nothing under ``multigrid_amd/`` imports it.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.

Parity status: PINNED.  ``tests/golden/*.npz`` hold trajectories recorded from
the unmodified reference (``tests/golden/make_golden.py`` imports it from
``/root/reference`` in the build container) and ``tests/test_oracle_golden.py``
replays every one of them through this file bit for bit.

The code keeps the reference's one-object-per-env, dict-in / dict-out shape so
that it also serves as the "reference-style Python step loop" CPU baseline.
Every function cites the reference lines it restates (paths relative to the
reference checkout).

State layout
------------
grid    (W, H, 3) integer array, ``grid[x, y] = (type, color, state)``
        (multigrid/core/grid.py:53).
agents  (A, 8) integer array, columns ``X Y DIR TERM CT CC CS COLOR`` - the
        reference's 9-column ``AgentState`` row (multigrid/core/agent.py:266-275)
        without the constant type column and with position first.
meta    (4,) integers: RedBlueDoors ``(red_x, red_y, blue_x, blue_y)``; unused
        by the other environment kinds (LockedHallway with more than six rooms
        counts its distinct door colours from the grid itself).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

# --- index tables (multigrid/core/constants.py:32-102, actions.py:4-15) -------
UNSEEN, EMPTY, WALL, FLOOR, DOOR, KEY, BALL, BOX, GOAL, LAVA, AGENT = range(11)
RED, GREEN, BLUE, PURPLE, YELLOW, GREY = range(6)
OPEN, CLOSED, LOCKED = 0, 1, 2
LEFT, RIGHT, FORWARD, PICKUP, DROP, TOGGLE, DONE = range(7)
# DIR_TO_VEC, multigrid/core/constants.py:20-29 (right, down, left, up)
DIR_VEC = ((1, 0), (0, 1), (-1, 0), (0, -1))

EMPTY_CELL = (EMPTY, 0, 0)  # WorldObj.empty(), world_object.py:131-137
WALL_CELL = (WALL, GREY, 0)  # Wall().encode(), utils/obs.py:13
UNSEEN_CELL = (UNSEEN, 0, 0)  # utils/obs.py:14

# agent row columns
X, Y, DIR, TERM, CT, CC, CS, COLOR = range(8)

KIND_EMPTY, KIND_REDBLUEDOORS, KIND_BLOCKEDUNLOCKPICKUP, KIND_LOCKEDHALLWAY, KIND_COMPETITIVE = range(5)
KIND_NAMES = ("empty", "redbluedoors", "blockedunlockpickup", "lockedhallway", "competitive")
AVAILABLE, PICKEDUP = 4, 3  # State.available / State.pickedup, multigrid/core/constants.py:91-102
# CompetativeRedBlueDoorEnvV3 state the arrays do not show, kept in `flags` (bits) and `meta` (counters)
FLAG_KEY_USED = (1, 2)      # the red / blue key OBJECT is "picked up" (competative_red_blue_door.py:434-435)
FLAG_DOOR_GONE = (4, 8)     # the red / blue door's cell was cleared under an eliminated agent (:503)
# reward_schemes[agent] columns (multigrid/envs/__init__.py:100-108)
RS_ELIMINATED, RS_KEY, RS_BALL, RS_INVALID = range(4)


@dataclass
class EnvSpec:
    """Static configuration of one environment (multigrid/base.py:87-113)."""

    kind: int
    width: int
    height: int
    num_agents: int
    max_steps: int
    view_size: int = 7
    see_through_walls: bool = False
    allow_agent_overlap: bool = True
    joint_reward: bool = False
    success_any: bool = True  # success_termination_mode == 'any'
    failure_any: bool = False  # failure_termination_mode == 'any'
    num_rooms: int = 0  # LockedHallway only (locked_hallway.py:129)
    mission: str = "maximize reward"
    # CompetativeRedBlueDoorEnvV3 only (competative_red_blue_door.py:114-130)
    death_match: bool = False
    per_agent_outputs: bool = True  # training_scheme "DTDE" / "CTDE" (dtde_step) rather than "CTCE"
    reward_schemes: Optional[Sequence[Sequence[float]]] = None  # [A][4], columns RS_*
    agent_names: Optional[Sequence[str]] = None


def agents_from_reference(agent_states: np.ndarray) -> np.ndarray:
    """(A, 9) reference AgentState rows -> (A, 8) rows of this module."""
    s = np.asarray(agent_states).astype(np.int64)
    out = np.zeros((s.shape[0], 8), dtype=np.int64)
    out[:, X], out[:, Y] = s[:, 3], s[:, 4]
    out[:, DIR], out[:, TERM] = s[:, 2], s[:, 5]
    out[:, CT:CS + 1] = s[:, 6:9]
    out[:, COLOR] = s[:, 1]
    return out


# --- observation -------------------------------------------------------------

def _view_origin(x: int, y: int, d: int, v: int) -> Tuple[int, int]:
    """Top-left world cell of the view square (utils/obs.py:272-310)."""
    h = v // 2
    if d == 0:
        return x, y - h
    if d == 1:
        return x - h, y
    if d == 2:
        return x - v + 1, y - h
    return x - h, y - v + 1


def gen_obs_grid(grid: np.ndarray, agents: np.ndarray, v: int) -> np.ndarray:
    """Egocentric sub-grids WITHOUT the visibility mask (utils/obs.py:127-203)."""
    n = agents.shape[0]
    w, h = grid.shape[0], grid.shape[1]
    if n > 1:  # obs.py:157-165 - agents are drawn only in multi-agent envs
        enc = grid.copy()
        for a in range(n):
            if not agents[a, TERM]:
                enc[agents[a, X], agents[a, Y]] = (AGENT, agents[a, COLOR], agents[a, DIR])
    else:  # obs.py:166-167
        enc = grid
    out = np.empty((n, v, v, 3), dtype=grid.dtype)
    for a in range(n):
        d = int(agents[a, DIR])
        tx, ty = _view_origin(int(agents[a, X]), int(agents[a, Y]), d, v)
        rot = (d + 1) % 4  # obs.py:174
        for i in range(v):
            for j in range(v):
                wx, wy = tx + i, ty + j
                if rot == 0:  # obs.py:183-190
                    ir, jr = i, j
                elif rot == 1:
                    ir, jr = j, v - i - 1
                elif rot == 2:
                    ir, jr = v - i - 1, v - j - 1
                else:
                    ir, jr = v - j - 1, i
                if 0 <= wx < w and 0 <= wy < h:  # obs.py:193-196
                    out[a, ir, jr] = enc[wx, wy]
                else:
                    out[a, ir, jr] = WALL_CELL
        out[a, v // 2, v - 1] = agents[a, CT:CS + 1]  # obs.py:201
    return out


def _see_behind(cell) -> bool:
    """utils/obs.py:44-61."""
    if cell[0] == WALL:
        return False
    if cell[0] == DOOR and cell[2] != OPEN:
        return False
    return True


def get_vis_mask(obs: np.ndarray) -> np.ndarray:
    """Visibility propagation (utils/obs.py:231-269)."""
    n, w, h = obs.shape[:3]
    vis = np.zeros((n, w, h), dtype=bool)
    vis[:, w // 2, h - 1] = True
    for a in range(n):
        sb = [[_see_behind(obs[a, i, j]) for j in range(h)] for i in range(w)]
        m = vis[a]
        for j in range(h - 1, -1, -1):
            for i in range(0, w - 1):  # forward pass, obs.py:254-259
                if m[i, j] and sb[i][j]:
                    m[i + 1, j] = True
                    if j > 0:
                        m[i + 1, j - 1] = True
                        m[i, j - 1] = True
            for i in range(w - 1, 0, -1):  # backward pass, obs.py:262-267
                if m[i, j] and sb[i][j]:
                    m[i - 1, j] = True
                    if j > 0:
                        m[i - 1, j - 1] = True
                        m[i, j - 1] = True
    return vis


def gen_obs_grid_encoding(grid, agents, v: int, see_through_walls: bool) -> np.ndarray:
    """utils/obs.py:64-99."""
    obs = gen_obs_grid(grid, agents, v)
    if not see_through_walls:
        vis = get_vis_mask(obs)
        obs[~vis] = UNSEEN_CELL
    return obs


# --- vectorised observation (same result, used when the port is timed) --------

_VIEW_TABLES: Dict[int, np.ndarray] = {}


def _view_table(v: int) -> np.ndarray:
    """(4, v, v, 2) world offsets of view cell (u, w) for each heading."""
    if v not in _VIEW_TABLES:
        t = np.zeros((4, v, v, 2), dtype=np.int64)
        for d in range(4):
            tx, ty = _view_origin(0, 0, d, v)
            rot = (d + 1) % 4
            for i in range(v):
                for j in range(v):
                    if rot == 0:
                        ir, jr = i, j
                    elif rot == 1:
                        ir, jr = j, v - i - 1
                    elif rot == 2:
                        ir, jr = v - i - 1, v - j - 1
                    else:
                        ir, jr = v - j - 1, i
                    t[d, ir, jr] = (tx + i, ty + j)
        _VIEW_TABLES[v] = t
    return _VIEW_TABLES[v]


def gen_obs_fast(grid, agents, v: int, see_through_walls: bool) -> np.ndarray:
    """Vectorised twin of :func:`gen_obs_grid_encoding` (checked equal in tests)."""
    n = agents.shape[0]
    w, h = grid.shape[0], grid.shape[1]
    pad = np.empty((w + 2 * v, h + 2 * v, 3), dtype=grid.dtype)
    pad[...] = WALL_CELL
    pad[v:v + w, v:v + h] = grid
    if n > 1:
        for a in range(n):
            if not agents[a, TERM]:
                pad[v + agents[a, X], v + agents[a, Y]] = (AGENT, agents[a, COLOR], agents[a, DIR])
    tab = _view_table(v)[agents[:, DIR]]  # (n, v, v, 2)
    wx = tab[..., 0] + agents[:, X, None, None] + v
    wy = tab[..., 1] + agents[:, Y, None, None] + v
    obs = pad[wx, wy]
    obs[:, v // 2, v - 1] = agents[:, CT:CS + 1]
    if see_through_walls:
        return obs
    full = (1 << v) - 1
    for a in range(n):
        t, s = obs[a, :, :, 0], obs[a, :, :, 2]
        sb = ~((t == WALL) | ((t == DOOR) & (s != OPEN)))
        rows_sb = [int(sum(1 << i for i in range(v) if sb[i, j])) for j in range(v)]
        cur = 1 << (v // 2)
        for j in range(v - 1, -1, -1):
            sbj = rows_sb[j]
            m = cur
            up = 0
            for i in range(0, v - 1):
                if (m >> i) & 1 and (sbj >> i) & 1:
                    m |= 1 << (i + 1)
                    up |= (1 << (i + 1)) | (1 << i)
            for i in range(v - 1, 0, -1):
                if (m >> i) & 1 and (sbj >> i) & 1:
                    m |= 1 << (i - 1)
                    up |= (1 << (i - 1)) | (1 << i)
            hidden = ~m & full
            for i in range(v):
                if (hidden >> i) & 1:
                    obs[a, i, j] = UNSEEN_CELL
            cur = up
    return obs


# --- environment ---------------------------------------------------------------

# --- OneHotObsWrapper (multigrid/wrappers.py:60-146) ---------------------------
# dim_sizes of wrappers.py:95: len(Type), len(Color), max(len(State), len(Direction))
ONE_HOT_DIMS = (11, 6, 7)


def one_hot(x: np.ndarray, dim_sizes: Sequence[int] = ONE_HOT_DIMS) -> np.ndarray:
    """``OneHotObsWrapper.one_hot`` (multigrid/wrappers.py:114-146), loop for loop: every 2D slice
    ``x[:, :, d]`` is encoded on its own run of ``dim_sizes[d]`` channels."""
    out = np.zeros((x.shape[0], x.shape[1], sum(dim_sizes)), dtype=np.uint8)
    dim_offset = 0
    for d in range(len(dim_sizes)):
        for i in range(x.shape[0]):
            for j in range(x.shape[1]):
                out[i, j, dim_offset + x[i, j, d]] = 1  # wrappers.py:141-142
        dim_offset += dim_sizes[d]
    return out


def one_hot_batch(images: np.ndarray, dim_sizes: Sequence[int] = ONE_HOT_DIMS) -> np.ndarray:
    """``one_hot`` over any number of leading dimensions (``[..., V, V, 3] -> [..., V, V, 24]``)."""
    images = np.asarray(images)
    out = np.zeros(images.shape[:-1] + (sum(dim_sizes),), dtype=np.uint8)
    dim_offset = 0
    for d, n in enumerate(dim_sizes):
        np.put_along_axis(out, (dim_offset + images[..., d:d + 1]).astype(np.int64), 1, axis=-1)
        dim_offset += n
    return out


def competitive_v2_observation(image: np.ndarray, direction: int) -> np.ndarray:
    """``CompetativeRedBlueDoorWrapperV2.observation`` for one agent (multigrid/wrappers.py:388-414), statement for
    statement: the image is one-hot encoded in the loop over the agents (:393-402) and the result is one-hot encoded
    AGAIN (:409; ``one_hot`` then reads the first three of its 24 channels as indices), a plane holding the direction is
    put in front (:410-416).  ``(V, V, 3) -> (V, V, 25)``."""
    first = one_hot(np.asarray(image).astype(np.int64))          # :402
    second = one_hot(first.astype(np.int64))                       # :409
    plane = np.full(second.shape[:2] + (1,), direction).astype("uint8")  # :410-412
    return np.concatenate((plane, second), axis=2)                 # :413-416


class Env:
    """One environment, stepped the way ``MultiGridEnv`` is (multigrid/base.py:37)."""

    def __init__(self, spec: EnvSpec, fast_obs: bool = False):
        self.spec = spec
        self.fast_obs = fast_obs
        self.grid = np.zeros((spec.width, spec.height, 3), dtype=np.int64)
        self.agents = np.zeros((spec.num_agents, 8), dtype=np.int64)
        self.meta = np.zeros(4, dtype=np.int64)
        self.step_count = 0
        # LockedHallway: doors already rewarded, as a bit per door colour
        # (locked_hallway.py:190,204-205; door colours are distinct for <= 6 rooms)
        self.unlocked_mask = 0
        self.unlocked_at = set()
        self.door_colours = 0
        self.missions = [""] * spec.num_agents  # CompetativeRedBlueDoor: the agents' mission texts (set by the caller at reset)
        self.obs_missions = list(self.missions)
        self.pickup_events = {}
        # RedBlueDoors: the blue Door OBJECT was re-closed after a failure while
        # its grid cell still reads open (redbluedoors.py:185 has no grid.update)
        self.blue_stale_open = False
        # CompetativeRedBlueDoorEnvV3: FLAG_KEY_USED / FLAG_DOOR_GONE bits, and meta[0] / meta[1] = how often the
        # discount factor of the red / blue ball has been squared (competative_red_blue_door.py:452-454)
        self.flags = 0

    # -- reset (base.py:310-362 with the layout supplied by the caller) --------
    def load_layout(self, grid, agents, meta=None):
        self.grid[...] = grid
        self.agents[...] = agents
        self.meta[...] = 0 if meta is None else meta
        self.step_count = 0  # base.py:350
        self.unlocked_mask = 0  # locked_hallway.py:190
        self.unlocked_at = set()
        doors = self.grid[..., 0] == DOOR  # len(self.rooms), locked_hallway.py:160-166
        self.door_colours = len(set(self.grid[..., 1][doors].tolist()))
        self.blue_stale_open = False
        self.flags = 0
        return self.gen_obs()

    # -- helpers ---------------------------------------------------------------
    def _reward(self) -> float:
        """base.py:655-659."""
        return 1 - 0.9 * (self.step_count / self.spec.max_steps)

    def _front(self, i: int) -> Tuple[int, int]:
        """agent.py:153-160, utils/misc.py:6-12."""
        dx, dy = DIR_VEC[int(self.agents[i, DIR])]
        return int(self.agents[i, X]) + dx, int(self.agents[i, Y]) + dy

    def _is_blue_door(self, fx: int, fy: int) -> bool:
        return (self.spec.kind == KIND_REDBLUEDOORS and fx == self.meta[2] and fy == self.meta[3])

    def _object_state(self, fx: int, fy: int) -> int:
        """State of the world OBJECT at a cell (differs from the grid cell only
        for the re-closed blue door, see ``blue_stale_open``)."""
        if self.blue_stale_open and self._is_blue_door(fx, fy):
            return CLOSED
        return int(self.grid[fx, fy, 2])

    def _agent_at(self, fx: int, fy: int) -> bool:
        """base.py:485,512 - every agent counts, terminated ones included."""
        a = self.agents
        return bool(((a[:, X] == fx) & (a[:, Y] == fy)).any())

    def on_success(self, i, rewards, terminations: Optional[dict]):
        """base.py:535-562 (``terminations`` falsy when called from handle_actions)."""
        sp = self.spec
        if sp.success_any:
            if terminations:
                self.agents[:, TERM] = 1
                for k in range(sp.num_agents):
                    terminations[k] = True
        else:
            self.agents[i, TERM] = 1
            if terminations:
                terminations[i] = True
        if sp.joint_reward:
            for k in range(sp.num_agents):
                rewards[k] = self._reward()
        else:
            rewards[i] = self._reward()

    def on_failure(self, i, rewards, terminations: Optional[dict]):
        """base.py:564-585."""
        sp = self.spec
        if sp.failure_any:
            if terminations:
                self.agents[:, TERM] = 1
                for k in range(sp.num_agents):
                    terminations[k] = True
        else:
            self.agents[i, TERM] = 1
            if terminations:
                terminations[i] = True

    # -- handle_actions (base.py:440-533) ---------------------------------------
    def handle_actions(self, actions, order) -> dict:
        sp, g, ag = self.spec, self.grid, self.agents
        rewards = {i: 0 for i in range(sp.num_agents)}
        for i in order:
            i = int(i)
            if ag[i, TERM]:  # base.py:467
                continue
            act = int(actions[i])
            if act == LEFT:  # base.py:471-472
                ag[i, DIR] = (ag[i, DIR] - 1) % 4
            elif act == RIGHT:  # base.py:475-476
                ag[i, DIR] = (ag[i, DIR] + 1) % 4
            elif act == FORWARD:  # base.py:479-494
                fx, fy = self._front(i)
                t = int(g[fx, fy, 0])
                can_overlap = t in (EMPTY, GOAL, FLOOR, LAVA) or (
                    t == DOOR and self._object_state(fx, fy) == OPEN)  # world_object.py:197,287,314,338,450
                if can_overlap:
                    if not sp.allow_agent_overlap and self._agent_at(fx, fy):
                        continue
                    ag[i, X], ag[i, Y] = fx, fy
                    if t == GOAL:
                        self.on_success(i, rewards, {})
                    if t == LAVA:
                        self.on_failure(i, rewards, {})
            elif act == PICKUP:  # base.py:497-504
                fx, fy = self._front(i)
                t = int(g[fx, fy, 0])
                ok = t in (KEY, BALL, BOX)
                if sp.kind == KIND_COMPETITIVE:  # competative_red_blue_door.py:573-577: own key, the others' ball
                    ok = (t == KEY and g[fx, fy, 1] == ag[i, COLOR]) or (t == BALL and g[fx, fy, 1] != ag[i, COLOR])
                if ok and ag[i, CT] == EMPTY:
                    ag[i, CT:CS + 1] = g[fx, fy]
                    g[fx, fy] = EMPTY_CELL
            elif act == DROP:  # base.py:507-516
                fx, fy = self._front(i)
                if ag[i, CT] != EMPTY and g[fx, fy, 0] == EMPTY and not self._agent_at(fx, fy):
                    g[fx, fy] = ag[i, CT:CS + 1]
                    if sp.kind == KIND_COMPETITIVE:
                        # the cell takes the OBJECT's encoding (grid.py:91-95); the agent row only held the copy
                        # made when the key was picked up (agent.py:385)
                        g[fx, fy, 2] = self._carried_state(i)
                    ag[i, CT:CS + 1] = EMPTY_CELL
            elif act == TOGGLE:  # base.py:519-524
                fx, fy = self._front(i)
                t = int(g[fx, fy, 0])
                if t == DOOR:  # Door.toggle, world_object.py:456-472
                    s = self._object_state(fx, fy)
                    if s == LOCKED:
                        if ag[i, CT] == KEY and ag[i, CC] == g[fx, fy, 1]:
                            g[fx, fy, 2] = OPEN
                    else:
                        g[fx, fy, 2] = CLOSED if s == OPEN else OPEN
                        if self._is_blue_door(fx, fy):
                            self.blue_stale_open = False  # grid.update re-syncs the cell
                elif t == BOX:  # Box.toggle, world_object.py:653-659 (contents None)
                    g[fx, fy] = EMPTY_CELL
            elif act == DONE:  # base.py:527-528
                pass
            else:
                raise ValueError(f"Unknown action: {act}")
        return rewards

    # -- gen_obs (base.py:410-438) ----------------------------------------------
    def gen_obs_array(self) -> np.ndarray:
        fn = gen_obs_fast if self.fast_obs else gen_obs_grid_encoding
        return fn(self.grid, self.agents, self.spec.view_size, self.spec.see_through_walls)

    def gen_obs(self) -> dict:
        img = self.gen_obs_array()
        return {
            i: {"image": img[i], "direction": self.agents[i, DIR], "mission": self.spec.mission}
            for i in range(self.spec.num_agents)
        }

    # -- step (base.py:364-408 followed by the subclass hook) --------------------
    def step(self, actions, order: Optional[Sequence[int]] = None):
        sp = self.spec
        if order is None:
            order = range(sp.num_agents)
        if sp.kind == KIND_COMPETITIVE:
            return self.step_competitive(actions, list(order))
        self.step_count += 1  # base.py:395
        rewards = self.handle_actions(actions, order)
        obs = self.gen_obs()
        terminations = {i: bool(self.agents[i, TERM]) for i in range(sp.num_agents)}  # base.py:400
        truncated = self.step_count >= sp.max_steps  # base.py:401
        truncations = {i: truncated for i in range(sp.num_agents)}

        if sp.kind == KIND_REDBLUEDOORS:  # redbluedoors.py:176-185
            rx, ry, bx, by = (int(m) for m in self.meta)
            for i in range(sp.num_agents):
                if int(actions[i]) == TOGGLE:
                    fx, fy = self._front(i)
                    if (fx, fy) == (bx, by) and self.grid[bx, by, 0] == DOOR \
                            and self._object_state(bx, by) == OPEN:
                        if self.grid[rx, ry, 2] == OPEN:
                            self.on_success(i, rewards, terminations)
                        else:
                            self.on_failure(i, rewards, terminations)
                            self.blue_stale_open = True  # object closed, cell untouched
        elif sp.kind == KIND_BLOCKEDUNLOCKPICKUP:  # blockedunlockpickup.py:165-167
            for i in range(sp.num_agents):
                if self.agents[i, CT] == BOX:
                    self.on_success(i, rewards, terminations)
        elif sp.kind == KIND_LOCKEDHALLWAY:  # locked_hallway.py:200-215
            # `unlocked_doors` holds door OBJECTS (:204-205).  Up to six rooms every door has a colour of its own and
            # is told by it; beyond that colours repeat (:150-151) and a door is told by where it stands.  `self.rooms`
            # is keyed by colour (:160-166), so the all-doors ending compares with the number of DISTINCT colours (:213).
            many = sp.num_rooms > 6
            for i in range(sp.num_agents):
                if int(actions[i]) == TOGGLE:
                    fx, fy = self._front(i)
                    if self.grid[fx, fy, 0] == DOOR and self.grid[fx, fy, 2] != LOCKED:
                        if many:
                            new = (fx, fy) not in self.unlocked_at
                            self.unlocked_at.add((fx, fy))
                        else:
                            bit = 1 << int(self.grid[fx, fy, 1])
                            new = not self.unlocked_mask & bit
                            self.unlocked_mask |= bit
                        if new:
                            if sp.joint_reward:
                                for k in rewards:
                                    rewards[k] += self._reward()
                            else:
                                rewards[i] += self._reward()
            opened = len(self.unlocked_at) if many else bin(self.unlocked_mask).count("1")
            if opened == (self.door_colours if many else sp.num_rooms):
                for i in range(sp.num_agents):
                    terminations[i] = True
        return obs, rewards, terminations, truncations, {}

    # -- CompetativeRedBlueDoorEnvV3 (multigrid/envs/competative_red_blue_door.py) ----------------------
    def _comp_doors(self):
        """Door positions of _gen_grid (:201-211): red at the bottom of the left wall, blue at the top of the right."""
        w, h = self.spec.width, self.spec.height
        return (w // 4, h - 2), (w // 4 + w // 2 - 1, 1)

    def _carried_state(self, i: int) -> int:
        """State of the OBJECT agent i carries: a key that has paid its pickup reward is State.pickedup
        (:434-435, world_object.py:534-552) whatever the copy in the agent row says."""
        ag = self.agents
        if ag[i, CT] == KEY and ag[i, CC] in (RED, BLUE) and self.flags & FLAG_KEY_USED[ag[i, CC] // 2]:
            return PICKEDUP
        return int(ag[i, CS])

    def _comp_door_open(self, which: int) -> bool:
        """``self.red_door.is_open`` / ``self.blue_door.is_open`` (:485): the door OBJECT, which keeps its last state
        (open: an agent stood in it) once its cell has been cleared."""
        if self.flags & FLAG_DOOR_GONE[which]:
            return True
        x, y = self._comp_doors()[which]
        return self.grid[x, y, 0] == DOOR and self.grid[x, y, 2] == OPEN

    def _comp_handle_steps(self, i, act, rewards, terminations):
        """_handle_steps, :475-523."""
        sp, ag = self.spec, self.agents
        if act != TOGGLE:
            return
        fx, fy = self._front(i)
        victim = None
        for k in range(sp.num_agents):  # :479-481, the last opponent standing there
            if (ag[k, X], ag[k, Y]) == (fx, fy) and ag[k, COLOR] != ag[i, COLOR]:
                victim = k
        doors = self._comp_doors()
        if victim is None:
            for which, (dx, dy) in enumerate(doors):  # :484 `fwd_obj == self.red_door` is object identity
                if (fx, fy) == (dx, dy) and self.grid[dx, dy, 0] == DOOR and not self.flags & FLAG_DOOR_GONE[which]:
                    if (self._comp_door_open(0) or self._comp_door_open(1)) and self.grid[dx, dy, 1] == ag[i, COLOR]:
                        for k in range(sp.num_agents):  # :487-492
                            if ag[k, COLOR] == ag[i, COLOR] and not ag[k, TERM]:
                                self.on_success(k, rewards, terminations)
                                self.info[k]["door_open_done"] = True
        elif sp.death_match:  # :498-523
            self.on_failure(victim, rewards, None)
            self.info[victim]["got_eliminated_done"] = True  # :502
            vx, vy = int(ag[victim, X]), int(ag[victim, Y])
            for which, (dx, dy) in enumerate(doors):
                if (vx, vy) == (dx, dy) and self.grid[dx, dy, 0] == DOOR:
                    self.flags |= FLAG_DOOR_GONE[which]
            self.grid[vx, vy] = EMPTY_CELL  # :503
            ag[victim, X], ag[victim, Y] = (13, 2) if ag[victim, COLOR] == BLUE else (2, 2)  # :504-506
            rewards[i] += sp.reward_schemes[i][RS_ELIMINATED]
            rewards[victim] -= 1
            if all(ag[k, TERM] for k in range(sp.num_agents) if ag[k, COLOR] != ag[i, COLOR]):  # :511-517
                for k in range(sp.num_agents):
                    if ag[k, COLOR] == ag[i, COLOR] and not ag[k, TERM]:
                        self.on_success(k, rewards, terminations)
                        self.info[k]["eliminated_opponents_done"] = True  # :517-522
                        self.info[k]["eliminated_opponent_num"] += 1

    def _comp_pickup_rewards(self, i, act, rewards):
        """_default_extra_action_handle_steps, :425-472 (the handler of agents without a policy object)."""
        sp, ag = self.spec, self.agents
        if act != PICKUP:
            return
        rs = sp.reward_schemes[i]
        (rx, ry), (bx, by) = self._comp_doors()
        if ag[i, CT] == KEY and self._carried_state(i) == AVAILABLE and ag[i, CC] == ag[i, COLOR]:  # :428-436
            if ag[i, CC] in (RED, BLUE):
                self.flags |= FLAG_KEY_USED[ag[i, CC] // 2]
            rewards[i] += rs[RS_KEY]
            self.pickup_events[i] = 1
            # "mimic communications", :438-443 (`training_scheme == "DTDE" or "CTDE"` is always true)
            self.missions[i] = "Go open the door with the key"
            for k in range(sp.num_agents):
                if ag[k, COLOR] == ag[i, COLOR] and k != i:
                    self.missions[k] = "Go move away the ball"
            return
        if ag[i, CT] == BALL and ag[i, CC] != ag[i, COLOR] and ag[i, CC] in (RED, BLUE):  # :445-454
            # Ball(color="blue", init_pos=next to the red door), Ball(color="red", init_pos=next to the blue door) :230-232
            init = (bx - 1, by) if ag[i, CC] == RED else (rx + 1, ry)
            if self._front(i) == init:
                slot = ag[i, CC] // 2
                d = 0.9  # Ball.__new__ default, world_object.py:583
                for _ in range(int(self.meta[slot])):
                    d *= d
                rewards[i] += rs[RS_BALL] * d
                self.meta[slot] += 1  # discount_factor *= discount_factor
                self.pickup_events[i] = 2
                self.missions[i] = "Go move away the ball"  # :456-468
                for k in range(sp.num_agents):
                    if ag[k, COLOR] == ag[i, COLOR] and k != i:
                        own_key = ag[k, CT] == KEY and ag[k, CC] == ag[k, COLOR]
                        self.missions[k] = "Go open the door with the key" if own_key else "Go pick up the key"
                return
        rewards[i] -= rs[RS_INVALID]  # :470-472

    def step_competitive(self, actions, order):
        """CompetativeRedBlueDoorEnvV3.step (:526-536): MultiGridEnv.step, then dtde_step / ctce_step (:319-423) with the
        default handlers; agents act, and are handled, in the order of the keys of the actions dict."""
        sp = self.spec
        self.step_count += 1
        rewards = self.handle_actions(actions, order)
        obs = self.gen_obs()
        self.obs_missions = list(self.missions)  # the texts this step's observations carry (:307,315); the handlers
        self.pickup_events = {i: 0 for i in range(sp.num_agents)}  # below rewrite them for the next (1 key, 2 ball pickup)
        terminations = {i: bool(self.agents[i, TERM]) for i in range(sp.num_agents)}
        truncated = self.step_count >= sp.max_steps
        truncations = {i: truncated for i in range(sp.num_agents)}
        if sp.per_agent_outputs and any(terminations.values()) and not all(terminations.values()):  # :375-378
            terminations = {i: False for i in terminations}
        # the info dict of the step, here per agent index (:380-388; ctce_step keys it by team and its entries are
        # shared by the team's agents, :320-328: or the flags and sum the counts over a team's agents)
        self.info = {i: {"door_open_done": False, "got_eliminated_done": False, "eliminated_opponents_done": False,
                         "eliminated_opponent_num": 0} for i in range(sp.num_agents)}
        for i in order:
            i = int(i)
            self._comp_handle_steps(i, int(actions[i]), rewards, terminations)
            self._comp_pickup_rewards(i, int(actions[i]), rewards)
        return obs, rewards, terminations, truncations, self.info

    def is_done(self) -> bool:
        """base.py:587-596."""
        return self.step_count >= self.spec.max_steps or bool(self.agents[:, TERM].all())
