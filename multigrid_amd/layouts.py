"""Host-side start-state ("layout") generators for the in-scope environments.

Each generator restates the reference's ``_gen_grid`` for one environment class and draws
from numpy ``Generator`` objects in exactly the reference's order, so that given the same
two generators it returns the reference's own layout (checked against recorded layouts in
``tests/test_layouts.py``).  The reference draws from TWO generators:

* ``mixin_rng``  - the construction-time generator kept by ``RandomMixin``
  (multigrid/utils/random.py:20; ``_rand_int`` / ``_rand_perm`` / ``_rand_color``);
* ``env_rng``    - ``gym.Env.np_random``, re-seeded by ``reset(seed)``; used only for the
  random door position of ``RoomGrid.add_door`` (multigrid/core/roomgrid.py:314).

A layout is ``(grid uint8 (W,H,3), agents uint8 (A,8), meta uint8 (4,), mission str)`` in the
array formats of ``include/mgrid.h``.
"""
from __future__ import annotations

from math import ceil
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .constants import (AVAILABLE, BALL, BLUE, BOX, COLOR_NAMES, DOOR, EMPTY, GOAL, GREEN, KEY, LOCKED, RED,
                        TYPE_NAMES, WALL, GREY)

Layout = Tuple[np.ndarray, np.ndarray, np.ndarray, str]

_DIR_VEC = ((1, 0), (0, 1), (-1, 0), (0, -1))  # multigrid/core/constants.py:20-29


class _Builder:
    """Grid + agent rows under construction, with the reference's placement helpers."""

    def __init__(self, width: int, height: int, agent_colors: Sequence[int], mixin_rng, env_rng=None):
        self.w, self.h = width, height
        self.grid = np.zeros((width, height, 3), dtype=np.uint8)
        self.grid[..., 0] = EMPTY  # Grid.__init__, multigrid/core/grid.py:53-54
        a = len(agent_colors)
        self.pos = np.full((a, 2), -1, dtype=np.int64)  # Agent.reset, multigrid/core/agent.py:172-173
        self.dir = np.full(a, -1, dtype=np.int64)
        self.colors = list(agent_colors)
        self.rng = mixin_rng
        self.env_rng = env_rng if env_rng is not None else mixin_rng

    # RandomMixin, multigrid/utils/random.py:22-90
    def rand_int(self, lo: int, hi: int) -> int:
        return int(self.rng.integers(lo, hi))

    def rand_color(self) -> int:
        return self.rand_int(0, len(COLOR_NAMES))

    def rand_perm(self, items: list) -> list:
        out = list(items)
        self.rng.shuffle(out)
        return out

    # Grid.wall_rect, multigrid/core/grid.py:168-186
    def wall_rect(self, x: int, y: int, w: int, h: int):
        wall = (WALL, GREY, 0)
        self.grid[x:x + w, y] = wall
        self.grid[x:x + w, y + h - 1] = wall
        self.grid[x, y:y + h] = wall
        self.grid[x + w - 1, y:y + h] = wall

    def is_free(self, x: int, y: int) -> bool:
        return self.grid[x, y, 0] == EMPTY

    def agent_at(self, x: int, y: int) -> bool:
        return bool(((self.pos[:, 0] == x) & (self.pos[:, 1] == y)).any())

    # MultiGridEnv.place_obj, multigrid/base.py:661-728
    def place_obj(self, cell: Optional[tuple], top=None, size=None, reject_next_to=False) -> Tuple[int, int]:
        top = (0, 0) if top is None else (max(top[0], 0), max(top[1], 0))
        size = (self.w, self.h) if size is None else size
        while True:
            x = self.rand_int(top[0], min(top[0] + size[0], self.w))
            y = self.rand_int(top[1], min(top[1] + size[1], self.h))
            if not self.is_free(x, y):
                continue
            if self.agent_at(x, y):
                continue
            if reject_next_to:  # multigrid/core/roomgrid.py:44-49
                d = np.sqrt(((np.array([x, y]) - self.pos) ** 2).sum(axis=-1))
                if (d <= 1).any():
                    continue
            break
        if cell is not None:
            self.grid[x, y] = cell
        return x, y

    # MultiGridEnv.place_agent, multigrid/base.py:738-749
    def place_agent(self, k: int, top=None, size=None, rand_dir=True) -> Tuple[int, int]:
        self.pos[k] = (-1, -1)
        x, y = self.place_obj(None, top, size)
        self.pos[k] = (x, y)
        if rand_dir:
            self.dir[k] = self.rand_int(0, 4)
        return x, y

    def finish(self, meta=(0, 0, 0, 0), mission: str = "") -> Layout:
        a = len(self.colors)
        agents = np.zeros((a, 8), dtype=np.uint8)
        assert (self.pos >= 0).all() and (self.dir >= 0).all()  # multigrid/base.py:338-339
        agents[:, 0], agents[:, 1], agents[:, 2] = self.pos[:, 0], self.pos[:, 1], self.dir
        agents[:, 4] = EMPTY  # carrying nothing: WorldObj.empty(), multigrid/core/agent.py:385
        agents[:, 7] = self.colors
        return self.grid, agents, np.asarray(meta, dtype=np.uint8), mission


def gen_empty(size: int, agent_colors: Sequence[int], mixin_rng, env_rng=None,
              agent_start_pos=(1, 1), agent_start_dir=0) -> Layout:
    """EmptyEnv._gen_grid, multigrid/envs/empty.py:152-171."""
    b = _Builder(size, size, agent_colors, mixin_rng, env_rng)
    b.wall_rect(0, 0, size, size)
    b.grid[size - 2, size - 2] = (GOAL, GREEN, 0)  # Goal(), world_object.py:284
    for k in range(len(agent_colors)):
        if agent_start_pos is not None and agent_start_dir is not None:
            b.pos[k] = agent_start_pos
            b.dir[k] = agent_start_dir
        else:
            b.place_agent(k)
    return b.finish(mission="get to the green goal square")


def gen_redbluedoors(size: int, agent_colors: Sequence[int], mixin_rng, env_rng=None) -> Layout:
    """RedBlueDoorsEnv._gen_grid, multigrid/envs/redbluedoors.py:141-168."""
    w, h = 2 * size, size
    b = _Builder(w, h, agent_colors, mixin_rng, env_rng)
    room_top, room_size = (w // 4, 0), (w // 2, h)
    b.wall_rect(0, 0, w, h)
    b.wall_rect(*room_top, *room_size)
    for k in range(len(agent_colors)):
        b.place_agent(k, top=room_top, size=room_size)
    rx, ry = room_top[0], b.rand_int(1, h - 1)
    b.grid[rx, ry] = (DOOR, RED, LOCKED)  # this fork creates both doors locked (:161,:167)
    bx, by = room_top[0] + room_size[0] - 1, b.rand_int(1, h - 1)
    b.grid[bx, by] = (DOOR, BLUE, LOCKED)
    return b.finish(meta=(rx, ry, bx, by), mission="open the red door then the blue door")


class _Rooms:
    """RoomGrid._gen_grid, multigrid/core/roomgrid.py:197-230."""

    def __init__(self, b: _Builder, room_size: int, num_rows: int, num_cols: int):
        self.b, self.rs, self.rows, self.cols = b, room_size, num_rows, num_cols
        for row in range(num_rows):
            for col in range(num_cols):
                b.wall_rect(col * (room_size - 1), row * (room_size - 1), room_size, room_size)
        b.dir[:] = 0
        b.pos[:] = ((num_cols // 2) * (room_size - 1) + room_size // 2,
                    (num_rows // 2) * (room_size - 1) + room_size // 2)

    def top(self, col: int, row: int) -> Tuple[int, int]:
        return col * (self.rs - 1), row * (self.rs - 1)

    def door_pos(self, col: int, row: int, d: int, rng=None) -> Tuple[int, int]:
        """Room.set_door_pos, multigrid/core/roomgrid.py:86-127."""
        left, top = self.top(col, row)
        right, bottom = left + self.rs - 1, top + self.rs - 1
        if d == 0:
            return right, int(rng.integers(top + 1, bottom)) if rng is not None else (top + bottom) // 2
        if d == 1:
            return (int(rng.integers(left + 1, right)) if rng is not None else (left + right) // 2), bottom
        if d == 2:
            return left, int(rng.integers(top + 1, bottom)) if rng is not None else (top + bottom) // 2
        return (int(rng.integers(left + 1, right)) if rng is not None else (left + right) // 2), top

    def remove_wall_down(self, col: int, row: int):
        """RoomGrid.remove_wall(dir=down), multigrid/core/roomgrid.py:347-349."""
        tx, ty = self.top(col, row)
        for i in range(1, self.rs - 1):
            self.b.grid[tx + i, ty + self.rs - 1] = (EMPTY, 0, 0)

    def place_agent(self, k: int, col: int, row: int):
        """RoomGrid.place_agent, multigrid/core/roomgrid.py:363-391."""
        b = self.b
        while True:
            b.place_agent(k, self.top(col, row), (self.rs, self.rs), True)
            dx, dy = _DIR_VEC[int(b.dir[k])]
            t = b.grid[b.pos[k, 0] + dx, b.pos[k, 1] + dy, 0]
            if t == EMPTY or t == WALL:
                break


def gen_blockedunlockpickup(agent_colors: Sequence[int], mixin_rng, env_rng=None, room_size: int = 6) -> Layout:
    """BlockedUnlockPickupEnv._gen_grid, multigrid/envs/blockedunlockpickup.py:136-158."""
    rows, cols = 1, 2
    w, h = (room_size - 1) * cols + 1, (room_size - 1) * rows + 1
    b = _Builder(w, h, agent_colors, mixin_rng, env_rng)
    rooms = _Rooms(b, room_size, rows, cols)
    rs = (room_size, room_size)
    # the box, in the right room: add_object -> place_in_room (roomgrid.py:232-270)
    box_color = b.rand_color()
    b.place_obj((BOX, box_color, 0), rooms.top(1, 0), rs, reject_next_to=True)
    # locked door between the rooms at a random height (roomgrid.py:272-321)
    door_color = b.rand_color()
    dx, dy = rooms.door_pos(0, 0, 0, rng=b.env_rng)
    b.grid[dx, dy] = (DOOR, door_color, LOCKED)
    # ball in front of the door (:149)
    b.grid[dx - 1, dy] = (BALL, b.rand_color(), 0)
    # key of the door's colour in the left room; WorldObj(type=key) has state 0 (roomgrid.py:269)
    b.place_obj((KEY, door_color, 0), rooms.top(0, 0), rs, reject_next_to=True)
    for k in range(len(agent_colors)):
        rooms.place_agent(k, 0, 0)
    return b.finish(mission=f"pick up the {COLOR_NAMES[box_color]} {TYPE_NAMES[BOX]}")


def gen_lockedhallway(num_rooms: int, agent_colors: Sequence[int], mixin_rng, env_rng=None, room_size: int = 5,
                      max_hallway_keys: int = 1, max_keys_per_room: int = 2) -> Layout:
    """LockedHallwayEnv._gen_grid, multigrid/envs/locked_hallway.py:146-184."""
    rows, cols = num_rooms // 2, 3
    w, h = (room_size - 1) * cols + 1, (room_size - 1) * rows + 1
    b = _Builder(w, h, agent_colors, mixin_rng, env_rng)
    rooms = _Rooms(b, room_size, rows, cols)
    left, hallway, right = 0, 1, 2
    ncol = len(COLOR_NAMES)
    color_sequence = list(range(ncol)) * ceil(num_rooms / ncol)
    color_sequence = b.rand_perm(color_sequence)[:num_rooms]
    for row in range(rows - 1):
        rooms.remove_wall_down(hallway, row)
    room_of = {}
    door_colors = b.rand_perm(color_sequence)
    for row in range(rows):
        for col, d in ((left, 0), (right, 2)):
            color = door_colors.pop()
            room_of[color] = (col, row)
            x, y = rooms.door_pos(col, row, d)
            b.grid[x, y] = (DOOR, color, LOCKED)
    num_hallway_keys = b.rand_int(1, max_hallway_keys + 1)
    hallway_top, hallway_size = rooms.top(hallway, 0), (room_size, h)
    key_state = 4  # Key() is created "available", world_object.py:519-521
    for key_color in color_sequence[:num_hallway_keys]:
        b.place_obj((KEY, key_color, key_state), hallway_top, hallway_size)
    key_index = num_hallway_keys
    while key_index < len(color_sequence):
        col, row = room_of[color_sequence[key_index - 1]]
        num_room_keys = b.rand_int(1, max_keys_per_room + 1)
        for key_color in color_sequence[key_index:key_index + num_room_keys]:
            b.place_obj((KEY, key_color, key_state), rooms.top(col, row), (room_size, room_size))
            key_index += 1
    for k in range(len(agent_colors)):
        b.place_agent(k, hallway_top, hallway_size)
    # beyond six rooms door colours repeat: meta[3] = len(self.rooms), the number of distinct door colours, at which
    # the all-doors ending is reached (locked_hallway.py:160-166,213); meta[0..1] = the doors paid so far (none)
    meta = (0, 0, 0, len(room_of)) if num_rooms > len(COLOR_NAMES) else (0, 0, 0, 0)
    return b.finish(meta=meta, mission="open all the doors")


def gen_competitive(size: int, agent_colors: Sequence[int], mixin_rng, env_rng=None, team_colors: Sequence[int] = (RED,),
                    has_obstacle: bool = False, randomization: bool = False) -> Layout:
    """CompetativeRedBlueDoorEnvV3._gen_grid, multigrid/envs/competative_red_blue_door.py:185-242.  ``team_colors``:
    the colours of the teams in the order of the ``teams`` dict (one key per team is placed in that order)."""
    w, h = 2 * size, size
    b = _Builder(w, h, agent_colors, mixin_rng, env_rng)
    room_top, room_size = (w // 4, 0), (w // 2, h)
    b.wall_rect(0, 0, w, h)
    b.wall_rect(*room_top, *room_size)
    rx, ry = room_top[0], h - 2                       # red door at the bottom of the left wall (:201-205)
    b.grid[rx, ry] = (DOOR, RED, LOCKED)
    bx, by = room_top[0] + room_size[0] - 1, 1        # blue door at the top of the right wall (:207-211)
    b.grid[bx, by] = (DOOR, BLUE, LOCKED)
    # Agents in the order sorted(key=(color == 'blue', color)) (:215): each is first placed at random (which
    # draws from the generator) and then moved to its fixed start next to the OTHER team's door.
    names = [COLOR_NAMES[c] for c in agent_colors]
    for idx, k in enumerate(sorted(range(len(names)), key=lambda k: (names[k] == "blue", names[k]))):
        if names[k] == "red":
            b.place_agent(k, top=(rx + 2, ry), size=(4, 4))
            b.pos[k], b.dir[k] = (bx - (idx + 2), by), 2
        else:
            b.place_agent(k, top=(bx - 1, by), size=(4, 4))
            b.pos[k], b.dir[k] = (rx + (idx + 1 if idx == 1 else idx), ry), 0
    if has_obstacle:                                   # a ball of the other colour in front of each door (:228-232)
        if RED in team_colors:
            b.grid[rx + 1, ry] = (BALL, BLUE, 0)
        if BLUE in team_colors:
            b.grid[bx - 1, by] = (BALL, RED, 0)
    key_positions = {RED: (7, 4), BLUE: (8, 3)}        # :236
    for c in team_colors:
        if randomization:
            b.place_obj((KEY, c, AVAILABLE), room_top, room_size)
        else:
            b.place_obj((KEY, c, AVAILABLE), key_positions[c], (1, 1))
    # the mission text is drawn from the mission space's own generator (base.py:329-330); the first of its
    # four texts (:161-166) stands for it here
    return b.finish(mission="Go pick up the key or the ball for opening the door")


def make_pool(gen, n: int, seed0: int = 0, **kw) -> Tuple[np.ndarray, np.ndarray, np.ndarray, List[str]]:
    """``n`` layouts from ``gen`` with mixin generators seeded ``seed0 .. seed0+n-1``."""
    grids, agents, metas, missions = [], [], [], []
    for s in range(n):
        g, a, m, mission = gen(mixin_rng=np.random.default_rng(seed0 + s),
                               env_rng=np.random.default_rng(seed0 + s + 100), **kw)
        grids.append(g)
        agents.append(a)
        metas.append(m)
        missions.append(mission)
    return np.stack(grids), np.stack(agents), np.stack(metas), missions
