"""Environment registry: the in-scope names of ``CONFIGURATIONS`` (multigrid/envs/__init__.py:43-625).

``make_spec(name, teams=...)`` turns a registry entry plus the reference's constructor keyword
arguments into (spec dict for ``BatchedMultiGrid`` / ``MgConfig``, layout generator).
"""
from __future__ import annotations

from functools import partial
from typing import Callable, Dict, Tuple

from . import layouts
from .constants import (COLOR_NAMES, KIND_BLOCKEDUNLOCKPICKUP, KIND_COMPETITIVE, KIND_EMPTY, KIND_LOCKEDHALLWAY,
                        KIND_REDBLUEDOORS)

# name -> (environment family, registry kwargs)
CONFIGURATIONS: Dict[str, Tuple[str, dict]] = {
    "MultiGrid-BlockedUnlockPickup-v0": ("blockedunlockpickup", {}),
    "MultiGrid-Empty-5x5-v0": ("empty", {"size": 5}),
    "MultiGrid-Empty-Random-5x5-v0": ("empty", {"size": 5, "agent_start_pos": None}),
    "MultiGrid-Empty-6x6-v0": ("empty", {"size": 6}),
    "MultiGrid-Empty-Random-6x6-v0": ("empty", {"size": 6, "agent_start_pos": None}),
    "MultiGrid-Empty-8x8-v0": ("empty", {}),
    "MultiGrid-Empty-16x16-v0": ("empty", {"size": 16}),
    "MultiGrid-LockedHallway-2Rooms-v0": ("lockedhallway", {"num_rooms": 2}),
    "MultiGrid-LockedHallway-4Rooms-v0": ("lockedhallway", {"num_rooms": 4}),
    "MultiGrid-LockedHallway-6Rooms-v0": ("lockedhallway", {"num_rooms": 6}),
    "MultiGrid-RedBlueDoors-6x6-v0": ("redbluedoors", {"size": 6}),
    "MultiGrid-RedBlueDoors-8x8-v0": ("redbluedoors", {"size": 8}),
}


# CompetativeRedBlueDoorEnvV3 entries, multigrid/envs/__init__.py:90-614.  Every entry gives every agent the same
# reward scheme (:100-108); a caller may pass its own ``reward_schemes={"red_0": {...}, ...}``.
DEFAULT_REWARD_SCHEME = {"eliminated_opponent_sparse_reward": 0.5, "key_pickup_sparse_reward": 0.5,
                         "ball_pickup_dense_reward": 0.5, "invalid_pickup_dense_penalty": 0.001}
REWARD_SCHEME_KEYS = ("eliminated_opponent_sparse_reward", "key_pickup_sparse_reward", "ball_pickup_dense_reward",
                      "invalid_pickup_dense_penalty")


def _competitive(teams, scheme, obstacle, death_match=False, randomization=False, max_steps=None):
    kw = {"size": 8, "allow_agent_overlap": False, "has_obsticle": obstacle, "death_match": death_match,
          "teams": teams, "training_scheme": scheme, "randomization": randomization}
    if max_steps:
        kw["max_steps"] = max_steps
    return ("competitive", kw)


_V3 = "MultiGrid-CompetativeRedBlueDoor-v3-"
_TWO = {"blue": 2, "red": 2}
CONFIGURATIONS.update({
    _V3 + "DTDE-Red-Single": _competitive({"red": 1}, "DTDE", False),
    _V3 + "DTDE-Red-Single-with-Obstacle": _competitive({"red": 1}, "DTDE", True),
    **{_V3 + f"{s}-Red": _competitive({"red": 2}, s, True, randomization=True) for s in ("CTCE", "DTDE", "CTDE")},
    **{_V3 + f"{s}-Red-Eval": _competitive({"red": 2}, s, True, randomization=True, max_steps=300)
       for s in ("CTCE", "DTDE", "CTDE")},
    **{_V3 + f"{s}-1v1": _competitive({"blue": 1, "red": 1}, s, False, True, True, 300) for s in ("DTDE", "CTCE")},
    _V3 + "CTCE-2v2": _competitive(_TWO, "CTCE", True, True),
    _V3 + "DTDE-2v2": _competitive(_TWO, "DTDE", True, True),
    _V3 + "CTDE-2v2": _competitive(_TWO, "CTDE", True, False),
    **{_V3 + f"{s}-2v2-Death-Match": _competitive(_TWO, s, False, True) for s in ("CTDE", "CTCE", "DTDE")},
})


def default_teams(num_agents: int) -> Dict[str, int]:
    """One agent per colour team, wrapping round after six (``teams=`` of multigrid/base.py:106)."""
    n = len(COLOR_NAMES)
    base, extra = divmod(num_agents, n)
    if num_agents <= n:
        return {COLOR_NAMES[i]: 1 for i in range(num_agents)}
    return {COLOR_NAMES[i]: base + (1 if i < extra else 0) for i in range(n)}


def agent_colors(teams: Dict[str, int]):
    """Agent i takes its team's colour, teams in dict order (multigrid/base.py:185-203)."""
    out = []
    for name, count in teams.items():
        out += [COLOR_NAMES.index(name)] * count
    return out


def make_spec(name: str, num_agents: int | None = None, teams: Dict[str, int] | None = None,
              agent_view_size: int = 5, **kwargs) -> Tuple[dict, Callable]:
    """Spec + layout generator for a registered configuration.

    ``kwargs`` are the reference's constructor arguments (``max_steps``, ``joint_reward``,
    ``allow_agent_overlap``, ``success_termination_mode``, ``failure_termination_mode``,
    ``see_through_walls`` and the per-family ``size`` / ``room_size`` / ``num_rooms`` ...).
    ``agent_view_size`` defaults to 5 like the reference's base class (multigrid/base.py:96; no registry
    entry overrides it); the benchmark configurations all pass 7.
    """
    family, reg = CONFIGURATIONS[name]
    kw = dict(reg, **kwargs)
    if family == "competitive":  # the registry entries of this family name their teams themselves
        teams = teams if teams is not None else kw["teams"]
        kw.pop("teams")
        kw.pop("agents", None)
    if teams is None:
        teams = default_teams(num_agents if num_agents is not None else 1)
    colors = agent_colors(teams)
    a = len(colors)
    spec = dict(num_agents=a, view_size=agent_view_size,
                allow_agent_overlap=kw.pop("allow_agent_overlap", True),       # base.py:97
                see_through_walls=kw.pop("see_through_walls", False),          # base.py:95
                success_any=kw.pop("success_termination_mode", "any") == "any",  # base.py:99
                num_rooms=0)
    if family == "empty":  # multigrid/envs/empty.py:110-150
        size = kw.pop("size", 8)
        spec.update(kind=KIND_EMPTY, width=size, height=size,
                    max_steps=kw.pop("max_steps", None) or 4 * size ** 2,
                    joint_reward=kw.pop("joint_reward", False), see_through_walls=True,
                    failure_any=kw.pop("failure_termination_mode", "all") == "any",
                    mission="get to the green goal square")
        gen = partial(layouts.gen_empty, size=size, agent_colors=colors,
                      agent_start_pos=kw.pop("agent_start_pos", (1, 1)),
                      agent_start_dir=kw.pop("agent_start_dir", 0))
    elif family == "redbluedoors":  # multigrid/envs/redbluedoors.py:101-139
        size = kw.pop("size", 8)
        spec.update(kind=KIND_REDBLUEDOORS, width=2 * size, height=size,
                    max_steps=kw.pop("max_steps", None) or 20 * size ** 2,
                    joint_reward=kw.pop("joint_reward", True),
                    failure_any=kw.pop("failure_termination_mode", "any") == "any",
                    mission="open the red door then the blue door")
        gen = partial(layouts.gen_redbluedoors, size=size, agent_colors=colors)
    elif family == "blockedunlockpickup":  # multigrid/envs/blockedunlockpickup.py:103-130
        rs = kw.pop("room_size", 6)
        spec.update(kind=KIND_BLOCKEDUNLOCKPICKUP, width=2 * (rs - 1) + 1, height=rs,
                    max_steps=kw.pop("max_steps", None) or 16 * rs ** 2,
                    joint_reward=kw.pop("joint_reward", True), success_any=True,
                    failure_any=kw.pop("failure_termination_mode", "all") == "any",
                    mission="pick up the box")
        gen = partial(layouts.gen_blockedunlockpickup, agent_colors=colors, room_size=rs)
    elif family == "lockedhallway":  # multigrid/envs/locked_hallway.py:98-144
        rooms, rs = kw.pop("num_rooms", 6), kw.pop("room_size", 5)
        spec.update(kind=KIND_LOCKEDHALLWAY, width=3 * (rs - 1) + 1, height=(rooms // 2) * (rs - 1) + 1,
                    max_steps=kw.pop("max_steps", None) or 8 * (rs * rooms) ** 2,
                    joint_reward=kw.pop("joint_reward", True), num_rooms=rooms,
                    failure_any=kw.pop("failure_termination_mode", "all") == "any",
                    mission="open all the doors")
        gen = partial(layouts.gen_lockedhallway, num_rooms=rooms, agent_colors=colors, room_size=rs,
                      max_hallway_keys=kw.pop("max_hallway_keys", 1),
                      max_keys_per_room=kw.pop("max_keys_per_room", 2))
    elif family == "competitive":  # multigrid/envs/competative_red_blue_door.py:114-183
        size = kw.pop("size", 8)
        names = [f"{team}_{i}" for team, n in teams.items() for i in range(n)]  # base.py:186-190
        schemes = kw.pop("reward_schemes", None) or {n: DEFAULT_REWARD_SCHEME for n in names}
        scheme = kw.pop("training_scheme", "CTCE")
        spec.update(kind=KIND_COMPETITIVE, width=2 * size, height=size,
                    max_steps=kw.pop("max_steps", None) or 20 * size ** 2,
                    joint_reward=kw.pop("joint_reward", False),
                    failure_any=kw.pop("failure_termination_mode", "all") == "any",
                    death_match=bool(kw.pop("death_match", False)),
                    # dtde_step (DTDE / CTDE) returns one entry per agent, ctce_step one per team (:319-423)
                    per_agent_outputs=scheme != "CTCE", training_scheme=scheme,
                    reward_schemes=[[float(schemes[n][k]) for k in REWARD_SCHEME_KEYS] for n in names],
                    agent_names=names, mission="Go pick up the key or the ball for opening the door")
        gen = partial(layouts.gen_competitive, size=size, agent_colors=colors,
                      team_colors=[COLOR_NAMES.index(t) for t in teams],
                      has_obstacle=bool(kw.pop("has_obsticle", False)), randomization=bool(kw.pop("randomization", False)))
    else:
        raise KeyError(family)
    if kw:
        raise TypeError(f"unexpected arguments for {name}: {sorted(kw)}")
    return spec, gen
