"""The reference's Python surface over the batched HIP simulator.

``MultiGridEnv`` here is a drop-in for one ``multigrid.base.MultiGridEnv`` instance
(multigrid/base.py:37): ``reset(seed) -> (obs, infos)`` and
``step({agent: action}) -> (obs, rewards, terminations, truncations, infos)`` with the same
dict-of-dict layout (integer agent keys, ``{"image", "direction", "mission"}`` observations),
``is_done()``, ``gen_obs()``, ``step_count`` / ``max_steps`` / ``num_agents``.  It drives a
``BatchedMultiGrid`` of one environment, draws layouts with the host-side generators and the
agent order from ``np_random`` exactly where the reference does, so the same generators give
the same trajectory.  ``RLlibWrapper`` and ``PettingZooWrapper`` mirror
multigrid/rllib/__init__.py:47-76 and multigrid/pettingzoo/__init__.py:35-70 by duck typing
(neither ray nor pettingzoo is required).  For throughput use ``BatchedMultiGrid`` directly.
"""
from __future__ import annotations

import ctypes
from collections import defaultdict
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, envs
from .batched import BatchedMultiGrid, StepOutput
from .constants import COLOR_NAMES


class Mission(str):
    """Mission text; like the reference's ``Mission`` (multigrid/core/mission.py:8) it exposes ``.string``."""

    @property
    def string(self) -> str:
        return str(self)


class BoxSpec:
    """Stand-in for ``gymnasium.spaces.Box`` where gymnasium is not installed: ``low / high / shape / dtype``."""

    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(((x >= self.low) & (x <= self.high)).all())

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


class DiscreteSpec:
    """Stand-in for ``gymnasium.spaces.Discrete``."""

    def __init__(self, n: int):
        self.n, self.shape, self.dtype = int(n), (), np.dtype(np.int64)

    def contains(self, x) -> bool:
        return 0 <= int(x) < self.n

    def __repr__(self):
        return f"Discrete({self.n})"


def _spaces():
    """gymnasium's own space classes when gymnasium can be imported, the stand-ins above otherwise."""
    try:
        from gymnasium import spaces  # noqa: PLC0415
        return spaces.Box, spaces.Discrete, spaces.Dict
    except Exception:  # gymnasium absent (or a name-only stand-in without these classes)
        return BoxSpec, DiscreteSpec, dict


def _gpu_addresses(pinned: torch.Tensor) -> bool:
    """Whether the current GPU addresses this pinned host tensor at its host address (hipHostGetDevicePointer), so that
    kernels can be handed ``data_ptr()`` as it is."""
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        dev_ptr = ctypes.c_void_p()
        rc = hip.hipHostGetDevicePointer(ctypes.byref(dev_ptr), ctypes.c_void_p(pinned.data_ptr()), ctypes.c_uint(0))
        return rc == 0 and dev_ptr.value == pinned.data_ptr()
    except (OSError, AttributeError):
        return False


class MultiGridEnv:
    metadata = {"render_modes": [], "render_fps": 20}

    def __init__(self, name: str, agents: Optional[int] = None, teams: Optional[Dict[str, int]] = None,
                 agent_view_size: int = 5, device: int | str = 0, layout_rng: Optional[np.random.Generator] = None,
                 **kwargs):
        self.name = name
        self.spec, self._gen = envs.make_spec(name, num_agents=agents, teams=teams,
                                              agent_view_size=agent_view_size, **kwargs)
        self.teams = teams if teams is not None else envs.default_teams(self.spec["num_agents"])
        self.num_agents = self.spec["num_agents"]
        self.max_steps = self.spec["max_steps"]
        self.width, self.height = self.spec["width"], self.spec["height"]
        # Everything a step returns lives in ONE block (the launch's outputs are views into it, the agent rows of the
        # state it leaves and the rewards as doubles among them), actions and agent order in another.  Both blocks are
        # pinned host memory that the GPU addresses directly: the kernel reads its 2 A input bytes from it and writes
        # its few hundred output bytes to it over the bus, so a step is one launch and one wait, with no copy
        # operations queued around it.  (Where the runtime does not map pinned memory at its host address, the blocks
        # are device memory and travel by one upload and one download per step.)
        a, v = self.num_agents, self.spec["view_size"]
        dev = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        n_img = a * v * v * 3
        self._off = dict(image=0, direction=n_img, terminated=n_img + a, truncated=n_img + 2 * a, info=n_img + 3 * a,
                         reward=(n_img + 4 * a + 7) // 8 * 8)  # doubles, 8-byte aligned
        self._off["rows"] = self._off["reward"] + 8 * a  # agent rows of the post-step state, 8 bytes per agent
        total = self._off["rows"] + 8 * a
        with torch.cuda.device(dev):
            self._out_host = torch.zeros(total, dtype=torch.uint8).pin_memory()
            self._in_host = torch.zeros((2, 1, a), dtype=torch.uint8).pin_memory()
            self._zero_copy = _gpu_addresses(self._out_host) and _gpu_addresses(self._in_host)
        self._out_np = self._out_host.numpy()
        self._in_np = self._in_host.numpy()
        self._out_dev = self._out_host if self._zero_copy else torch.zeros(total, dtype=torch.uint8, device=dev)
        self._in_dev = self._in_host if self._zero_copy else torch.zeros((2, 1, a), dtype=torch.uint8, device=dev)
        o, blk = self._off, self._out_dev
        packed = StepOutput(image=blk[:n_img].view(1, a, v, v, 3), direction=blk[o["direction"]:o["direction"] + a].view(1, a),
                            reward=None,
                            terminated=blk[o["terminated"]:o["terminated"] + a].view(1, a),
                            truncated=blk[o["truncated"]:o["truncated"] + a].view(1, a),
                            rows=blk[o["rows"]:].view(1, a, 8),
                            reward64=blk[o["reward"]:o["rows"]].view(torch.float64).view(1, a),
                            info=blk[o["info"]:o["info"] + a].view(1, a) if self.spec["kind"] == _lib.KIND_COMPETITIVE else None)
        self._sim = BatchedMultiGrid(self.spec, 1, device=dev, autoreset=_lib.AUTORESET_NONE, out=packed)
        self._act_ptr = ctypes.c_void_p(self._in_dev[0].data_ptr())
        self._order_ptr = ctypes.c_void_p(self._in_dev[1].data_ptr())
        self._layout0 = torch.zeros(1, dtype=torch.int32, device=dev)
        # RandomMixin keeps the generator that existed at construction time (multigrid/base.py:153);
        # gym.Env.np_random is the same object until reset(seed) replaces it.
        self._np_random = layout_rng if layout_rng is not None else np.random.default_rng()
        self._layout_rng = self._np_random
        self.mission = Mission(self.spec["mission"])
        self.agent_mission = self.mission
        self.step_count = 0
        self._rows = np.zeros((self.num_agents, 8), dtype=np.uint8)
        self.agent_names = []
        for team, n in self.teams.items():
            self.agent_names += [f"{team}_{i}" for i in range(n)]

    # -- spaces -------------------------------------------------------------------------------------
    def agent_observation_space(self, i: int = 0):
        """One agent's observation space (multigrid/core/agent.py:116-128): image Box(0, 255, (V, V, 3), int),
        direction Discrete(4); the mission space is left out (text, host side only)."""
        box, discrete, dct = _spaces()
        v = self.spec["view_size"]
        return dct({"image": box(0, 255, (v, v, 3), np.int64), "direction": discrete(4)})

    def agent_action_space(self, i: int = 0):
        """multigrid/core/agent.py:144: Discrete(len(Action))."""
        return _spaces()[1](7)

    @property
    def observation_space(self):
        """Joint space keyed ``"{color}_{team_index}"`` (multigrid/base.py:262-265)."""
        return _spaces()[2]({n: self.agent_observation_space(i) for i, n in enumerate(self.agent_names)})

    @property
    def action_space(self):
        """multigrid/base.py:285."""
        return _spaces()[2]({n: self.agent_action_space(i) for i, n in enumerate(self.agent_names)})

    # -- gym.Env ------------------------------------------------------------------------------------
    @property
    def np_random(self) -> np.random.Generator:
        return self._np_random

    def reset(self, seed: Optional[int] = None, **kwargs):
        """multigrid/base.py:310-362."""
        if seed is not None:
            self._np_random = np.random.default_rng(seed)
        grid, agents, meta, mission = self._gen(mixin_rng=self._layout_rng, env_rng=self._np_random)
        self.mission = Mission(mission)
        self.agent_mission = self._sample_agent_mission(seed)
        self._sim.upload_layouts(grid[None], agents[None], meta[None])
        self._sim.reset(layout_idx=self._layout0)
        self.step_count = 0
        return self._obs_dict(self._fetch()), defaultdict(dict)

    def _sample_agent_mission(self, seed: Optional[int]) -> Mission:
        """The mission the observations carry: ``reset`` seeds the mission space with the reset seed, draws one sample
        and hands it to every agent (multigrid/base.py:329-332).  A mission space of one text gives that text.
        BlockedUnlockPickup's space is colours x (box, key) (blockedunlockpickup.py:117-120), so its agents are told to
        pick up a RANDOM coloured box or key, whatever the environment's own ``mission`` (set afterwards by
        ``_gen_grid``, :158) says.  Parity unpinned: the draw is gymnasium's ``MultiDiscrete.sample``, which is not
        importable here; restated from its published source - ``(np_random.random(nvec.shape) * nvec).astype(dtype)``
        on the generator ``Space.seed(seed)`` makes, ``Generator(PCG64(SeedSequence(seed)))``."""
        if self.spec["kind"] != _lib.KIND_BLOCKEDUNLOCKPICKUP:
            return self.mission
        rng = np.random.default_rng(seed)  # (seed None: fresh entropy, as Space.seed(None))
        idx = (rng.random(2) * np.array([len(COLOR_NAMES), 2])).astype(np.int64)
        return Mission(f"pick up the {COLOR_NAMES[idx[0]]} {('box', 'key')[idx[1]]}")

    def step(self, actions: Dict[int, int]):
        """multigrid/base.py:364-408 plus the environment's own step hook."""
        a = self.num_agents
        acts = [int(actions[i]) for i in range(a)]
        # base.py:457-461
        order = np.zeros(1, dtype=np.uint8) if a == 1 else self._np_random.random(size=a).argsort()
        got = self._launch_step(acts, order)
        # the doubles _reward() computes (multigrid/base.py:655-659; LockedHallway adds one per door,
        # locked_hallway.py:207-211), worked out on the device in the reference's order of operations
        rew = got["reward"]
        term, trunc = got["terminated"].astype(bool), got["truncated"].astype(bool)
        rewards = {i: (float(rew[i]) if rew[i] != 0 else 0) for i in range(a)}
        terminations = {i: np.bool_(term[i]) for i in range(a)}
        truncations = {i: bool(trunc[i]) for i in range(a)}
        return self._obs_dict(got), rewards, terminations, truncations, defaultdict(dict)

    def gen_obs(self):
        """multigrid/base.py:410-438."""
        self._sim.gen_obs()
        return self._obs_dict(self._fetch())

    def is_done(self) -> bool:
        """multigrid/base.py:587-596."""
        return self.step_count >= self.max_steps or bool(self._terminated.all())

    @property
    def _terminated(self) -> np.ndarray:
        """``agent_states.terminated`` (not always what ``step`` returned: LockedHallway's all-doors ending sets only
        the returned flags, locked_hallway.py:213-215): column 3 of the agent rows that came back with the step."""
        return self._rows[:, 3].astype(bool)

    def close(self):
        self._sim.close()

    # -- helpers ------------------------------------------------------------------------------------
    def _launch_step(self, acts, order):
        """One upload (actions and order), the step launch, one download of everything it returned."""
        self._in_np[0, 0, :] = acts
        self._in_np[1, 0, :] = order
        sim = self._sim
        if not self._zero_copy:
            self._in_dev.copy_(self._in_host, non_blocking=True)
        rc = sim._mg_step(sim._h, self._act_ptr, self._order_ptr, ctypes.byref(sim._mgout), sim._stream())
        if rc != 0:
            _lib.check(rc, sim._h)
        self.step_count += 1
        return self._fetch()

    def _fetch(self):
        """The packed output block of the last launch as host arrays (copies: the pinned block is reused)."""
        # (the agent rows of the state the launch left behind - is_done, agent_states - travel in the same block)
        if not self._zero_copy:
            self._out_host.copy_(self._out_dev, non_blocking=True)
        torch.cuda.current_stream(self._sim.device).synchronize()
        a, v, o, h = self.num_agents, self.spec["view_size"], self._off, self._out_np
        self._rows = h[o["rows"]:o["rows"] + 8 * a].reshape(a, 8).copy()
        return dict(image=h[:o["direction"]].reshape(a, v, v, 3).astype(np.int64),
                    direction=h[o["direction"]:o["direction"] + a].astype(np.int64),
                    terminated=h[o["terminated"]:o["terminated"] + a].copy(),
                    truncated=h[o["truncated"]:o["truncated"] + a].copy(),
                    info=h[o["info"]:o["info"] + a].copy(),
                    reward=h[o["reward"]:o["reward"] + 8 * a].view(np.float64).copy())

    def _obs_dict(self, got):
        img, direction = got["image"], got["direction"]
        return {i: {"image": img[i], "direction": direction[i], "mission": self.agent_mission}
                for i in range(self.num_agents)}

    @property
    def agent_states(self) -> np.ndarray:
        """(A, 9) rows in the reference's column order (multigrid/core/agent.py:266-275)."""
        rows = self._sim.get_state()["agents"][0].cpu().numpy().astype(np.int64)  # (the state may have been set since)
        out = np.zeros((self.num_agents, 9), dtype=np.int64)
        out[:, 0], out[:, 1], out[:, 2] = 10, rows[:, 7], rows[:, 2]
        out[:, 3], out[:, 4], out[:, 5] = rows[:, 0], rows[:, 1], rows[:, 3]
        out[:, 6:9] = rows[:, 4:7]
        return out

    @property
    def grid_state(self) -> np.ndarray:
        """(W, H, 3) ``grid.state`` (multigrid/core/grid.py:53)."""
        return self._sim.get_state()["grid"][0].cpu().numpy().astype(np.int64)


# the texts of CompetativeRedBlueDoorEnvV3's mission space, in its order (competative_red_blue_door.py:158-168)
COMPETITIVE_MISSIONS = ("Go pick up the key or the ball for opening the door", "Go pick up the key",
                        "Go move away the ball", "Go open the door with the key")


def competitive_missions_after_step(missions, info, order, rows):
    """The agents' mission texts (indices into ``COMPETITIVE_MISSIONS``) after a step's handlers have run
    (``_default_extra_action_handle_steps``, competative_red_blue_door.py:438-468; the observations of the NEXT step
    carry them).  ``info``: the step's info bytes, bit 6 / 7 = the agent's pickup was a rewarded key / ball pickup;
    ``order``: the order in which the agents were handled; ``rows``: the agent rows the step left."""
    missions = list(missions)
    key_type = 5  # Type.key
    for i in order:
        mates = [k for k in range(len(missions)) if k != i and rows[k][7] == rows[i][7]]
        if info[i] & 0x40:  # :438-443
            missions[i] = 3
            for k in mates:
                missions[k] = 2
        elif info[i] & 0x80:  # :456-468
            missions[i] = 2
            for k in mates:
                missions[k] = 3 if rows[k][4] == key_type and rows[k][5] == rows[k][7] else 1
    return missions


class CompetativeRedBlueDoorEnv(MultiGridEnv):
    """Drop-in for ``CompetativeRedBlueDoorEnvV3`` with its default step handlers (no policy objects):
    multigrid/envs/competative_red_blue_door.py:24.  With ``training_scheme`` "DTDE" / "CTDE" everything is keyed by
    agent name (``"red_0"``), actions ``{name: action}`` (:372-423); with "CTCE" by team, actions
    ``{team: [action of each team member]}``, observations ``{team: [{"agent_id", "image", "direction", "mission"}]}``,
    rewards summed and flags and-ed per team (:319-370).  Agents act in the order of the keys of the actions dict
    (:624-643).  The step's info dict (:320-328, :380-388) is rebuilt from the bytes the launch returns, and so are the
    mission texts the reference rewrites after rewarded key and ball pickups (:438-468).  The text the agents start an
    episode with is a sample of the mission space (multigrid/base.py:329-332), drawn as ``_sample_agent_mission``
    says (parity unpinned there)."""

    INFO_KEYS = ("door_open_done", "got_eliminated_done", "eliminated_opponents_done", "eliminated_opponent_num")

    @staticmethod
    def _info_entry(flags: int, count: int) -> dict:
        return {"door_open_done": bool(flags & 1), "got_eliminated_done": bool(flags & 2),
                "eliminated_opponents_done": bool(flags & 4), "eliminated_opponent_num": int(count)}

    def __init__(self, name: str, device: int | str = 0, layout_rng: Optional[np.random.Generator] = None, **kwargs):
        spec, _ = envs.make_spec(name, **kwargs)
        if spec["kind"] != _lib.KIND_COMPETITIVE:
            raise ValueError(f"{name} is not a CompetativeRedBlueDoor configuration")
        self.training_scheme = spec["training_scheme"]
        super().__init__(name, device=device, layout_rng=layout_rng, teams=kwargs.pop("teams", None), **kwargs)
        self.agent_names = list(self.spec["agent_names"])
        self.team_members: Dict[str, list] = {}
        for i, n in enumerate(self.agent_names):
            self.team_members.setdefault(n.rsplit("_", 1)[0], []).append(i)
        self.teams = {t: len(m) for t, m in self.team_members.items()}
        self._missions = [0] * self.num_agents  # indices into COMPETITIVE_MISSIONS; drawn at reset

    def _format_obs(self, obs):
        if self.training_scheme == "CTCE":  # gen_obs, :298-309
            return {team: [dict(obs[k], agent_id=j) for j, k in enumerate(members)]
                    for team, members in self.team_members.items()}
        return {self.agent_names[k]: obs[k] for k in range(self.num_agents)}  # :310-316

    def _sample_agent_mission(self, seed: Optional[int]) -> Mission:
        rng = np.random.default_rng(seed)  # MissionSpace.seed(seed), then one sample for every agent (base.py:329-332)
        first = int((rng.random(1) * np.array([len(COMPETITIVE_MISSIONS)])).astype(np.int64)[0])
        self._missions = [first] * self.spec["num_agents"]
        return Mission(COMPETITIVE_MISSIONS[first])

    def _obs_dict(self, got):
        obs = super()._obs_dict(got)
        for k, text in enumerate(self._missions):
            obs[k]["mission"] = Mission(COMPETITIVE_MISSIONS[text])
        return obs

    def reset(self, seed: Optional[int] = None, **kwargs):
        obs, infos = super().reset(seed=seed, **kwargs)
        return self._format_obs(obs), infos

    def gen_obs(self):
        return self._format_obs(super().gen_obs())

    def step(self, actions):
        a = self.num_agents
        acts = np.full((1, a), 6, dtype=np.uint8)  # agents the caller leaves out do nothing ("done")
        order = []
        if self.training_scheme == "CTCE":  # :624-629
            for team, team_actions in actions.items():
                for j, act in enumerate(team_actions):
                    k = self.team_members[team][j]
                    acts[0, k] = int(act)
                    order.append(k)
        else:  # :631-641
            for name, act in actions.items():
                k = self.agent_names.index(name)
                acts[0, k] = int(act)
                order.append(k)
        order += [k for k in range(a) if k not in order]
        got = self._launch_step(acts[0], order)
        rew = got["reward"]  # doubles, accumulated on the device as the reference accumulates them
        term, trunc = got["terminated"].astype(bool), got["truncated"].astype(bool)
        obs = self._format_obs(self._obs_dict(got))
        bits = got["info"]  # per agent: flags in bits 0-2, eliminated_opponent_num in bits 4-5, pickup events in 6-7
        # (the observations above carry the texts as they were before this step's handlers ran, :307,315)
        self._missions = competitive_missions_after_step(self._missions, bits, order, self._rows)
        if self.training_scheme == "CTCE":  # :346-368; the info entries are the teams' (:320-328)
            rewards = {t: float(sum(rew[k] for k in m)) for t, m in self.team_members.items()}
            terminations = {t: bool(all(term[k] for k in m)) for t, m in self.team_members.items()}
            truncations = {t: bool(all(trunc[k] for k in m)) for t, m in self.team_members.items()}
            info = {t: self._info_entry(int(np.bitwise_or.reduce(bits[m]) & 7), int(((bits[m] >> 4) & 3).sum()))
                    for t, m in self.team_members.items()}
        else:  # :408-421
            rewards = {n: float(rew[k]) for k, n in enumerate(self.agent_names)}
            terminations = {n: bool(term[k]) for k, n in enumerate(self.agent_names)}
            truncations = {n: bool(trunc[k]) for k, n in enumerate(self.agent_names)}
            info = {n: self._info_entry(int(bits[k]) & 7, (int(bits[k]) >> 4) & 3) for k, n in enumerate(self.agent_names)}
        return obs, rewards, terminations, truncations, info


class RLlibWrapper:
    """``MultiAgentEnv`` view: ``__all__`` keys added to terminations and truncations
    (multigrid/rllib/__init__.py:71-76)."""

    def __init__(self, env: MultiGridEnv):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def get_agent_ids(self):
        return set(range(self.env.num_agents))

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, actions):
        obs, rewards, terminations, truncations, infos = self.env.step(actions)
        terminations["__all__"] = all(terminations.values())
        truncations["__all__"] = all(truncations.values())
        return obs, rewards, terminations, truncations, infos


class PettingZooWrapper:
    """``ParallelEnv`` view (multigrid/pettingzoo/__init__.py:35-70)."""

    def __init__(self, env: MultiGridEnv):
        self.env = env
        self.reset = env.reset
        self.step = env.step
        self.close = env.close

    @property
    def agents(self):
        if self.env.is_done():
            return []
        return [i for i in range(self.env.num_agents) if not self.env._terminated[i]]

    @property
    def possible_agents(self):
        return list(range(self.env.num_agents))

    # The reference indexes its name-keyed joint spaces with the integer agent ids it uses everywhere else
    # (multigrid/pettingzoo/__init__.py:66-70), which raises KeyError there; here an id selects that agent's space.
    @property
    def observation_spaces(self):
        return {i: self.env.agent_observation_space(i) for i in range(self.env.num_agents)}

    @property
    def action_spaces(self):
        return {i: self.env.agent_action_space(i) for i in range(self.env.num_agents)}

    def observation_space(self, agent_id):
        return self.env.agent_observation_space(int(agent_id))

    def action_space(self, agent_id):
        return self.env.agent_action_space(int(agent_id))


def make(name: str, **kwargs) -> MultiGridEnv:
    """``gym.make``-style constructor over the registry (multigrid/envs/__init__.py:43-625)."""
    if envs.CONFIGURATIONS[name][0] == "competitive":
        return CompetativeRedBlueDoorEnv(name, **kwargs)
    return MultiGridEnv(name, **kwargs)
