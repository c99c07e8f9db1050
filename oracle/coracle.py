"""ctypes front-end of ``oracle/mg_oracle.c`` (TEST INFRASTRUCTURE - see that file's header).

``BatchOracle`` keeps ``E`` environments in numpy arrays laid out exactly like the
reference's ``grid.state`` / ``agent_states`` and steps them all with one C call.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmg_oracle.so")

SPEC_FIELDS = ("kind", "width", "height", "num_agents", "view_size", "max_steps",
               "see_through_walls", "allow_agent_overlap", "joint_reward",
               "success_any", "failure_any", "num_rooms")


class CSpec(ctypes.Structure):
    _fields_ = [(name, ctypes.c_int32) for name in SPEC_FIELDS] + [
        ("death_match", ctypes.c_int32), ("per_agent_outputs", ctypes.c_int32),
        ("reward_schemes", (ctypes.c_double * 4) * 8)]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "mg_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-B", "-C", HERE, "libmg_oracle.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        try:
            build()  # (compiles only when the library is missing or older than its source)
        except (OSError, subprocess.CalledProcessError):
            if not os.path.exists(LIB_PATH):  # (no compiler here: the library that travelled with the tree is used)
                raise
        _lib = ctypes.CDLL(LIB_PATH)
        p = ctypes.c_void_p
        _lib.mgo_gen_obs.argtypes = [ctypes.POINTER(CSpec), ctypes.c_int64, p, p, p]
        _lib.mgo_gen_obs.restype = None
        _lib.mgo_step.argtypes = [ctypes.POINTER(CSpec), ctypes.c_int64] + [p] * 13
        _lib.mgo_step.restype = None
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class BatchOracle:
    def __init__(self, spec: dict, num_envs: int):
        self.spec = dict(spec)
        self.cspec = CSpec(**{k: int(spec[k]) for k in SPEC_FIELDS})
        self.cspec.death_match = int(bool(spec.get("death_match", False)))
        self.cspec.per_agent_outputs = int(bool(spec.get("per_agent_outputs", True)))
        for i, row in enumerate(spec.get("reward_schemes") or []):
            for j, v in enumerate(row):
                self.cspec.reward_schemes[i][j] = float(v)
        self.E = int(num_envs)
        w, h, a, v = (int(spec[k]) for k in ("width", "height", "num_agents", "view_size"))
        self.A, self.V = a, v
        e = self.E
        self.grid = np.zeros((e, w, h, 3), np.uint8)
        self.agents = np.zeros((e, a, 8), np.uint8)
        self.meta = np.zeros((e, 4), np.uint8)
        self.steps = np.zeros(e, np.int32)
        self.flags = np.zeros(e, np.uint8)
        self.image = np.zeros((e, a, v, v, 3), np.uint8)
        self.direction = np.zeros((e, a), np.uint8)
        self.reward = np.zeros((e, a), np.float64)
        self.terminated = np.zeros((e, a), np.uint8)
        self.truncated = np.zeros((e, a), np.uint8)
        # CompetativeRedBlueDoor: the step's info dict per agent, packed (bit 0 door_open_done, bit 1
        # got_eliminated_done, bit 2 eliminated_opponents_done, bits 4-7 eliminated_opponent_num)
        self.info = np.zeros((e, a), np.uint8)

    def load(self, env_ids, grid, agents, meta):
        """Reset the given environments to the given start states (base.py:310-362)."""
        self.grid[env_ids] = grid
        self.agents[env_ids] = agents
        self.meta[env_ids] = meta
        self.steps[env_ids] = 0
        self.flags[env_ids] = 0

    def gen_obs(self) -> np.ndarray:
        lib().mgo_gen_obs(ctypes.byref(self.cspec), self.E, _ptr(self.grid), _ptr(self.agents),
                          _ptr(self.image))
        return self.image

    def step(self, actions: np.ndarray, order=None):
        actions = np.ascontiguousarray(actions, np.uint8)
        assert actions.shape == (self.E, self.A)
        if order is not None:
            order = np.ascontiguousarray(order, np.uint8)
            assert order.shape == (self.E, self.A)
        lib().mgo_step(ctypes.byref(self.cspec), self.E, _ptr(self.grid), _ptr(self.agents),
                       _ptr(self.meta), _ptr(self.steps), _ptr(self.flags), _ptr(actions),
                       _ptr(order), _ptr(self.image), _ptr(self.direction), _ptr(self.reward),
                       _ptr(self.terminated), _ptr(self.truncated), _ptr(self.info))
        return self.image, self.direction, self.reward, self.terminated, self.truncated
