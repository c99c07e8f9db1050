#!/usr/bin/env python3
"""Record the ``(V, V, 25)`` observation of the UNMODIFIED reference ``CompetativeRedBlueDoorWrapperV2``
(multigrid/wrappers.py:326-414: a direction plane in front of an image that ``observation()`` one-hot encodes twice).

Runs only in the build container (needs ``/root/reference`` and ``refshim/``); writes ``onehot_v2.npz``:
``images`` uint8 (N, V, V, 3) and ``directions`` uint8 (N,) as the single-agent environment's ``gen_obs`` returned them
(a reset and random steps of MultiGrid-CompetativeRedBlueDoor-v3-DTDE-Red-Single-with-Obstacle), and ``observations``
uint8 (N, V, V, 25), what the wrapper's ``observation()`` made of each.
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, "/root/reference")

from multigrid.envs import CONFIGURATIONS  # noqa: E402
from multigrid.wrappers import CompetativeRedBlueDoorWrapperV2  # noqa: E402


def main():
    out = {"images": [], "directions": [], "observations": []}
    for view in (7, 5):
        cls, kw = CONFIGURATIONS["MultiGrid-CompetativeRedBlueDoor-v3-DTDE-Red-Single-with-Obstacle"]
        env = cls(policies_map={}, agent_view_size=view, **kw)
        wrapper = CompetativeRedBlueDoorWrapperV2(env)
        env._RandomMixin__np_random = np.random.default_rng(3)
        env.reset(seed=4)
        rng = np.random.default_rng(5)
        imgs, dirs, obs_out = [], [], []
        for t in range(60):
            raw = env.gen_obs()
            imgs.append(np.asarray(raw["red_0"]["image"]).astype(np.uint8))
            dirs.append(int(raw["red_0"]["direction"]))
            obs_out.append(np.asarray(wrapper.observation(copy.deepcopy(raw))).astype(np.uint8))
            env.step({"red_0": int(rng.choice(7, p=[0.15, 0.15, 0.4, 0.1, 0.05, 0.1, 0.05]))})
        out[f"images_v{view}"] = np.stack(imgs)
        out[f"directions_v{view}"] = np.array(dirs, np.uint8)
        out[f"observations_v{view}"] = np.stack(obs_out)
        assert out[f"observations_v{view}"].shape == (60, view, view, 25)
    for k in ("images", "directions", "observations"):
        out.pop(k)
    path = os.path.join(HERE, "onehot_v2.npz")
    np.savez_compressed(path, **out)
    print("onehot_v2.npz", {k: v.shape for k, v in out.items()}, f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
