#!/usr/bin/env python3
"""Record one-hot observations from the UNMODIFIED reference ``OneHotObsWrapper``.

Runs only in the build container (needs ``/root/reference`` and the name-only stand-ins of
``refshim/``, see ``make_golden.py``); writes ``onehot.npz`` next to itself:

* ``cells`` / ``cells_onehot``: every (type, colour, state) triple the reference's index tables
  allow, as a ``(N, 1, 3)`` strip, and what ``OneHotObsWrapper.one_hot`` returns for it;
* ``images`` / ``images_onehot``: observation images taken from the recorded trajectories
  (views of 3, 5 and 7 cells are kept in separate arrays ``images_v*``) and their encodings;
* ``dim_sizes``: the wrapper's own ``dim_sizes`` (multigrid/wrappers.py:95), read from an instance.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, "/root/reference")

from multigrid.core.constants import Color, Direction, State, Type  # noqa: E402
from multigrid.envs.empty import EmptyEnv  # noqa: E402
from multigrid.wrappers import OneHotObsWrapper  # noqa: E402


def main():
    env = EmptyEnv(agents=2, teams={"red": 1, "blue": 1}, policies_map={}, agent_view_size=7)
    wrapped = OneHotObsWrapper(env)
    dim_sizes = np.asarray(wrapped.dim_sizes)
    assert list(dim_sizes) == [len(Type), len(Color), max(len(State), len(Direction))]
    out = {"dim_sizes": dim_sizes.astype(np.int64)}

    cells = np.array([[t, c, s] for t in range(len(Type)) for c in range(len(Color))
                      for s in range(max(len(State), len(Direction)))], dtype=np.int64)[:, None, :]
    out["cells"] = cells.astype(np.uint8)
    out["cells_onehot"] = OneHotObsWrapper.one_hot(cells, dim_sizes)

    # through the wrapper itself: reset + a few steps of the wrapped reference environment
    env._RandomMixin__np_random = np.random.default_rng(5)
    obs, _ = wrapped.reset(seed=5)
    raw = [env.gen_obs()]
    enc = [obs]
    rng = np.random.default_rng(6)
    for _ in range(12):
        obs, *_ = wrapped.step({0: int(rng.integers(0, 7)), 1: int(rng.integers(0, 7))})
        raw.append(env.gen_obs())
        enc.append(obs)
    out["wrapper_images"] = np.stack([[r[i]["image"] for i in range(2)] for r in raw]).astype(np.uint8)
    out["wrapper_onehot"] = np.stack([[e[i]["image"] for i in range(2)] for e in enc]).astype(np.uint8)

    rng = np.random.default_rng(0)
    for group, key in (("chaos_blockedunlockpickup", "v7a"), ("chaos_lockedhallway4", "v7b"),
                       ("cfg5_empty16", "v7c"), ("chaos_redbluedoors_all", "v7d"),
                       ("chaos_view5", "v5"), ("chaos_view3", "v3")):
        with np.load(os.path.join(HERE, group + ".npz")) as z:
            img, length = z["image"], z["length"]
        picks = []
        for n in range(img.shape[0]):
            for t in rng.choice(int(length[n]), size=min(4, int(length[n])), replace=False):
                picks.append(img[n, t])
        imgs = np.concatenate(picks)  # (K, V, V, 3)
        out[f"images_{key}"] = imgs
        out[f"images_{key}_onehot"] = np.stack([OneHotObsWrapper.one_hot(im.astype(np.int64), dim_sizes) for im in imgs])
    np.savez_compressed(os.path.join(HERE, "onehot.npz"), **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
