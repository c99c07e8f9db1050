"""Loading of the golden trajectories recorded from the reference (tests/golden/*.npz)."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def group_names():
    names = sorted(os.path.splitext(os.path.basename(p))[0]
                   for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    # ("ctce_2v2" holds per-TEAM rewards and flags, see tests/test_oracle_golden.py::test_ctce_outputs)
    return [n for n in names if n not in ("layouts", "onehot", "onehot_v2", "ctce_2v2")]


def load_onehot_fixture():
    """One-hot encodings recorded from the reference's OneHotObsWrapper (tests/golden/make_golden_onehot.py)."""
    with np.load(os.path.join(GOLDEN_DIR, "onehot.npz")) as z:
        return {k: z[k] for k in z.files}


def load_group(name):
    """Return (spec dict, dict of arrays) of one fixture group."""
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        data = {k: z[k] for k in z.files}
    spec = json.loads(str(data.pop("spec")))
    return spec, data


def load_layout_fixture():
    with np.load(os.path.join(GOLDEN_DIR, "layouts.npz")) as z:
        return {k: z[k] for k in z.files}


INFO_KEYS = ("door_open_done", "got_eliminated_done", "eliminated_opponents_done", "eliminated_opponent_num")


def pack_info(rows):
    """The recorded info entries ``[..., 4]`` (the reference's info dict of a CompetativeRedBlueDoor step, in the order of
    INFO_KEYS) as the low six bits of the bytes the library and the C oracle return: bit 0 door_open_done, bit 1
    got_eliminated_done, bit 2 eliminated_opponents_done, bits 4-5 eliminated_opponent_num.  (Bits 6 / 7 of those bytes
    say that the agent's pickup was a rewarded key / ball pickup - INFO_MASK leaves them out; the mission texts recorded
    in the fixtures pin them.)"""
    rows = np.asarray(rows).astype(np.uint8)
    return ((rows[..., 0] != 0) | ((rows[..., 1] != 0) << 1) | ((rows[..., 2] != 0) << 2) | (rows[..., 3] << 4)).astype(np.uint8)


INFO_MASK = 0x3F
# the texts the `mission` / `reset_mission` arrays of the competitive fixtures index (tests/golden/make_golden.py)
MISSION_TEXTS = ("Go pick up the key or the ball for opening the door", "Go pick up the key", "Go move away the ball",
                 "Go open the door with the key")


def pack_info_dict(info, keys):
    """The same bytes from an info dict ``{key: {name: value}}`` for the given keys, in order."""
    return pack_info([[int(info[k][name]) for name in INFO_KEYS] for k in keys])


def team_info(members, packed):
    """Per-agent info bytes -> per-team bytes as ctce_step reports them (its entries are shared by the team's agents,
    multigrid/envs/competative_red_blue_door.py:320-328): flags or-ed, counts summed over each team's agents."""
    out = []
    for team in members:
        flags = 0
        count = 0
        for k in team:
            flags |= int(packed[k]) & 7
            count += (int(packed[k]) >> 4) & 3
        out.append(flags | (count << 4))
    return np.asarray(out, np.uint8)
