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
