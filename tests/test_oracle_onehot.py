"""The one-hot restatement (oracle/pyref.py) against encodings recorded from the reference's OneHotObsWrapper."""
import numpy as np

from golden_util import load_onehot_fixture
from oracle import pyref


def test_dim_sizes_are_the_wrappers():
    d = load_onehot_fixture()
    assert tuple(d["dim_sizes"]) == pyref.ONE_HOT_DIMS


def test_every_cell_encoding():
    d = load_onehot_fixture()
    np.testing.assert_array_equal(pyref.one_hot(d["cells"].astype(np.int64)), d["cells_onehot"])
    np.testing.assert_array_equal(pyref.one_hot_batch(d["cells"]), d["cells_onehot"])


def test_recorded_images():
    d = load_onehot_fixture()
    keys = [k for k in d if k.startswith("images_") and not k.endswith("_onehot")]
    assert len(keys) == 6
    for k in keys:
        np.testing.assert_array_equal(pyref.one_hot_batch(d[k]), d[k + "_onehot"], err_msg=k)
        for im, want in list(zip(d[k], d[k + "_onehot"]))[:8]:
            np.testing.assert_array_equal(pyref.one_hot(im.astype(np.int64)), want, err_msg=k)


def test_through_the_wrapper():
    """What reset() / step() of the wrapped reference environment returned."""
    d = load_onehot_fixture()
    np.testing.assert_array_equal(pyref.one_hot_batch(d["wrapper_images"]), d["wrapper_onehot"])


def test_competitive_v2_observation():
    """The 25-plane observation of CompetativeRedBlueDoorWrapperV2, recorded from the reference
    (tests/golden/make_golden_onehot_v2.py)."""
    import os
    from golden_util import GOLDEN_DIR
    with np.load(os.path.join(GOLDEN_DIR, "onehot_v2.npz")) as z:
        d = {k: z[k] for k in z.files}
    for v in (7, 5):
        for im, di, want in zip(d[f"images_v{v}"], d[f"directions_v{v}"], d[f"observations_v{v}"]):
            np.testing.assert_array_equal(pyref.competitive_v2_observation(im, int(di)), want)
