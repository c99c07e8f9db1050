"""bench.py's line hygiene, checked without a GPU: a field named `*frac*` is a part of a whole and never exceeds 1;
the committed counter summaries are found for plain and for stream-shard workloads."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_fraction_check_finds_nested_offenders():
    good = {"roofline": {"frac": 0.74, "frac_counter_bytes": None, "achieved_over_measured_copy": 1.2},
            "other_configs": [{"roofline_frac": 0.5}, {"roofline_frac": 1.0}]}
    assert bench.check_fractions(good) is good
    bad = {"roofline": {"frac": 0.74}, "other_configs": [{"roofline_frac": 0.5}, {"roofline_frac_x": 1.012}]}
    with pytest.raises(ValueError, match=r"other_configs\[1\]\.roofline_frac_x"):
        bench.check_fractions(bad)
    with pytest.raises(ValueError):
        bench.check_fractions({"frac": -0.1})


def test_no_time_ratio_is_named_a_fraction():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "roofline_frac_at_one_step_bytes" not in src and "frac_of_measured_copy" not in src


def test_last_committed_bench_line_passes_when_present():
    """The bench lines kept under profiles/ since the rule exists (r03_bench*.json onwards) obey it."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_bench*.json"))):
        line = json.load(open(path))
        bench.check_fractions(line.get("parsed", line))


def test_counter_traffic_lookup():
    name5 = bench.CONFIGS[5][0]
    one = bench.measured_traffic(name5, 131072)
    two = bench.measured_traffic(name5, 131072, stream_shards=2)
    assert one and two and 0.9 < two / one < 1.1  # two slices move what one launch of the whole batch moves
    assert bench.measured_traffic(name5, 12345) is None
    assert 1976 * 131072 * 0.95 < one < bench.algorithmic_bytes_per_env_step(
        dict(num_agents=8, width=16, height=16, view_size=7)) * 131072
