"""The multi-rank path of bench.py on a box with one GPU: two ranks share GPU 0 and talk over gloo
(MGRID_BENCH_REHEARSE=1).  Sharding by env_id_offset, barriers, maximum over the ranks, the statistics all-reduce
and its check are the code `python bench.py --gpus N` runs on a node; only the backend and the device choice differ."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_on_one_gpu(scaling):
    env = dict(os.environ, MGRID_BENCH_REHEARSE="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "2", "--steps", "60",
                        "--warmup", "10", "--no-cpu", "--scaling", scaling], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["rccl_ranks"] is None
    assert line["backend"] == "gloo" and line["config"]["layout_pool"].startswith("mg_generate_layouts")
    cc = line["collective_check"]
    assert cc["ok"] and cc["ranks_seen"] == 2 and cc["steps_sum"] == 120
    per_gpu = 4096 if scaling == "weak" else 2048
    assert line["config"]["envs_per_gpu"] == per_gpu
    assert line["value"] > 0 and "rehearsal" in line


@pytest.mark.gpu
def test_rccl_group_of_one_rank():
    """`--force-process-group` on one GPU: the process joins an RCCL ("nccl") group of one rank and the barriers, the
    maximum over ranks, the statistics all-reduce and the check vector all go through it on GPU tensors - the calls a
    multi-GPU run makes, executed on the hardware there is."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MGRID_BENCH_REHEARSE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-process-group",
                        "--config", "2", "--steps", "60", "--warmup", "10", "--no-cpu", "--no-extra", "--pool", "device"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["backend"] == "nccl" and line["rccl_ranks"] == 1 and line["n_gpus"] == 1
    cc = line["collective_check"]
    assert cc["ok"] and cc["ranks_seen"] == 1 and cc["steps_sum"] == 60
    assert line["config"]["layout_pool"].startswith("mg_generate_layouts")
    assert line["value"] > 0 and line["episode_stats"]["episodes"] >= 0
