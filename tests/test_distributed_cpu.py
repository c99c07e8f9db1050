"""Multi-rank host logic on CPU (gloo, world size 2): shard arithmetic and the statistics all-reduce."""
import os
import socket
import subprocess
import sys
import textwrap

from multigrid_amd.distributed import shard_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything_once():
    for total in (1, 7, 4096, 65537, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0
            for (s0, n0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + n0 == s1
            assert spans[-1][0] + spans[-1][1] == total
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1


WORKER = textwrap.dedent("""
    import os, sys, torch
    sys.path.insert(0, os.environ["MG_ROOT"])
    from multigrid_amd import distributed as D
    rank, world = D.init("gloo")
    assert world == 2
    start, count = D.shard_range(1000, rank, world)
    stats = torch.tensor([count, start, rank + 1, 0, 0, 40 + rank, 0, 7], dtype=torch.int64)
    D.allreduce_stats(stats)
    assert stats.tolist() == [1000, 500, 3, 0, 0, 41, 0, 14], stats.tolist()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_stats_allreduce_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MG_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out
        assert "ok" in out


def test_bench_launcher_two_ranks_gloo():
    """`python bench.py --gpus 2` as the driver starts it: the parent spawns the ranks itself; here on the CPU
    (gloo) with the kernels left out, so that the launcher, the process group and both collectives run."""
    import json
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "3",
                        "--selftest-gloo"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 9
    assert line["collective_check"] == {"ranks_seen": 2, "steps_sum": 18, "launches_sum": 24, "ok": True}
    assert line["stats_launches"] == 12  # the maximum over ranks, not their sum


def test_bench_launcher_reports_a_failed_rank():
    """A rank that dies must not leave the others waiting in a collective: the parent ends them and fails."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--selftest-gloo"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120,
                       env=dict(env, MGRID_SELFTEST_FAIL_RANK="1"))
    assert p.returncode != 0 and "{" not in p.stdout


def test_bench_finds_committed_counter_traffic():
    """bench.py prices `roofline.traffic` from the committed PMC summaries: every BASELINE batch has a record there."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for cid, envs_ in ((2, 4096), (3, 65536), (4, 65536), (4, 262144), (5, 131072), (5, 1048576)):
        name, agents = bench.CONFIGS[cid][0], bench.CONFIGS[cid][1]
        tr = bench.measured_traffic(name, envs_)
        assert tr and tr > 0, (cid, envs_)
        alg = envs_ * bench.algorithmic_bytes_per_env_step(
            {"num_agents": agents, "view_size": 7, **_grid_of(name)})
        assert 0.8 < tr / alg < 1.1, (cid, envs_, tr, alg)  # cells are 2 bytes in HBM, 3 in the algorithmic count


def _grid_of(name):
    from multigrid_amd import envs
    spec, _ = envs.make_spec(name, num_agents=2, agent_view_size=7)
    return {"width": spec["width"], "height": spec["height"]}
