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
    stats = torch.tensor([count, start, rank + 1, 0, 0, 0, 0, 7], dtype=torch.int64)
    D.allreduce_stats(stats)
    assert stats.tolist() == [1000, 500, 3, 0, 0, 0, 0, 14], stats.tolist()
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
