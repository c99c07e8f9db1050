"""Experiment: one batch held as S shards (own handle, own HIP stream each) that step without waiting for each
other.  A one-round launch is bound by the life of a wavefront plus the launch floor, not by bytes; shards on
separate streams overlap those.  Prints the whole-batch step time for S = 1, 2, 4, 8.

    python tools/exp_streams.py [config ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib, envs, layouts  # noqa: E402
from multigrid_amd.batched import BatchedMultiGrid  # noqa: E402


def run(cfg_id, shards, steps=1504, warm=1008, graph=False, num_envs=0):
    name, agents, e0, _, kw = bench.CONFIGS[cfg_id]
    num_envs = num_envs or e0
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7, **kw)
    dev = torch.device("cuda", 0)
    pool = layouts.make_pool(gen, 256, seed0=0)[:3]
    es = num_envs // shards
    hs, streams, acts = [], [], []
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    for s in range(shards):
        h = BatchedMultiGrid(spec, es, device=dev, autoreset=_lib.AUTORESET_SAME_STEP, env_id_offset=s * es, seed=1234,
                             batch_envs=num_envs if os.environ.get("EXP_BATCH_HINT", "1") == "1" else 0)
        h.upload_layouts(*pool)
        h.reset()
        h.set_state(steps=torch.randint(0, spec["max_steps"], (es,), generator=g, device=dev, dtype=torch.int32))
        hs.append(h)
        streams.append(torch.cuda.Stream(device=dev))
        acts.append(torch.randint(0, 7, (16, es, agents), generator=g, device=dev, dtype=torch.uint8))
    torch.cuda.synchronize(dev)

    graphs = []

    def go(n):
        if graphs:
            for t in range(n // 16):
                for s in range(shards):
                    with torch.cuda.stream(streams[s]):
                        graphs[s].replay()
            return
        for t in range(n):
            for s in range(shards):
                with torch.cuda.stream(streams[s]):
                    hs[s].step(acts[s][t % 16])

    if graph:
        go(32)
        torch.cuda.synchronize(dev)
        for s in range(shards):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=streams[s]):
                for t in range(16):
                    hs[s].step(acts[s][t])
            graphs.append(gr)
        torch.cuda.synchronize(dev)

    go(warm)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    go(steps)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    b = bench.algorithmic_bytes_per_env_step(spec)
    print(f"cfg{cfg_id} E={num_envs} tile={os.environ.get('MGRID_TILE', 'auto')} graph={int(graph)} shards={shards}: {wall / steps * 1e6:7.2f} us per whole-batch step "
          f"(host issue {t_issue / steps * 1e6:6.2f} us), frac {num_envs * b / (wall / steps) / 8e12:.3f}", flush=True)
    for h in hs:
        h.close()


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("cfgs", type=int, nargs="*", default=[2, 3, 4])
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--envs", type=int, default=0)
    ap.add_argument("--shards", type=int, nargs="*", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=1504)
    a = ap.parse_args()
    for c in a.cfgs:
        for s in a.shards:
            run(c, s, steps=a.steps, warm=max(64, min(1008, a.steps)), graph=a.graph, num_envs=a.envs)
