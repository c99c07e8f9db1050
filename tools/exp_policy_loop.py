"""A closed loop - observations -> policy -> actions -> step - on one stream over the whole batch, and pipelined
over two stream shards: each slice's policy runs on the slice's stream, so one slice's policy overlaps the other
slice's step and no launch boundary stalls both.  The policy is a rule on three cells of the observation (a handful
of small tensor operations), deterministic, so both forms must end in the same state.

    python tools/exp_policy_loop.py [config ...]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multigrid_amd import _lib, envs, layouts  # noqa: E402
from multigrid_amd.batched import BatchedMultiGrid  # noqa: E402
from multigrid_amd.sharded import StreamShardedMultiGrid  # noqa: E402


def policy(image, direction, t, out):
    """uint8 actions into `out`: forward when the cell ahead is empty, otherwise a turn picked from the image."""
    ahead = image[:, :, 3, 5, 0]  # view column 3, row 5: the cell in front of the agent (it stands at row 6)
    mix = image[:, :, 3, 6, 1].to(torch.int32) + image[:, :, 2, 6, 0] + direction + t  # own cell, cell to the left
    out.copy_(torch.where(ahead == 1, 2, mix % 2))
    return out


def closed_loop(cfg_id, shards, iters, graph_iters=8, num_envs=0):
    name, agents, e0, _, kw = bench.CONFIGS[cfg_id]
    e = num_envs or e0
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7, **kw)
    dev = torch.device("cuda", 0)
    pool = layouts.make_pool(gen, 256, seed0=0)[:3]
    common = dict(device=dev, autoreset=_lib.AUTORESET_SAME_STEP, seed=1234)
    env = StreamShardedMultiGrid(spec, e, shards=shards, **common) if shards > 1 else BatchedMultiGrid(spec, e, **common)
    env.upload_layouts(*pool)
    out = env.reset()
    acts = torch.zeros((e, agents), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    if shards > 1:
        parts = [(env.shards[k], env.streams[k], lo, hi) for k, (lo, hi) in enumerate(env.bounds)]
    else:
        parts = [(env, torch.cuda.Stream(device=dev), 0, e)]

    def iteration(sh, lo, hi, t):
        policy(out.image[lo:hi], out.direction[lo:hi], t, acts[lo:hi])
        sh.step(acts[lo:hi])

    # every slice: a few eager iterations, then a HIP graph of `graph_iters` iterations replayed on its stream
    graphs = []
    for sh, st, lo, hi in parts:
        with torch.cuda.stream(st):
            for t in range(3):
                iteration(sh, lo, hi, t)
    torch.cuda.synchronize(dev)
    for sh, st, lo, hi in parts:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for t in range(graph_iters):
                iteration(sh, lo, hi, t)
        graphs.append(g)
    torch.cuda.synchronize(dev)

    def go(n):
        for _ in range(n // graph_iters):
            for g, (_, st, _, _) in zip(graphs, parts):
                with torch.cuda.stream(st):
                    g.replay()

    go(40 * graph_iters)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    go(iters)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / iters
    state = env.get_state()
    torch.cuda.synchronize(dev)
    digest = {k: int(v.to(torch.int64).sum()) for k, v in state.items()}
    stats = env.stats().cpu().tolist()
    env.close()
    return dt, digest, stats


if __name__ == "__main__":
    for c in [int(x) for x in sys.argv[1:]] or [5, 4, 3]:
        res = {}
        for s in (1, 2):
            res[s] = closed_loop(c, s, 1600)
            print(f"cfg{c} closed loop, {s} stream(s): {res[s][0] * 1e6:7.1f} us per iteration (policy + step of the whole batch), "
                  f"episodes {res[s][2][0]}", flush=True)
        same = res[1][1] == res[2][1] and res[1][2] == res[2][2]
        print(f"cfg{c}: same final state and statistics with one and two streams: {same}; "
              f"{res[1][0] / res[2][0]:.2f} x", flush=True)
        if not same:
            sys.exit(1)
