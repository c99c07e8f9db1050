#!/usr/bin/env python3
"""Throughput of the batched MultiGrid step on MI355X, next to a CPU port of the reference loop.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2..5] [--envs E] [--scaling weak|strong]

One "step" = one ``mg_step`` launch over every environment of the rank (actions already in
HBM, uniform random on {0..6}; agent order drawn on the device; episodes that end restart from
a pool of 1024 host-generated layouts inside the same launch; the environments start at step
counts scattered over [0, max_steps) so that episodes end at the steady-state rate from the
first timed step on).

``--gpus N`` with N > 1 starts N fresh processes, one per GPU, from a parent that never touches
a GPU (or joins the ranks that ``python -m torch.distributed.run --nproc-per-node N`` already
started): every rank holds its own shard of environments, there is no data-path collective, and
the only exchange is one RCCL all-reduce of the 64-byte episode-statistics vector (plus a
three-word check vector) after the timed region.  ``--scaling weak`` (default) keeps the per-GPU
batch fixed (config 5: 131072 environments per GPU, 2^20 on 8 GPUs = BASELINE configs[4]);
``--scaling strong`` splits the BASELINE total over the N GPUs.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# BASELINE.json configs: name, agents, environments per GPU (weak scaling), BASELINE total, extra kwargs
CONFIGS = {
    1: ("MultiGrid-Empty-8x8-v0", 2, 1, 1, {}),
    2: ("MultiGrid-RedBlueDoors-8x8-v0", 2, 4096, 4096, {}),
    3: ("MultiGrid-BlockedUnlockPickup-v0", 2, 65536, 65536, {}),
    4: ("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536, 262144, {}),   # 262144 over 4 GPUs
    5: ("MultiGrid-Empty-16x16-v0", 8, 131072, 1 << 20, {}),          # 2^20 over 8 GPUs
}


def algorithmic_bytes_per_env_step(spec, one_hot=False, steps_per_launch=1) -> float:
    """Bytes one environment step has to move with uint8 fields: actions + order in, image + direction +
    reward + terminated + truncated out (156 per agent), and - once per launch - agent rows and environment
    scalars in and out, the whole grid in, changed cells out.  With one-hot observations the 3-byte cells of
    the image leave as 24 bytes each."""
    a, w, h, v = spec["num_agents"], spec["width"], spec["height"], spec["view_size"]
    img = v * v * (24 if one_hot else 3)
    return (9 + img) * a + (2 * (8 * a + 8) + 3 * w * h + 3 * a) / steps_per_launch


def measured_traffic(name, num_envs, stream_shards=1):
    """HBM bytes per whole-batch step from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic*.json: one
    launch; profiles/*_stream_shards.json: the launches of all slices), when one matches this workload; the counters
    cannot be read from inside this process.  The latest round's file wins."""
    import glob
    best = None
    if stream_shards > 1:
        cid = next((c for c, v in CONFIGS.items() if v[0] == name), None)
        tag = f"cfg{cid}_{num_envs}_s{stream_shards}"
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_stream_shards.json"))):
            try:
                d = json.load(open(path))
            except (OSError, ValueError):
                continue
            for rec in (d if isinstance(d, list) else d.get("workloads", [])):
                if rec.get("workload") == tag and rec.get("hbm_bytes_per_launch"):
                    best = rec["hbm_bytes_per_launch"] * stream_shards  # one launch per slice and step
        return best
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic*.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if isinstance(d, dict) and isinstance(d.get("workloads"), list):
            d = d["workloads"]
        for rec in (d if isinstance(d, list) else [d]):
            w = rec.get("workload", "")
            if name in w and f"{num_envs} envs" in w and "overlap" not in w and "one-hot" not in w:
                best = rec.get("hbm_bytes_per_launch")
    return best


def fraction_fields(obj, path=""):
    """Every (path, value) of a bench line whose key says it is a fraction of something."""
    out = []
    if isinstance(obj, dict):
        for k, v in obj.items():
            here = f"{path}.{k}" if path else k
            if isinstance(v, (dict, list)):
                out += fraction_fields(v, here)
            elif "frac" in k and isinstance(v, (int, float)):
                out.append((here, float(v)))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            out += fraction_fields(v, f"{path}[{i}]")
    return out


def check_fractions(line):
    """A field named `*frac*` is a part of a whole: none may exceed 1 (a time comparison belongs under another
    name).  Raises with the offending paths."""
    bad = [(k, v) for k, v in fraction_fields(line) if not (0.0 <= v <= 1.0)]
    if bad:
        raise ValueError(f"bench line carries fractions outside [0, 1]: {bad}")
    return line


# Rehearsal of the multi-rank path on a box with ONE GPU (MGRID_BENCH_REHEARSE=1): every rank uses GPU 0 and the
# collectives run over gloo on host copies.  Everything else - sharding by env_id_offset, barriers, maximum over the
# ranks, the statistics all-reduce and its check - is the code a real `--gpus N` run executes.
REHEARSE = os.environ.get("MGRID_BENCH_REHEARSE") == "1"
SUSTAINED_CLOCK_STEPS = 1000  # untimed steps (W included) before the first timed one, see gpu_leg()


def local_device_index():
    return 0 if REHEARSE else int(os.environ.get("LOCAL_RANK", 0))


def all_reduce_(t, op):
    import torch
    if REHEARSE and t.is_cuda:
        c = t.cpu()
        torch.distributed.all_reduce(c, op=op)
        t.copy_(c)
    else:
        torch.distributed.all_reduce(t, op=op)
    return t


def gpu_leg(args, cfg_id, num_envs, rank, world, steps, warmup, graph=False, one_hot=False, repeats=1,
            steps_per_launch=1, stream_shards=1, collective=None, **override):
    """Times `repeats` runs of `steps` step launches each (every run bracketed by a barrier and a device
    synchronisation on both sides, maximum over ranks) and returns the median run.  With `stream_shards` S > 1 the
    rank's batch is held as S slices that step on streams of their own without waiting for each other
    (mg_step_shards): a step is then S launches, and the device time is taken from events on those streams."""
    import torch

    from multigrid_amd import _lib, envs, layouts
    from multigrid_amd.batched import BatchedMultiGrid

    name, agents, _, _, kw = CONFIGS[cfg_id]
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7, **dict(kw, **override))
    dev = torch.device("cuda", local_device_index())
    torch.cuda.set_device(dev)
    if collective is None:  # the statistics all-reduce runs whenever a process group holds more than this rank
        collective = world > 1
    S = max(1, stream_shards)
    if S > 1:
        from multigrid_amd.sharded import StreamShardedMultiGrid
        env = StreamShardedMultiGrid(spec, num_envs, shards=S, device=dev, autoreset=_lib.AUTORESET_SAME_STEP,
                                     env_id_offset=rank * num_envs, seed=1234, one_hot=one_hot, compact_image=not one_hot)
    else:
        env = BatchedMultiGrid(spec, num_envs, device=dev, autoreset=_lib.AUTORESET_SAME_STEP,
                               env_id_offset=rank * num_envs, seed=1234, one_hot=one_hot, compact_image=not one_hot)
    if args.pool == "device":
        # the pool drawn by the library on the GPU (mg_generate_layouts): layout l depends on (seed, l) only, so
        # every rank holds the same pool without 2-6 s of host generators per rank
        env.generate_layouts(gen, args.layouts, seed=0)
    else:
        grids, ags, metas, _ = layouts.make_pool(gen, args.layouts, seed0=0)
        env.upload_layouts(grids, ags, metas)
    env.reset()
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    # episodes in every phase of their life: ends (and restarts from the pool) arrive at the steady-state rate
    env.set_state(steps=torch.randint(0, spec["max_steps"], (num_envs,), generator=g, device=dev, dtype=torch.int32))
    T = steps_per_launch
    ring = 16 if T == 1 else 2 * T
    actions = torch.randint(0, 7, (ring, num_envs, agents), generator=g, device=dev, dtype=torch.uint8)
    outn = None
    if T > 1:
        from multigrid_amd.batched import StepOutput

        def rollout_buffers(e):
            a, v = agents, spec["view_size"]
            return StepOutput(image=torch.empty((T, e, a, v, v, 3), dtype=torch.uint8, device=dev),
                              direction=torch.empty((T, e, a), dtype=torch.uint8, device=dev),
                              reward=torch.empty((T, e, a), dtype=torch.float32, device=dev),
                              terminated=torch.empty((T, e, a), dtype=torch.uint8, device=dev),
                              truncated=torch.empty((T, e, a), dtype=torch.uint8, device=dev))
        if S > 1:  # every slice has its own [T, E_k, ...] inputs and outputs
            outn = [rollout_buffers(hi - lo) for lo, hi in env.bounds]
            acts_k = [[actions[h * T:(h + 1) * T, lo:hi].contiguous() for h in range(2)] for lo, hi in env.bounds]
        else:
            outn = rollout_buffers(num_envs)
    if S > 1:
        env.fork()  # the slices' streams start after everything built above (actions, their per-slice copies)

    shard_graphs = []

    def run(n, t0=0):
        if shard_graphs:  # every slice replays its own graph of `ring` launches on its own stream
            assert n % ring == 0
            for _ in range(n // ring):
                for k, gr in enumerate(shard_graphs):
                    with torch.cuda.stream(env.streams[k]):
                        gr.replay()
        elif S > 1 and T > 1:  # one fused launch of T steps per slice and stream
            assert n % T == 0
            for t in range(n // T):
                env.step_n([ak[t % 2] for ak in acts_k], outs=outn, fork=False, join=False)
        elif S > 1:  # the actions are complete and nothing here reads the outputs: the slices run free
            for t in range(n):
                env.step(actions[(t0 + t) % ring], fork=False, join=False)
        elif T == 1:
            for t in range(n):
                env.step(actions[(t0 + t) % ring])
        else:
            assert n % T == 0
            for t in range(n // T):
                env.step_n(actions[(t % 2) * T:(t % 2 + 1) * T], out=outn)

    run(warmup if T == 1 else max(T, warmup // T * T))
    # A GPU that has idled through the set-up takes some 25 ms of work to reach its sustained clocks (measured: the
    # same step is 5 % slower in the first milliseconds).  With a short warm-up the timed steps would be measured on
    # the way up, so further UNTIMED steps bring it there first; the line reports them (`extra_warmup_steps`).
    extra_warmup = max(0, SUSTAINED_CLOCK_STEPS - warmup) // max(1, T) * max(1, T)
    if extra_warmup:
        run(extra_warmup, warmup)
    graphed = None
    if graph and S > 1:
        env.fork()
        torch.cuda.synchronize(dev)
        for k, (sh, (lo, hi)) in enumerate(zip(env.shards, env.bounds)):
            gr = torch.cuda.CUDAGraph()
            mine = [actions[t, lo:hi] for t in range(ring)]
            with torch.cuda.graph(gr, stream=env.streams[k]):
                for t in range(ring):
                    sh.step(mine[t])
            shard_graphs.append(gr)
        torch.cuda.synchronize(dev)
    elif graph:
        # replay `ring` launches per graph launch; the step kernel keeps no host-side state
        graphed = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            run(ring)
            with torch.cuda.graph(graphed, stream=s):
                run(ring)
        torch.cuda.current_stream(dev).wait_stream(s)
    walls, gpus = [], []
    for rep in range(repeats):
        torch.cuda.synchronize(dev)
        if collective:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        on = env.streams if S > 1 else [torch.cuda.current_stream(dev)]  # the streams the kernels are launched on
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in on]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in on]
        wall0 = time.perf_counter()
        for ev, st in zip(ev0, on):
            ev.record(st)
        if graphed is not None:
            assert steps % ring == 0
            for _ in range(steps // ring):
                graphed.replay()
        else:
            run(steps, warmup + rep * steps)
        for ev, st in zip(ev1, on):
            ev.record(st)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - wall0
        gpu_s = max(a.elapsed_time(b) for a, b in zip(ev0, ev1)) / 1e3
        if collective:
            torch.distributed.barrier()
            t = torch.tensor([wall, gpu_s], dtype=torch.float64, device=dev)
            all_reduce_(t, torch.distributed.ReduceOp.MAX)
            wall, gpu_s = float(t[0]), float(t[1])
        walls.append(wall)
        gpus.append(gpu_s)
    mid = sorted(range(repeats), key=lambda i: walls[i])[repeats // 2]
    stats = env.stats()
    check = torch.tensor([1, steps, int(stats[5])], dtype=torch.int64, device=dev)
    if collective:  # the path's only collective: 64 + 24 bytes over RCCL, off the step path
        from multigrid_amd import distributed
        if REHEARSE:
            stats = distributed.allreduce_stats(stats.cpu())
        else:
            distributed.allreduce_stats(stats)
        all_reduce_(check, torch.distributed.ReduceOp.SUM)
    out = dict(spec=spec, name=name, agents=agents, num_envs=num_envs, wall=walls[mid], gpu_s=gpus[mid],
               walls=walls, stats=stats.cpu().numpy(), check=check.cpu().tolist(), max_steps=spec["max_steps"],
               extra_warmup=extra_warmup)
    env.close()
    return out


def latency_one_env(args):
    """BASELINE configs[0]: one MultiGrid-Empty-8x8 environment, 2 agents - the cost of a single step launch
    (enqueue to completion, one synchronising step at a time) and the back-to-back launch interval."""
    import torch

    from multigrid_amd import _lib, envs, layouts
    from multigrid_amd.batched import BatchedMultiGrid

    name, agents, _, _, kw = CONFIGS[1]
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7, **kw)
    dev = torch.device("cuda", local_device_index())
    env = BatchedMultiGrid(spec, 1, device=dev, autoreset=_lib.AUTORESET_SAME_STEP, seed=1)
    grids, ags, metas, _ = layouts.make_pool(gen, 4, seed0=0)
    env.upload_layouts(grids, ags, metas)
    env.reset()
    acts = torch.randint(0, 7, (64, 1, agents), device=dev, dtype=torch.uint8)
    for t in range(50):
        env.step(acts[t % 64])
    torch.cuda.synchronize(dev)
    lat = []
    for t in range(200):
        t0 = time.perf_counter()
        env.step(acts[t % 64])
        torch.cuda.synchronize(dev)
        lat.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for t in range(2000):
        env.step(acts[t % 64])
    torch.cuda.synchronize(dev)
    interval = (time.perf_counter() - t0) / 2000
    env.close()
    # the same environment behind the reference's own API shape: dicts in, dicts of host arrays out
    # (multigrid/base.py:364), one upload, two launches, one download and one wait per step
    import numpy as np

    from multigrid_amd import gym_api
    genv = gym_api.make(name, agents=agents, agent_view_size=7, layout_rng=np.random.default_rng(0))
    genv.reset(seed=0)
    picks = np.random.default_rng(1).integers(0, 7, size=(1200, agents)).tolist()
    for t in range(200):
        genv.step(dict(enumerate(picks[t])))
    t0 = time.perf_counter()
    for t in range(200, 1200):
        genv.step(dict(enumerate(picks[t])))
        if genv.is_done():
            genv.reset()
    dict_step = (time.perf_counter() - t0) / 1000
    genv.close()
    return dict(workload=f"{name}, {agents} agents, 1 env", sync_step_latency_us=statistics.median(lat) * 1e6,
                back_to_back_launch_us=interval * 1e6, agent_steps_per_s=agents / interval,
                dict_api_step_us=dict_step * 1e6, dict_api_agent_steps_per_s=agents / dict_step)


def other_entry_points(args):
    """The path's other two launches, `MultiGridEnv.gen_obs` (mg_gen_obs) and `MultiGridEnv.reset` of every environment
    (mg_reset with a start state picked per environment), on the per-GPU batches of BASELINE configs 3 - 5: launch
    time by HIP events, bytes per environment (state in, images out; for a reset the pool entry in and the state out
    as well) and the fraction of the 8 TB/s roofline."""
    import torch

    from multigrid_amd import envs, layouts
    from multigrid_amd.batched import BatchedMultiGrid

    dev = torch.device("cuda", local_device_index())
    out = []
    for cfg_id in (3, 4, 5):
        name, agents, num_envs, _, kw = CONFIGS[cfg_id]
        spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7, **kw)
        env = BatchedMultiGrid(spec, num_envs, device=dev)
        grids, ags, metas, _ = layouts.make_pool(gen, 256, seed0=0)
        env.upload_layouts(grids, ags, metas)
        env.reset()
        idx = torch.randint(0, 256, (num_envs,), device=dev, dtype=torch.int32)
        a, v = agents, spec["view_size"]
        hs = (16 + 8 * (a + a % 2) + 2 * (a + 2) + 15) // 16 * 16  # header block and grid of one environment, as held
        gs = (2 * spec["width"] * spec["height"] + 15) // 16 * 16   # on the device
        images = (3 * v * v + 1) * a
        for entry, fn, nbytes in (("mg_gen_obs", env.gen_obs, hs + gs + images),
                                  ("mg_reset, every environment", lambda: env.reset(layout_idx=idx), 2 * (hs + gs) + images)):
            for _ in range(50):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(dev)
            e0.record()
            for _ in range(300):
                fn()
            e1.record()
            torch.cuda.synchronize(dev)
            us = e0.elapsed_time(e1) / 300 * 1e3
            out.append(dict(workload=f"{name}, {agents} agents, {num_envs} envs", entry_point=entry, launch_us=us,
                            algorithmic_bytes_per_env=nbytes, roofline_frac=num_envs * nbytes / (us * 1e-6) / (HBM_PEAK_GBS * 1e9)))
        env.close()
    return out


def memory_rates(dev):
    """What this GPU's memory system gives plain streaming kernels (tools/membench/copy_rate.hip: 16 bytes per lane,
    grid-stride): a device-to-device copy (bytes read + bytes written per second) and a fill (bytes written per
    second), 1 GiB each, far beyond the caches.  The best of a few launch shapes is reported."""
    import ctypes

    import torch
    path = os.path.join(ROOT, "tools", "membench", "libmembench.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    mb = ctypes.CDLL(path)
    mb.mb_stream_ms.restype = ctypes.c_float
    mb.mb_stream_ms.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out = {}
    for name, src, moved in (("copy", a, 2 * n), ("fill", None, n)):
        best = 0.0
        for blocks in (2048, 8192, 32768):
            ms = mb.mb_stream_ms(None if src is None else ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                 n, 10, blocks, st)
            if ms <= 0:
                raise RuntimeError("the streaming yardstick kernel failed")
            best = max(best, moved * 10 / (ms / 1e3) / 1e9)
        out[name] = best
    return out


def pyref_rate(cfg_id, seconds):
    """The reference-style Python step loop (oracle/pyref.py, one environment, one core)."""
    import numpy as np

    from multigrid_amd import envs
    from oracle import pyref

    name, agents, _, _, kw = CONFIGS[cfg_id]
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7, **kw)
    rng = np.random.default_rng(0)
    grid, ags, meta, _ = gen(mixin_rng=np.random.default_rng(0), env_rng=np.random.default_rng(100))
    env = pyref.Env(pyref.EnvSpec(**spec), fast_obs=True)
    env.load_layout(grid, ags, meta)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            acts = rng.integers(0, 7, size=agents)
            order = rng.random(size=agents).argsort() if agents > 1 else (0,)
            env.step({i: int(acts[i]) for i in range(agents)}, order)
            if env.is_done():
                env.load_layout(grid, ags, meta)
            n += 1
    return n * agents / (time.perf_counter() - t0), n, name, agents


def cpu_leg(cfg_id, seconds=10.0):
    """The C restatement of the reference's step (oracle/mg_oracle.c) on one host core - the reported baseline - and
    beside it the same on every core, and the reference-style Python step loop (oracle/pyref.py, one environment)."""
    import numpy as np

    from multigrid_amd import envs
    from oracle.coracle import BatchOracle

    py_rate, n, name, agents = pyref_rate(cfg_id, 5.0)
    py1_rate, n1, name1, agents1 = pyref_rate(1, 4.0)
    _, _, _, _, kw = CONFIGS[cfg_id]
    spec, gen = envs.make_spec(name, num_agents=agents, agent_view_size=7, **kw)
    rng = np.random.default_rng(0)
    grid, ags, meta, _ = gen(mixin_rng=np.random.default_rng(0), env_rng=np.random.default_rng(100))

    e = 4096
    bo = BatchOracle(spec, e)
    bo.load(np.arange(e), grid[None], ags[None], meta[None])
    acts = rng.integers(0, 7, size=(8, e, agents)).astype(np.uint8)
    order = np.argsort(rng.random(size=(8, e, agents)), axis=-1).astype(np.uint8)
    m = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        bo.step(acts[m % 8], order[m % 8])
        done = (bo.terminated.all(axis=1) | bo.truncated.all(axis=1))
        if done.any():
            ids = np.nonzero(done)[0]
            bo.load(ids, grid[None], ags[None], meta[None])
        m += 1
    c_rate = m * e * agents / (time.perf_counter() - t0)

    # the same C port on every host core this process may use: one batch of environments per thread (the
    # library call releases the interpreter lock), all threads stepping for the same 4 seconds
    import threading
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", cores))))
    counts = [0] * cores
    stop_at = time.perf_counter() + 4.0

    def worker(k):
        b = BatchOracle(spec, e)
        b.load(np.arange(e), grid[None], ags[None], meta[None])
        n_ = 0
        while time.perf_counter() < stop_at:
            b.step(acts[n_ % 8], order[n_ % 8])
            n_ += 1
        counts[k] = n_

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    c_all_rate = sum(counts) * e * agents / (time.perf_counter() - t0)
    return dict(
        value=c_rate, unit="agent-steps/s", cores=1, kind="port",
        sample=f"{m} batched steps of {e} {name} envs ({agents} agents, uniform-random actions, finished episodes "
               f"restarted) through the C restatement of the reference's step (oracle/mg_oracle.c, gcc -O2, one "
               f"thread), {seconds:.0f} s",
        all_cores_value=c_all_rate, all_cores=cores,
        all_cores_sample=f"{sum(counts)} batched steps of {e} envs over {cores} threads, 4 s",
        python_port_value=py_rate, python_port_cores=1,
        python_port_sample=f"{n} steps of one {name} env through the Python restatement of the reference's step loop "
                           f"(oracle/pyref.py, vectorised gen_obs), 5 s",
        config1_python_port_value=py1_rate, config1_python_port_cores=1,
        config1_python_port_sample=f"{n1} steps of one {name1} env ({agents1} agents), same Python loop, 4 s "
                                   f"(BASELINE configs[0])")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=5, choices=[2, 3, 4, 5])
    ap.add_argument("--envs", type=int, default=0, help="environments per GPU (default: the BASELINE size)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: the per-GPU batch stays fixed; strong: the BASELINE total is split over the GPUs")
    ap.add_argument("--layouts", type=int, default=1024)
    ap.add_argument("--graph", action="store_true", help="replay the launches from a HIP graph")
    ap.add_argument("--fused", type=int, default=1, metavar="T",
                    help="T > 1: one fused launch per T steps (mg_step_n) instead of one launch per step")
    ap.add_argument("--stream-shards", type=int, default=1, metavar="S",
                    help="S > 1: every rank holds its batch as S slices stepping on streams of their own "
                         "(mg_step_shards), none waiting for the others")
    ap.add_argument("--repeats", type=int, default=0, help="timed runs of --steps launches; the median is reported "
                    "(default: 5 when --steps < 200, else 1)")
    ap.add_argument("--no-overlap", action="store_true", help="allow_agent_overlap=False (agents block each other)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other configs")
    ap.add_argument("--pool", choices=["auto", "host", "device"], default="auto",
                    help="layout pool: host generators (bit-identical to the reference's layouts) or mg_generate_layouts "
                         "on the GPU (same distribution, drawn in milliseconds); auto = host for one rank, device for several")
    ap.add_argument("--force-process-group", action="store_true",
                    help="with --gpus 1: join an RCCL process group of ONE rank and run the barriers and the statistics "
                         "all-reduce on GPU tensors through it (executes the multi-GPU code path on one GPU)")
    ap.add_argument("--selftest-gloo", action="store_true",
                    help="exercise the multi-process launcher and the collectives on the CPU (gloo); no GPU, no kernel")
    return ap.parse_args(argv)


def spawn_ranks(args, argv) -> int:
    """Parent of a multi-GPU run: starts one fresh process per rank and relays rank 0's line.  It makes no GPU
    call itself (it does not even import torch), and no process is replaced by another."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                       LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            # RCCL's intra-node transports open the peers' buffers through HIP IPC handles; the hosts this runs on
            # support the dmabuf form only, and without this setting hipIpcGetMemHandle fails with "invalid
            # argument" at the first collective.  Kept if the caller's environment already says something else.
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        rc = 0
        alive = list(procs)
        while alive:
            time.sleep(0.2)
            for p in list(alive):
                if p.poll() is not None:
                    alive.remove(p)
                    if p.returncode != 0:
                        rc = rc or p.returncode
            if rc and alive:  # a rank died: the others would wait for it forever
                for p in alive:
                    p.kill()
                for p in alive:
                    p.wait()
                alive = []
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
    return rc


def selftest_gloo(args, rank, world):
    """The launcher path, the process group and both collectives on the CPU: every rank reports `steps`
    pretend launches; the printed line has the same keys as a real one."""
    import torch
    import torch.distributed as dist

    from multigrid_amd import distributed
    if os.environ.get("MGRID_SELFTEST_FAIL_RANK") == str(rank):  # a rank that dies before the first collective
        sys.exit(7)
    stats = torch.tensor([0, 0, 0, 0, 0, args.steps + args.warmup, 0, 0], dtype=torch.int64)
    check = torch.tensor([1, args.steps, args.steps + args.warmup], dtype=torch.int64)
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
        distributed.allreduce_stats(stats)
        dist.all_reduce(check, op=dist.ReduceOp.SUM)
    wall = max(time.perf_counter() - t0, 1e-9)
    ok = check.tolist()[:2] == [world, world * args.steps]
    if rank == 0:
        print(json.dumps({"metric": "agent_steps_per_sec", "value": 0.0, "unit": "agent-steps/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / max(1, args.steps) * 1e3,
                          "higher_is_better": True, "scaling": args.scaling, "selftest": True, "backend": "gloo",
                          "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                          "collective_check": {"ranks_seen": check[0].item(), "steps_sum": check[1].item(),
                                               "launches_sum": check[2].item(), "ok": ok},
                          "stats_launches": int(stats[5])}))
    return 0 if ok else 3


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args, argv)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    backend = None
    if world > 1 or args.force_process_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # a group of one rank: the same calls a multi-GPU run makes, on the one GPU there is
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("LOCAL_RANK", "0")
        backend = "gloo" if (args.selftest_gloo or REHEARSE) else "nccl"
        if backend == "gloo":
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))
    if args.pool == "auto":
        args.pool = "device" if world > 1 else "host"
    if args.selftest_gloo:
        rc = selftest_gloo(args, rank, world)
        if world > 1:
            torch.distributed.destroy_process_group()
        return rc

    name, agents, per_gpu, total, _ = CONFIGS[args.config]
    num_envs = args.envs or (per_gpu if args.scaling == "weak" else max(1, total // world))
    repeats = args.repeats or (5 if args.steps < 200 else 1)
    T = max(1, args.fused)
    S = max(1, args.stream_shards)
    r = gpu_leg(args, args.config, num_envs, rank, world, args.steps, args.warmup, graph=args.graph, repeats=repeats,
                steps_per_launch=T, stream_shards=S, collective=backend is not None,
                **({"allow_agent_overlap": False} if args.no_overlap else {}))
    spec = r["spec"]
    total_agent_steps = world * num_envs * agents * args.steps
    value = total_agent_steps / r["wall"]
    launch_s = r["gpu_s"] / (args.steps / T)
    bytes_step = algorithmic_bytes_per_env_step(spec, steps_per_launch=T)
    achieved = num_envs * bytes_step * T / launch_s / 1e9
    traffic = measured_traffic(name, num_envs, S) if T == 1 else None
    ranks_seen, steps_sum, launches_sum = r["check"]
    wall_step_s = r["wall"] / args.steps

    line = {
        "metric": "agent_steps_per_sec", "value": value, "unit": "agent-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["wall"] / args.steps * 1e3, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "repeats": repeats, "ms_per_step_all_repeats": [w / args.steps * 1e3 for w in r["walls"]],
        "config": {"workload": f"{name}, {agents} agents, 7x7 view, {num_envs} envs per GPU "
                               f"({world * num_envs} total), uniform-random actions, on-device agent order, "
                               f"{'agents may not overlap, ' if args.no_overlap else ''}"
                               f"same-step auto-reset from {args.layouts} layouts, start step counts scattered "
                               f"over [0, {r['max_steps']})",
                   "baseline_config": args.config, "envs_per_gpu": num_envs, "agents": agents,
                   "grid": [spec["width"], spec["height"]], "steps_per_launch": T, "stream_shards": S,
                   "extra_warmup_steps": r["extra_warmup"],
                   "layout_pool": "mg_generate_layouts on the GPU" if args.pool == "device" else "host generators",
                   "launch": "hip_graph" if args.graph else "eager", "parallelism": f"env-shard x{world}"},
        "backend": backend,
        "rccl_ranks": (torch.distributed.get_world_size() if backend == "nccl" else None) if backend else 1,
        "collective_check": {"ranks_seen": ranks_seen, "steps_sum": steps_sum, "launches_sum": launches_sum,
                             "ok": ranks_seen == world and steps_sum == world * args.steps},
        # `frac` is priced at the kernel's launch duration, measured by HIP events on the launch stream(s) (what
        # rocprofv3 shows per kernel); `frac_at_ms_per_step` prices the same bytes at the wall clock `ms_per_step` is
        # taken from (host issue, barriers and synchronisation included), so that the two clocks of this line are
        # both on it, by name
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "clock": "HIP events on the launch stream",
                     "frac_at_ms_per_step": num_envs * bytes_step / wall_step_s / 1e9 / HBM_PEAK_GBS,
                     "traffic_unit": "bytes per whole-batch step, 2 x FETCH_SIZE + WRITE_SIZE from profiles/",
                     "frac_counter_bytes": (traffic / (launch_s / T) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "kernel": ("mg_tile_kernel" if T == 1 else "mg_tile_kernel (multi-step variant)") +
                               (f", {S} launches side by side on {S} streams per step: bytes of all of them over the "
                                f"time of a whole-batch step" if S > 1 else ""),
                     "algorithmic_bytes_per_env_step": bytes_step, "launch_us": launch_s * 1e6,
                     "us_per_step": launch_s / T * 1e6},
        "episode_stats": {"episodes": int(r["stats"][0]), "mean_length": float(r["stats"][1]) / max(1, int(r["stats"][0])),
                          "terminated": int(r["stats"][2]), "truncated": int(r["stats"][3]),
                          "reward_sum": float(r["stats"][4]) / 2 ** 24},
    }

    if REHEARSE:
        line["rehearsal"] = f"{world} ranks sharing GPU 0, collectives over gloo: exercises the multi-rank code, not a scaling number"
    if rank == 0 and world == 1 and not args.no_extra:
        rates = memory_rates(torch.device("cuda", local_device_index()))
        line["roofline"]["measured_copy_GBs"] = rates["copy"]
        line["roofline"]["measured_fill_GBs"] = rates["fill"]
        # (a time comparison, not a part of a whole: the step re-reads cached state, a 1 GiB copy cannot)
        line["roofline"]["achieved_over_measured_copy"] = achieved / rates["copy"]
        line["roofline"]["measured_with"] = "tools/membench/copy_rate.hip: 16 B per lane, 1 GiB, best of 3 launch shapes"
        extra = []

        def add(cid, e2, label="", n_steps=400, n_warm=50, one_hot=False, **kw):
            n2, a2 = CONFIGS[cid][0], CONFIGS[cid][1]
            Tx = kw.get("steps_per_launch", 1)
            x = gpu_leg(args, cid, e2, 0, 1, n_steps, n_warm, one_hot=one_hot, **kw)
            ls = x["gpu_s"] / (n_steps / Tx)
            b2 = algorithmic_bytes_per_env_step(x["spec"], one_hot=one_hot, steps_per_launch=Tx)
            Sx = kw.get("stream_shards", 1)
            plain = not (one_hot or kw.get("graph") or Tx > 1 or "allow_agent_overlap" in kw)
            tr = measured_traffic(n2, e2, Sx) if plain else None
            extra.append({"baseline_config": cid, "workload": f"{n2}, {a2} agents, {e2} envs" + (", " + label if label else ""),
                          "value": e2 * a2 * n_steps / x["wall"], "launch_us": ls * 1e6, "steps_per_launch": Tx,
                          "stream_shards": kw.get("stream_shards", 1),
                          "us_per_step": ls / Tx * 1e6, "algorithmic_bytes_per_env_step": b2,
                          # (a fused launch needs fewer bytes per step - the state travels once per launch - and is
                          # priced at its own bytes; compare times through `us_per_step`)
                          "roofline_frac": e2 * b2 * Tx / ls / 1e9 / HBM_PEAK_GBS,
                          "roofline_frac_counter_bytes": (tr / (ls / Tx) / 1e9 / HBM_PEAK_GBS) if tr else None,
                          "episodes": int(x["stats"][0])})

        for cid in (2, 3, 4, 5):
            if cid != args.config:
                add(cid, CONFIGS[cid][2])
        # the literal BASELINE batches of configs[3] and configs[4] on ONE GPU: working sets far beyond the
        # 256 MiB Infinity Cache
        add(4, CONFIGS[4][3], n_steps=200, n_warm=20)
        add(5, CONFIGS[5][3], n_steps=200, n_warm=20)
        # configs[4] again with agents blocking each other, and with one-hot observations
        # (OneHotObsWrapper, multigrid/wrappers.py:60-146) written by the step launch instead of the compact image
        add(5, CONFIGS[5][2], "allow_agent_overlap=False", 200, 30, allow_agent_overlap=False)
        add(5, CONFIGS[5][2], "one-hot observations (V,V,24)", 200, 30, one_hot=True)
        # launch-bound batches: the same steps replayed from a HIP graph, and fused into one launch per T steps
        for cid in (2, 3, 4):
            add(cid, CONFIGS[cid][2], "HIP graph of 16 launches", 400, 48, graph=True)
        for cid in (2, 3, 4, 5):
            for Tx in (8, 64):
                add(cid, CONFIGS[cid][2], f"one launch per {Tx} steps (mg_step_n)", 384 if cid != 5 else 128, 64,
                    steps_per_launch=Tx)
        # the same one-step launches with the batch held as two slices on two streams (mg_step_shards): the start
        # and the drain of one slice's launch overlap the other's work
        for cid, e2 in ((5, CONFIGS[5][2]), (4, CONFIGS[4][3]), (4, CONFIGS[4][2]), (3, CONFIGS[3][2])):
            add(cid, e2, "2 stream shards (mg_step_shards)", 400, 48, stream_shards=2)
        add(5, CONFIGS[5][2], "allow_agent_overlap=False, 2 stream shards (mg_step_shards)", 400, 48,
            stream_shards=2, allow_agent_overlap=False)
        for cid in (3, 4):
            add(cid, CONFIGS[cid][2], "2 stream shards, each replaying a HIP graph of 16 launches", 400, 48,
                stream_shards=2, graph=True)
        for cid in (3, 4, 5):
            add(cid, CONFIGS[cid][2], "2 stream shards, one launch per 8 steps each (mg_step_n)", 384 if cid != 5 else 128,
                64, stream_shards=2, steps_per_launch=8)
        line["other_configs"] = extra
        # the headline workload again, held as two slices on two streams: the same kernel and the same results,
        # whole-batch steps no longer separated by the drain of one launch and the start of the next
        if S == 1 and T == 1 and not args.graph and not args.no_overlap:
            twin = next((x for x in extra if x["baseline_config"] == args.config and x["stream_shards"] == 2 and
                         x["workload"].startswith(f"{name}, {agents} agents, {num_envs} envs, 2 stream")), None)
            if twin:
                line["stream_sharded"] = {k: twin[k] for k in ("workload", "value", "stream_shards", "us_per_step",
                                                               "roofline_frac")}
                line["stream_sharded"]["speedup_over_one_stream"] = twin["value"] / value
        line["config1_one_env"] = latency_one_env(args)
        line["other_entry_points"] = other_entry_points(args)
    if rank == 0 and world == 1 and not args.no_cpu:
        line["cpu_baseline"] = cpu_leg(args.config)
        line["cpu_baseline"]["host_cpus"] = os.cpu_count()
    if rank == 0:
        print(json.dumps(check_fractions(line)))
    if backend is not None:
        torch.distributed.destroy_process_group()
    return 0 if line["collective_check"]["ok"] else 3


if __name__ == "__main__":
    sys.exit(main())
