#!/usr/bin/env python3
"""Throughput of the batched MultiGrid step on MI355X, next to a CPU port of the reference loop.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 1..5] [--envs E]

One "step" = one ``mg_step`` launch over every environment of the rank (actions already in
HBM, uniform random on {0..6}; agent order drawn on the device; episodes that end restart from
a pool of 1024 host-generated layouts inside the same launch).  For N > 1 launch with
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``: every rank holds
its own shard of environments (no data-path collective); the only collective is one RCCL
all-reduce of the 64-byte episode-statistics vector after the timed region.

Default workload: the per-GPU shard of BASELINE configs[4] - MultiGrid-Empty-16x16-v0,
8 agents, 131072 environments per GPU (2^20 over 8 GPUs) - with weak scaling.  ``--config``
selects the other BASELINE configurations at their stated sizes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak

# BASELINE.json configs: name, agents, environments per GPU, extra kwargs
CONFIGS = {
    1: ("MultiGrid-Empty-8x8-v0", 2, 1, {}),
    2: ("MultiGrid-RedBlueDoors-8x8-v0", 2, 4096, {}),
    3: ("MultiGrid-BlockedUnlockPickup-v0", 2, 65536, {}),
    4: ("MultiGrid-LockedHallway-4Rooms-v0", 4, 65536, {}),   # 262144 over 4 GPUs
    5: ("MultiGrid-Empty-16x16-v0", 8, 131072, {}),           # 2^20 over 8 GPUs
}


def algorithmic_bytes_per_env_step(spec, one_hot=False) -> int:
    """Bytes one environment step has to move with uint8 fields, one step per launch:
    actions + order in, image + direction + reward + terminated + truncated out (156 per agent),
    agent rows and environment scalars in and out, the whole grid in, changed cells out.
    With one-hot observations the 3-byte cells of the image leave as 24 bytes each."""
    a, w, h, v = spec["num_agents"], spec["width"], spec["height"], spec["view_size"]
    img = v * v * (24 if one_hot else 3)
    return (9 + img) * a + 2 * (8 * a + 8) + 3 * w * h + 3 * a


def measured_traffic(cfg_id, num_envs):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json), when one
    matches this workload; the counters cannot be read from inside this process."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        name = CONFIGS[cfg_id][0]
        if name in d.get("workload", "") and f"{num_envs} envs" in d.get("workload", ""):
            best = d.get("hbm_bytes_per_launch")
    return best


def gpu_leg(args, cfg_id, num_envs, rank, world, steps, warmup, graph=False, one_hot=False, **override):
    import numpy as np
    import torch

    from multigrid_amd import _lib, envs, layouts
    from multigrid_amd.batched import BatchedMultiGrid

    name, agents, _, kw = CONFIGS[cfg_id]
    spec, gen = envs.make_spec(name, num_agents=agents, **dict(kw, **override))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(dev)
    env = BatchedMultiGrid(spec, num_envs, device=dev, autoreset=_lib.AUTORESET_SAME_STEP,
                           env_id_offset=rank * num_envs, seed=1234, one_hot=one_hot, compact_image=not one_hot)
    grids, ags, metas, _ = layouts.make_pool(gen, args.layouts, seed0=0)
    env.upload_layouts(grids, ags, metas)
    env.reset()
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    ring = 16
    actions = torch.randint(0, 7, (ring, num_envs, agents), generator=g, device=dev, dtype=torch.uint8)

    def run(n, t0=0):
        for t in range(n):
            env.step(actions[(t0 + t) % ring])

    run(warmup)
    graphed = None
    if graph:
        # replay `ring` launches per graph launch; the step kernel keeps no host-side state
        graphed = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            run(ring)
            with torch.cuda.graph(graphed, stream=s):
                run(ring)
        torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    wall0 = time.perf_counter()
    ev0.record()
    if graphed is not None:
        assert steps % ring == 0
        for _ in range(steps // ring):
            graphed.replay()
    else:
        run(steps, warmup)
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - wall0
    gpu_s = ev0.elapsed_time(ev1) / 1e3
    if world > 1:
        torch.distributed.barrier()
        t = torch.tensor([wall, gpu_s], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        wall, gpu_s = float(t[0]), float(t[1])
    stats = env.stats()
    if world > 1:  # the path's only collective: 64 bytes over RCCL, off the step path
        torch.distributed.all_reduce(stats, op=torch.distributed.ReduceOp.SUM)
    stats = stats.cpu().numpy()
    out = dict(spec=spec, name=name, agents=agents, num_envs=num_envs, wall=wall, gpu_s=gpu_s,
               stats=stats, lds=None)
    env.close()
    return out


def memory_rates(dev):
    """What this GPU's memory system gives plain streaming kernels: a device-to-device copy (bytes read +
    bytes written per second) and a fill (bytes written per second), 1 GiB each, far beyond the caches."""
    import torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    out = {}
    for name, fn, moved in (("copy", lambda: b.copy_(a), 2 * n), ("fill", lambda: a.fill_(1), n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        out[name] = moved * 10 / (e0.elapsed_time(e1) / 1e3) / 1e9
    return out


def cpu_leg(cfg_id, seconds=12.0):
    """The reference-style Python step loop (oracle/pyref.py, one env, one core) and the C port."""
    import numpy as np

    from multigrid_amd import envs
    from oracle import pyref
    from oracle.coracle import BatchOracle

    name, agents, _, kw = CONFIGS[cfg_id]
    spec, gen = envs.make_spec(name, num_agents=agents, **kw)
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
    py_rate = n * agents / (time.perf_counter() - t0)

    e = 4096
    bo = BatchOracle(spec, e)
    bo.load(np.arange(e), grid[None], ags[None], meta[None])
    acts = rng.integers(0, 7, size=(8, e, agents)).astype(np.uint8)
    order = np.argsort(rng.random(size=(8, e, agents)), axis=-1).astype(np.uint8)
    m = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 4.0:
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
        value=py_rate, unit="agent-steps/s", cores=1, kind="port",
        sample=f"{n} steps of one {name} env ({agents} agents, uniform-random actions) through the Python "
               f"restatement of the reference step loop (oracle/pyref.py, vectorised gen_obs), {seconds:.0f} s",
        c_port_value=c_rate, c_port_cores=1,
        c_port_sample=f"{m} batched steps of {e} envs through oracle/mg_oracle.c (gcc -O2, one thread)",
        c_port_all_cores_value=c_all_rate, c_port_all_cores=cores,
        c_port_all_cores_sample=f"{sum(counts)} batched steps of {e} envs over {cores} threads, 4 s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=5, choices=[2, 3, 4, 5])
    ap.add_argument("--envs", type=int, default=0, help="environments per GPU (default: the BASELINE size)")
    ap.add_argument("--layouts", type=int, default=1024)
    ap.add_argument("--graph", action="store_true", help="replay the launches from a HIP graph")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the short runs of the other configs")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if args.gpus > 1:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks "
                             f"(WORLD_SIZE={world})")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ["LOCAL_RANK"])))

    name, agents, base_envs, _ = CONFIGS[args.config]
    num_envs = args.envs or base_envs
    r = gpu_leg(args, args.config, num_envs, rank, world, args.steps, args.warmup, graph=args.graph)
    spec = r["spec"]
    total_agent_steps = world * num_envs * agents * args.steps
    value = total_agent_steps / r["wall"]
    launch_s = r["gpu_s"] / args.steps
    bytes_step = algorithmic_bytes_per_env_step(spec)
    achieved = num_envs * bytes_step / launch_s / 1e9

    line = {
        "metric": "agent_steps_per_sec", "value": value, "unit": "agent-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": r["wall"] / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{name}, {agents} agents, 7x7 view, {num_envs} envs per GPU "
                               f"({world * num_envs} total), uniform-random actions, on-device agent order, "
                               f"same-step auto-reset from {args.layouts} layouts",
                   "baseline_config": args.config, "envs_per_gpu": num_envs, "agents": agents,
                   "grid": [spec["width"], spec["height"]], "steps_per_launch": 1,
                   "launch": "hip_graph" if args.graph else "eager", "parallelism": f"env-shard x{world}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.config, num_envs),
                     "traffic_unit": "bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE from profiles/",
                     "kernel": "mg_tile_kernel", "algorithmic_bytes_per_env_step": bytes_step,
                     "launch_us": launch_s * 1e6},
        "episode_stats": {"episodes": int(r["stats"][0]), "mean_length": float(r["stats"][1]) / max(1, int(r["stats"][0])),
                          "terminated": int(r["stats"][2]), "truncated": int(r["stats"][3]),
                          "reward_sum": float(r["stats"][4]) / 2 ** 24},
    }

    if rank == 0 and world == 1 and not args.no_extra:
        rates = memory_rates(torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))))
        line["roofline"]["measured_copy_GBs"] = rates["copy"]
        line["roofline"]["measured_fill_GBs"] = rates["fill"]
        line["roofline"]["frac_of_measured_copy"] = achieved / rates["copy"]
        extra = []
        for cid in (2, 3, 4, 5):
            if cid == args.config:
                continue
            n2, a2, e2, _ = CONFIGS[cid]
            x = gpu_leg(args, cid, e2, 0, 1, 400, 50)
            ls = x["gpu_s"] / 400
            b2 = algorithmic_bytes_per_env_step(x["spec"])
            extra.append({"baseline_config": cid, "workload": f"{n2}, {a2} agents, {e2} envs",
                          "value": e2 * a2 * 400 / x["wall"], "launch_us": ls * 1e6,
                          "roofline_frac": e2 * b2 / ls / 1e9 / HBM_PEAK_GBS})
        # configs[4] again with agents blocking each other, and with one-hot observations
        # (OneHotObsWrapper, multigrid/wrappers.py:60-146) written by the step launch instead of the compact image
        n5, a5, e5, _ = CONFIGS[5]
        for label, kw5, oh in (("allow_agent_overlap=False", dict(allow_agent_overlap=False), False),
                               ("one-hot observations (V,V,24)", {}, True)):
            x = gpu_leg(args, 5, e5, 0, 1, 200, 30, one_hot=oh, **kw5)
            ls = x["gpu_s"] / 200
            b5 = algorithmic_bytes_per_env_step(x["spec"], one_hot=oh)
            extra.append({"baseline_config": 5, "workload": f"{n5}, {a5} agents, {e5} envs, {label}",
                          "value": e5 * a5 * 200 / x["wall"], "launch_us": ls * 1e6,
                          "algorithmic_bytes_per_env_step": b5,
                          "roofline_frac": e5 * b5 / ls / 1e9 / HBM_PEAK_GBS})
        line["other_configs"] = extra
    if rank == 0 and world == 1 and not args.no_cpu:
        line["cpu_baseline"] = cpu_leg(args.config)
        line["cpu_baseline"]["host_cpus"] = os.cpu_count()
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
