#!/usr/bin/env python3
"""Condense tools/profile_shards.sh's output: per workload the step kernel's launches from the kernel trace
(count, average duration, how much of the traced span had one / two launches in flight), the bench line of the
same run, and FETCH_SIZE / WRITE_SIZE per launch (KiB counters; FETCH_SIZE doubled as gfx950 needs for wide reads).

    python tools/prof_shards_summary.py gpurun_out/prof_shards > profiles/r02_stream_shards.json
"""
import csv
import glob
import json
import os
import re
import sys

src = sys.argv[1]
STEP = re.compile(r"mg_tile_kernel<\d, (true|false), 0,")


def bench_line(path):
    try:
        for line in open(path):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return {}


def counter(d):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if STEP.search(r["Kernel_Name"]):
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


out = []
for kt in sorted(glob.glob(os.path.join(src, "kt_*.json"))):
    name = os.path.basename(kt)[3:-5]
    spans = []
    for f in glob.glob(os.path.join(src, "kt_" + name, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if STEP.search(r["Kernel_Name"]):
                spans.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    spans.sort()
    spans = spans[len(spans) // 2:]  # the timed half: warm-up launches come first
    rec = {"workload": name, "launches_traced": len(spans)}
    if spans:
        rec["avg_launch_us"] = sum(e - s for s, e in spans) / len(spans) / 1e3
        events = sorted([(s, 1) for s, _ in spans] + [(e, -1) for _, e in spans])
        depth, last, busy = 0, events[0][0], {}
        for t, d in events:
            busy[depth] = busy.get(depth, 0) + t - last
            depth, last = depth + d, t
        total = events[-1][0] - events[0][0]
        rec["span_us"] = total / 1e3
        rec["fraction_of_span_with_n_launches_in_flight"] = {str(k): v / total for k, v in sorted(busy.items())}
    b = bench_line(kt)
    if b:
        rec["bench_us_per_whole_batch_step"] = b["roofline"]["us_per_step"]
        rec["bench_value"] = b["value"]
        rec["stream_shards"] = b["config"].get("stream_shards", 1)
        rec["algorithmic_bytes_per_whole_batch_step"] = b["roofline"]["algorithmic_bytes_per_env_step"] * b["config"]["envs_per_gpu"]
    f, w = counter(os.path.join(src, f"pmc_{name}_FETCH_SIZE")), counter(os.path.join(src, f"pmc_{name}_WRITE_SIZE"))
    if f is not None and w is not None:
        rec["FETCH_SIZE_KiB_per_launch"], rec["WRITE_SIZE_KiB_per_launch"] = f, w
        rec["hbm_bytes_per_launch"] = (2 * f + w) * 1024
    out.append(rec)
json.dump(out, sys.stdout, indent=1)
