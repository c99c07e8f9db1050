#!/usr/bin/env python3
"""Condense the output of tools/profile_round.sh into the files kept under profiles/.

    python tools/prof_summary.py gpurun_out/r2prof r02

Writes profiles/<tag>_kernel_stats.csv (the step kernel's row of every rocprofv3 --kernel-trace --stats run),
profiles/<tag>_pmc_traffic.json (FETCH_SIZE / WRITE_SIZE per launch, FETCH_SIZE doubled as gfx950 needs for wide
coalesced reads) and profiles/<tag>_pmc_sq.json (SQ counters per launch and per wavefront)."""
import collections
import csv
import glob
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEP = re.compile(r"mg_tile_kernel<\d, (true|false), 0,|mg_rollout_kernel")


def bench_line(path):
    try:
        for line in open(path):
            if line.startswith("{"):
                return json.loads(line)
    except OSError:
        pass
    return {}


def counters(d):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if STEP.search(r["Kernel_Name"]):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


stats_rows, traffic, sq = [], [], {}
for kt in sorted(glob.glob(os.path.join(src, "kt_*"))):
    if not os.path.isdir(kt):
        continue
    name = os.path.basename(kt)[3:]
    line = bench_line(kt + ".json")
    wl = line.get("config", {}).get("workload", name)
    for r in csv.DictReader(open(os.path.join(kt, "kt_kernel_stats.csv"))):
        if STEP.search(r["Name"]):
            stats_rows.append(dict(workload=name, kernel=re.sub(r"\(anonymous namespace\)::|^void |\(.*$", "", r["Name"].replace("void (anonymous namespace)::", "")),
                                   calls=r["Calls"], avg_ns=r["AverageNs"], min_ns=r["MinNs"], max_ns=r["MaxNs"],
                                   stddev=r["StdDev"], bench_launch_us=line.get("roofline", {}).get("launch_us")))
    fetch = counters(os.path.join(src, f"pmc_{name}_fetch")).get("FETCH_SIZE")
    write = counters(os.path.join(src, f"pmc_{name}_write")).get("WRITE_SIZE")
    if fetch and write:
        rd, wr = 2 * fetch[0] * 1024, write[0] * 1024
        avg_us = [float(r["avg_ns"]) / 1e3 for r in stats_rows if r["workload"] == name]
        envs = line.get("config", {}).get("envs_per_gpu")
        alg = line.get("roofline", {}).get("algorithmic_bytes_per_env_step")
        rec = dict(workload=wl.split(", uniform")[0].replace(" per GPU", ""), tag=name,
                   command=f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one run each) -- python3 bench.py {' '.join(line.get('argv', []))}",
                   launches_averaged=fetch[1], FETCH_SIZE_KiB=fetch[0], WRITE_SIZE_KiB=write[0],
                   read_bytes=rd, write_bytes=wr, hbm_bytes_per_launch=rd + wr,
                   algorithmic_bytes_per_launch=(envs * alg) if envs and alg else None,
                   kernel_trace_avg_us=avg_us[0] if avg_us else None)
        if avg_us:
            rec["counter_GBs"] = (rd + wr) / avg_us[0] / 1e3
            rec["frac_counter_bytes"] = rec["counter_GBs"] / 8000.0
            if rec["algorithmic_bytes_per_launch"]:
                rec["frac_algorithmic"] = rec["algorithmic_bytes_per_launch"] / avg_us[0] / 1e3 / 8000.0
        traffic.append(rec)
    c = {}
    for part in ("sq1", "sq2"):
        c.update({k: v[0] for k, v in counters(os.path.join(src, f"pmc_{name}_{part}")).items()})
    if c:
        waves = c.get("SQ_WAVES")
        rec = dict(per_launch=c)
        if waves:
            rec["per_wavefront"] = {k: v / waves for k, v in c.items() if k != "SQ_WAVES"}
        sq[name] = rec

with open(os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(stats_rows[0]))
    w.writeheader()
    w.writerows(stats_rows)
note = ("FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read on gfx950, so read bytes = 2 x FETCH_SIZE; "
        "WRITE_SIZE is exact for 16-B-per-lane stores; both count Infinity-Cache hits")
json.dump(dict(correction=note, workloads=traffic), open(os.path.join(root, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
json.dump(dict(units="SQ_WAVE_CYCLES / WAIT / ACTIVE_INST in quad-cycles summed over all wavefronts; INSTS in wavefront instructions",
               workloads=sq), open(os.path.join(root, "profiles", f"{tag}_pmc_sq.json"), "w"), indent=1)
for r in stats_rows:
    print(r["workload"], r["calls"], r["avg_ns"], r["bench_launch_us"])
for t in traffic:
    print(t["tag"], f"read {t['read_bytes']/1e6:.1f} MB write {t['write_bytes']/1e6:.1f} MB", "alg", t["algorithmic_bytes_per_launch"],
          "frac_counter", round(t.get("frac_counter_bytes", 0), 3), "frac_alg", round(t.get("frac_algorithmic", 0), 3))
for k, v in sq.items():
    print(k, {a: round(b, 1) for a, b in v.get("per_wavefront", {}).items()})
