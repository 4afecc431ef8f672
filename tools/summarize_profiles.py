#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of a round (gpurun_out/) into the small files kept
under profiles/:  python tools/summarize_profiles.py r01 <kernel-trace dir> <FETCH dir> <WRITE dir> <bench json>"""
import collections
import csv
import glob
import json
import shutil
import sys

tag, trace_dir, fetch_dir, write_dir, bench_json = sys.argv[1:6]
stats = glob.glob(f"{trace_dir}/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
shutil.copy(bench_json, f"profiles/{tag}_bench_n1.json")
out = {
    "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 40 --warmup 5 "
               "--no-cpu-baseline --latency-queries 5   (one counter per pass, kernel-trace in its own run)",
    "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB. On gfx950 FETCH_SIZE counts 64 B per 128 B request "
                 "of a wide coalesced read, so read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section); "
                 "the stream probe (a pure coalesced read of the same 480 MB) calibrates that factor; WRITE_SIZE is exact.",
    "kernels": {},
}
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mi355::" in r["Kernel_Name"] and r["Counter_Name"] == name:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        e = out["kernels"].setdefault(k, {})
        e[name + "_KiB_mean"] = sum(v) / len(v)
        e[name + "_launches"] = len(v)
rows = 10_000_000
for k, d in out["kernels"].items():
    if "FETCH_SIZE_KiB_mean" in d:
        d["read_bytes_per_launch_corrected"] = 2 * d["FETCH_SIZE_KiB_mean"] * 1024
    if "WRITE_SIZE_KiB_mean" in d:
        d["write_bytes_per_launch"] = d["WRITE_SIZE_KiB_mean"] * 1024
    if "scan_kernel" in k or "stream_probe" in k:
        d["algorithmic_bytes_per_launch"] = rows * 48
        d["hbm_bytes_per_launch"] = d.get("read_bytes_per_launch_corrected", 0) + d.get("write_bytes_per_launch", 0)
        d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / (rows * 48)
json.dump(out, open(f"profiles/{tag}_pmc_hbm_traffic.json", "w"), indent=1)
for k, d in out["kernels"].items():
    print(k[:70], {kk: round(vv, 3) if isinstance(vv, float) else vv for kk, vv in d.items() if "per_launch" in kk or "over" in kk})
