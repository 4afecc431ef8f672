#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of a round (gpurun_out/) into the small files kept
under profiles/:
  python tools/summarize_profiles.py r02 <kernel-trace dir> <FETCH dir> <WRITE dir> <bench json> [rows]
All three runs profile the same command, `python3 bench.py ...` (10 M rows unless given)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def newest(pattern):
    """gpurun merges every run's files into the same directory: take the latest."""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


tag, trace_dir, fetch_dir, write_dir, bench_json = sys.argv[1:6]
rows = int(sys.argv[6]) if len(sys.argv) > 6 else 10_000_000
stats = newest(f"{trace_dir}/**/*kernel_stats.csv")
keep = []
with open(stats) as f:
    rd = csv.DictReader(f)
    for r in rd:
        if "mi355::" in r["Name"]:
            r["Name"] = r["Name"].split("(")[0].replace("void ", "")
            keep.append(r)
with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
    w.writeheader()
    w.writerows(keep)
shutil.copy(bench_json, f"profiles/{tag}_bench_n1.json")
out = {
    "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 40 --warmup 5 "
               "--no-cpu-baseline --no-c5-shard --latency-queries 5   (one counter per pass, kernel-trace in its own run)",
    "unit_note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB. On gfx950 FETCH_SIZE counts 64 B per 128 B request "
                 "of a wide coalesced read, so read bytes = 2 * FETCH_SIZE * 1024 (MI355X_MICROARCH.md, HBM section); "
                 "the stream probe (a pure coalesced read of the same bytes) calibrates that factor; WRITE_SIZE is exact.",
    "kernels": {},
}
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    f = newest(f"{d}/**/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mi355::" in r["Kernel_Name"] and r["Counter_Name"] == name:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        e = out["kernels"].setdefault(k, {})
        e[name + "_KiB_mean"] = sum(v) / len(v)
        e[name + "_launches"] = len(v)
for k, d in out["kernels"].items():
    if "FETCH_SIZE_KiB_mean" in d:
        d["read_bytes_per_launch_corrected"] = 2 * d["FETCH_SIZE_KiB_mean"] * 1024
    if "WRITE_SIZE_KiB_mean" in d:
        d["write_bytes_per_launch"] = d["WRITE_SIZE_KiB_mean"] * 1024
    alg = None
    if "scan_kernel" in k or "stream_probe" in k or "scan_multi_kernel" in k:
        alg = rows * 48
    elif "scan_q8_kernel" in k:
        alg = (rows + 3) // 4 * 48
        d["note"] = ("scan over the 8-bit replica: 12 B per row + the fp32 rows it cannot rule out (~10 000 per query at top-100); "
                     "a streamed launch also carries the seed riders' sample for the next query (256 regions x 2048 rows x 12 B = 6.3 MB) "
                     "and the previous query's merge")
    elif "scan_half_multi_kernel" in k:
        alg = (rows + 1) // 2 * 48
        d["note"] = "multi-query pass over the fp16 replica (24 B per row whatever the number of queries, <= 32) + the fp32 rows of its candidates"
    elif "scan_half_kernel" in k:
        alg = (rows + 1) // 2 * 48
        d["note"] = "scan over the fp16 replica: 24 B per row + the fp32 rows it cannot rule out (a few thousand per query)"
    elif "bq_pass_kernel" in k:
        # bq_pass_kernel<NB, kCollect, kVariant, kFromReplica, kTileMax>: pass 2 (kCollect) reads every row once, pass 1
        # every 4th 64-row tile; 24 B per row from the fp16 replica, 48 B from the fp32 matrix.  With kTileMax pass 1 also
        # leaves 128 B per query block and visited tile (its per-lane maxima, fp16) and pass 2 reads them back
        args = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")]
        per_row = 24 if len(args) > 3 and args[3] == "true" else 48
        alg = rows * per_row if args[1] == "true" else rows * per_row // 4
        if len(args) > 4 and args[4] == "true":
            alg += (rows // 64 // 4) * int(args[0]) * 128
        d["note"] = "batched path: rows are read once per pass whatever the number of queries (<= 1024 per pass)"
    if alg is not None:
        d["algorithmic_bytes_per_launch"] = alg
        d["hbm_bytes_per_launch"] = d.get("read_bytes_per_launch_corrected", 0) + d.get("write_bytes_per_launch", 0)
        d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / alg
json.dump(out, open(f"profiles/{tag}_pmc_hbm_traffic.json", "w"), indent=1)
for k, d in out["kernels"].items():
    print(k[:70], {kk: round(vv, 3) if isinstance(vv, float) else vv for kk, vv in d.items() if "per_launch" in kk or "over" in kk})
