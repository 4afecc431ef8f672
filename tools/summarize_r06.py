#!/usr/bin/env python3
"""Round-6 outputs of tools/collect_r06.sh (gpurun_out/r6p) -> the small files kept under profiles/:
  python tools/summarize_r06.py [gpurun_out/r6p]
r06_bench_n1.json / r06_bench_n1_20steps.json   the default bench line and the driver's 20-step form (PART A)
r06_kernel_stats.csv             rocprofv3 --kernel-trace --stats of the default command (TWO lanes: a streamed kernel's average
                                 there is a launch with the other lane's in flight)
r06_kernel_stats_one_handle.csv  the same command with --lanes 1: every streamed kernel alone on the chip (what DESIGN §5 quotes)
r06_pmc_hbm_traffic.json         FETCH_SIZE / WRITE_SIZE passes of the --lanes 1 command (separate --pmc runs)
r06_batched_kernel_stats.csv, r06_pmc_batched_pass.json   the batched two-pass path (PART C)
r06_clustered.json, r06_other_configs.jsonl, r06_virtual8.jsonl   (PART D)
Parts that have not been collected are skipped."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r6p"


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "")


def kernel_stats(d, run=None):
    f = newest(f"{root}/{d}/**/*kernel_stats.csv")
    rows = []
    if f:
        for r in csv.DictReader(open(f)):
            if "mi355::" in r["Name"]:
                r["Name"] = short(r["Name"])
                if run:
                    r["Run"] = run
                rows.append(r)
    return rows


def write_csv(path, rows):
    if not rows:
        return
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    print(path, len(rows), "kernels")


def counter_means(d):
    f = newest(f"{root}/{d}/**/*counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in out.items()}


# ---- PART A
if os.path.exists(f"{root}/bench.json") and newest(f"{root}/trace/**/*kernel_stats.csv") and newest(f"{root}/fetch/**/*counter_collection.csv"):
    subprocess.run([sys.executable, "tools/summarize_profiles.py", "r06", f"{root}/trace", f"{root}/fetch", f"{root}/write", f"{root}/bench.json"], check=True)
    p = "profiles/r06_pmc_hbm_traffic.json"
    d = json.load(open(p))
    d["command"] = ("rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --lanes 1 --steps 40 --warmup 5 --no-cpu-baseline --no-config0 "
                    "--no-clustered --no-single-lane --latency-queries 50 --no-c5-shard   (ONE handle: a launch's bytes are its own; one counter per pass)")
    json.dump(d, open(p, "w"), indent=1)
if os.path.exists(f"{root}/bench20.json"):
    shutil.copy(f"{root}/bench20.json", "profiles/r06_bench_n1_20steps.json")
write_csv("profiles/r06_kernel_stats_one_handle.csv", kernel_stats("trace1", "python3 bench.py --lanes 1 (one handle: streamed kernels alone on the chip)"))

# ---- PART C
rows = []
for r, label in (("12500000", "12.5 M rows x 1024 queries"), ("10000000", "10 M rows x 1024 queries")):
    rows += kernel_stats(f"bq_trace_{r}", label)
write_csv("profiles/r06_batched_kernel_stats.csv", rows)
a, b = counter_means("bq_sq_a"), counter_means("bq_sq_b")
if a and b:
    kernels = {}
    for k in a:
        if "bq_pass_kernel<32" not in k:
            continue
        c = dict(a[k])
        c.update(b.get(k, {}))
        m = c.get("SQ_INSTS_MFMA")
        if m:
            c["valu_per_mfma"] = c["SQ_INSTS_VALU"] / m
            c["salu_per_mfma"] = c["SQ_INSTS_SALU"] / m
            c["lds_per_mfma"] = c["SQ_INSTS_LDS"] / m
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs share the MFMAs
            c["gpu_cycles_per_mfma_per_simd"] = (c["GRBM_GUI_ACTIVE"] / 8.0) / (m / 1024.0)
            c["valu_issue_cycles_per_mfma"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / m
            c["mfma_pipe_cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / m
            c["waiting_share"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
            c["issue_stalled_share"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
        kernels[k] = c
    json.dump({"command": "rocprofv3 --pmc <one counter set per run> -- python3 tools/run_batched.py --rows 12500000 --batch 1024 --reps 6",
               "units": "SQ_INSTS_* wave-instructions, SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles summed over waves, "
                        "SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per v_mfma_f32_32x32x16_f16), GRBM_GUI_ACTIVE cycles summed over the 8 XCDs; means per launch",
               "reading": "gpu_cycles_per_mfma_per_simd ~= valu_issue_cycles_per_mfma + mfma_pipe_cycles_per_mfma: on a SIMD the vector "
                          "instructions of the reduction and the matrix pipe's 32 cycles per MFMA add up rather than overlap (DESIGN.md §5.5)",
               "kernels": kernels}, open("profiles/r06_pmc_batched_pass.json", "w"), indent=1)
    print("profiles/r06_pmc_batched_pass.json", {k[-28:]: round(v.get("gpu_cycles_per_mfma_per_simd", 0), 1) for k, v in kernels.items()})

# ---- PART D
if os.path.exists(f"{root}/clustered.json") and os.path.getsize(f"{root}/clustered.json"):
    d = json.load(open(f"{root}/clustered.json"))
    json.dump({"command": "python bench.py --catalogue clustered-contiguous --steps 100 --warmup 10 --no-cpu-baseline --no-config0 --no-c5-shard --latency-queries 100",
               "clustered": d.get("clustered")}, open("profiles/r06_clustered.json", "w"), indent=1)
    print("profiles/r06_clustered.json")
if os.path.exists(f"{root}/other_configs.jsonl"):
    shutil.copy(f"{root}/other_configs.jsonl", "profiles/r06_other_configs.jsonl")
v = [f"{root}/{n}" for n in ("virtual8.json", "virtual8_replicated.json", "virtual2_replicated.json") if os.path.exists(f"{root}/{n}") and os.path.getsize(f"{root}/{n}")]
if v:
    with open("profiles/r06_virtual8.jsonl", "w") as f:
        for path in v:
            f.write(open(path).read().strip() + "\n")
    print("profiles/r06_virtual8.jsonl", len(v))
