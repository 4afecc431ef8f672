#!/usr/bin/env python3
"""The round-4 rocprofv3 outputs of tools/collect_profiles.sh (gpurun_out/r4p) -> the small files kept under profiles/:
  python tools/summarize_r04.py [gpurun_out/r4p]
r04_pmc_scan_q8 / _scan_half / _scan_multi / _scan_half_multi / _batched_pass (.json, SQ counters + what they say),
r04_half_multi_kernel_stats.csv, r04_batched_kernel_stats.csv (per-kernel times of the multi-query stream and of the
batched path with and without tile skipping)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r4p"
SQ1 = "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
UNITS = ("SQ_INSTS_* wave-instructions, SQ_*_CYCLES / SQ_WAIT_* quad-cycles summed over waves, GRBM_GUI_ACTIVE cycles summed "
         "over the 8 XCDs, FETCH_SIZE / WRITE_SIZE KiB (FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read on "
         "gfx950: read bytes = 2 x FETCH_SIZE x 1024); means per launch")


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def short(name):
    return name.split("(")[0].replace("void ", "")


def counters(dirs, needle):
    kernels = collections.defaultdict(dict)
    for d in dirs:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(newest(f"{root}/{d}/**/*counter_collection.csv"))):
            k = short(r["Kernel_Name"])
            if needle in k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            for c, v in cs.items():
                kernels[k][c] = {"mean": sum(v) / len(v), "launches": len(v)}
    return kernels


def stats(d, needles):
    out = []
    for r in csv.DictReader(open(newest(f"{root}/{d}/**/*kernel_stats.csv"))):
        if any(n in r["Name"] for n in needles):
            r["Name"] = short(r["Name"])
            out.append(r)
    return out


def m(k, c):
    return k[c]["mean"] if c in k else None


def shares(k):
    """What the SQ1 set says about one kernel: instructions per wave-cycle shares."""
    wc = m(k, "SQ_WAVE_CYCLES")
    out = {}
    if wc:
        out["waiting_share_of_wave_cycles"] = round(m(k, "SQ_WAIT_ANY") / wc, 3)
        out["issue_stalled_share_of_wave_cycles"] = round(m(k, "SQ_WAIT_INST_ANY") / wc, 3)
    if m(k, "GRBM_GUI_ACTIVE"):
        out["kernel_cycles"] = round(m(k, "GRBM_GUI_ACTIVE") / 8)
    return out


def dump(name, obj):
    json.dump(obj, open(f"profiles/{name}", "w"), indent=1)
    print("wrote profiles/" + name)


# ---- single-query scans over the replicas ---------------------------------------------------------------------------
for tag, d, only, needle, rows_per_wave_tile in (("scan_q8", "pmc_q8", 2, "scan_q8_kernel", 256), ("scan_half", "pmc_half", 3, "scan_half_kernel", 128)):
    ks = counters([d], needle)
    reading = {}
    for k, c in ks.items():
        r = shares(c)
        r["valu_per_row"] = round(m(c, "SQ_INSTS_VALU") * 64 / 10_000_000, 2)   # wave-instructions x 64 lanes / rows
        r["salu_per_wave_per_1000_rows"] = round(m(c, "SQ_INSTS_SALU") / 10_000, 2)
        reading[k] = r
    dump(f"r04_pmc_{tag}.json", {
        "command": f"rocprofv3 --pmc {SQ1} --output-format csv -- python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 120 --check 8 --only {only}",
        "units": UNITS, "kernels": ks, "reading": reading})

# ---- the exact multi-query pass over the fp32 rows -------------------------------------------------------------------
ks = counters(["pmc_multi_a", "pmc_multi_b"], "scan_multi_kernel")
ks.update(counters(["pmc_multi_a", "pmc_multi_b"], "mi355::scan_kernel"))
reading = {}
for k, c in ks.items():
    r = shares(c)
    tiles = 10_000_000 / 64
    r["per_64_row_wave_tile"] = {"valu": round(m(c, "SQ_INSTS_VALU") / tiles, 1), "salu": round(m(c, "SQ_INSTS_SALU") / tiles, 1),
                                 "lds": round(m(c, "SQ_INSTS_LDS") / tiles, 1)}
    if m(c, "SQ_LDS_BANK_CONFLICT") is not None and m(c, "SQ_WAVE_CYCLES"):
        r["lds_bank_conflict_share_of_wave_cycles"] = round(m(c, "SQ_LDS_BANK_CONFLICT") / m(c, "SQ_WAVE_CYCLES"), 4)
    reading[k] = r
dump("r04_pmc_scan_multi.json", {
    "command": f"rocprofv3 --pmc <{SQ1} | SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR> --output-format csv -- "
               "python3 tools/run_multi_pass.py   (10 M rows, 72 queries = 6 passes of 12, x3; 5 single-query fp32 scans; one run per counter set)",
    "units": UNITS, "kernels": ks, "reading": reading,
    "conclusion": "the exact 12-query pass over the fp32 rows (route_multi_fp32: shards without a replica, or MI355REC_BATCH_MULTI) is "
                  "instruction- and barrier-bound, not LDS-bound, as in round 2; AUTO only takes it below 65 536 rows"})

# ---- the multi-query pass over the fp16 replica, as a stream of 12-query batches ---------------------------------------
ks = counters(["pmc_hm_a", "pmc_hm_b", "pmc_hm_f", "pmc_hm_w"], "scan_half_multi_kernel")
st = stats("trace_hm", ["half_multi", "merge_kernel"])
st32 = stats("trace_hm32", ["half_multi", "merge_kernel"])
with open("profiles/r04_half_multi_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=["stream_of"] + list(st[0].keys()))
    w.writeheader()
    for tag, rows in (("12-query batches", st), ("32-query batches", st32)):
        for r in rows:
            w.writerow(dict(r, stream_of=tag))
print("wrote profiles/r04_half_multi_kernel_stats.csv")
reading = {}
for k, c in ks.items():
    r = shares(c)
    rows = 10_000_000
    alg = rows * 24
    if m(c, "FETCH_SIZE") is not None:
        r["read_bytes_per_launch"] = round(2 * m(c, "FETCH_SIZE") * 1024)
        r["write_bytes_per_launch"] = round(m(c, "WRITE_SIZE") * 1024)
        r["algorithmic_bytes_per_launch"] = alg
        r["traffic_over_algorithmic"] = round((r["read_bytes_per_launch"] + r["write_bytes_per_launch"]) / alg, 3)
        r["traffic_note"] = ("a streamed launch also reads the NEXT batch's sample (5 % of the rows from 5 queries per batch up: 12.6 MB), "
                             "the fp32 rows of this batch's candidates and the previous batch's lists (12 queries x ~490 lists x 800 B = 4.7 MB), "
                             "and writes this batch's lists (the same 4.7 MB)")
    if m(c, "SQ_INSTS_MFMA"):
        r["mfma_per_launch"] = m(c, "SQ_INSTS_MFMA")
        r["valu_per_mfma"] = round(m(c, "SQ_INSTS_VALU") / m(c, "SQ_INSTS_MFMA"), 1)
        r["mfma_busy_share_of_kernel"] = round(m(c, "SQ_VALU_MFMA_BUSY_CYCLES") / (4 * 256) / (m(c, "GRBM_GUI_ACTIVE") / 8), 3) if m(c, "GRBM_GUI_ACTIVE") else None
    reading[k] = r
for r in st:
    if "scan_half_multi_kernel<true" in r["Name"]:
        reading.setdefault(r["Name"], {})["avg_us_in_kernel_trace_12_queries"] = round(float(r["AverageNs"]) / 1e3, 2)
for r in st32:
    if "scan_half_multi_kernel<true" in r["Name"]:
        reading.setdefault(r["Name"], {})["avg_us_in_kernel_trace_32_queries"] = round(float(r["AverageNs"]) / 1e3, 2)
dump("r04_pmc_scan_half_multi.json", {
    "command": "rocprofv3 --pmc <counters> --output-format csv -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60   (one run per "
               "counter set: SQ1, SQ_INSTS_MFMA ..., FETCH_SIZE, WRITE_SIZE; kernel times from --kernel-trace --stats runs of the same command "
               "and of --only-stream 32: profiles/r04_half_multi_kernel_stats.csv; tools/collect_profiles.sh PART=B)",
    "units": UNITS, "kernels": ks, "reading": reading})

# ---- the batched two-pass path ---------------------------------------------------------------------------------------
rows_out = []
for R in (10_000_000, 12_500_000):
    for pv, label in ((2, "tile skipping (default)"), (5, "no skipping (MI355REC_BATCH_MFMA_NOSKIP)")):
        log = open(f"{root}/bq_trace_{R}_p{pv}.log").read()
        line = next(l for l in log.splitlines() if l.startswith("{"))
        run = json.loads(line)
        for r in stats(f"bq_trace_{R}_p{pv}", ["bq_", "queued"]):
            rows_out.append(dict(r, rows=R, path=label, ms_per_batch=run["ms_per_batch"], pairs_done=run.get("pairs_done"),
                                 pairs_total=run.get("pairs_total")))
with open("profiles/r04_batched_kernel_stats.csv", "w", newline="") as f:
    keys = ["rows", "path", "ms_per_batch", "pairs_done", "pairs_total"] + [k for k in rows_out[0].keys() if k not in ("rows", "path", "ms_per_batch", "pairs_done", "pairs_total")]
    w = csv.DictWriter(f, fieldnames=keys)
    w.writeheader()
    w.writerows(rows_out)
print("wrote profiles/r04_batched_kernel_stats.csv")
out = {"command": "rocprofv3 --pmc <counters> --output-format csv -- python3 tools/run_batched.py --rows 12500000 --batch 1024 --reps 6 --path {2|5}   "
                  "(path 2 = the default: pass 2 skips the (tile, query block) pairs pass 1's maxima rule out; path 5 = MI355REC_BATCH_MFMA_NOSKIP, the same "
                  "kernels without the skip; one run per counter set; FETCH_SIZE / WRITE_SIZE for path 2 only; kernel times: profiles/r04_batched_kernel_stats.csv)",
       "units": UNITS, "paths": {}}
for pv in (2, 5):
    ks = counters([f"bq_pmc_a_p{pv}", f"bq_pmc_b_p{pv}"] + (["bq_pmc_f", "bq_pmc_w"] if pv == 2 else []), "bq_pass_kernel<32")
    reading = {}
    for k, c in ks.items():
        r = shares(c)
        mf = m(c, "SQ_INSTS_MFMA")
        if mf:
            cyc = m(c, "GRBM_GUI_ACTIVE") / 8
            r["mfma_per_launch"] = mf
            r["valu_per_mfma"] = round(m(c, "SQ_INSTS_VALU") / mf, 2)
            r["salu_per_mfma"] = round(m(c, "SQ_INSTS_SALU") / mf, 2)
            r["lds_per_mfma"] = round(m(c, "SQ_INSTS_LDS") / mf, 2)
            r["cycles_per_mfma_per_simd"] = round(cyc / (mf / 1024), 1)   # 1024 SIMDs
            r["mfma_busy_share_of_kernel"] = round(m(c, "SQ_VALU_MFMA_BUSY_CYCLES") / 1024 / cyc, 3)
            r["valu_active_share_of_kernel"] = round(m(c, "SQ_ACTIVE_INST_VALU") / 1024 / cyc, 3) if m(c, "SQ_ACTIVE_INST_VALU") else None
        if m(c, "FETCH_SIZE") is not None:
            r["read_bytes_per_launch"] = round(2 * m(c, "FETCH_SIZE") * 1024)
            r["write_bytes_per_launch"] = round(m(c, "WRITE_SIZE") * 1024)
        reading[k] = r
    out["paths"]["2 (skip)" if pv == 2 else "5 (no skip)"] = {"kernels": ks, "reading": reading}
dump("r04_pmc_batched_pass.json", out)
print(json.dumps({p: v["reading"] for p, v in out["paths"].items()}, indent=1))
