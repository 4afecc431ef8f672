#!/usr/bin/env python3
"""Round-5 rocprofv3 outputs of tools/collect_r05.sh (gpurun_out/r5p) -> the small files kept under profiles/:
  python tools/summarize_r05.py [gpurun_out/r5p]
r05_pmc_scan_half_multi.json   the multi-query pass's HBM traffic, split: a streamed launch of 12 queries against the pass
                               of a call on its own, that call's sample launch and its merge launch
r05_half_multi_kernel_stats.csv per-kernel times of the streams of 12- / 32-query batches and of single calls
r05_pmc_batched_pass.json, r05_batched_kernel_stats.csv   the batched two-pass path at 12.5 M rows (PART C)
r05_clustered.jsonl            every route over contiguous clusters (PART D, tools/clustered.sh)
r05_latency.jsonl, r05_other_configs.jsonl, r05_virtual8.jsonl   (PART D)
Parts that have not been collected are skipped."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5p"
UNITS = ("FETCH_SIZE / WRITE_SIZE: rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced "
         "read, so read bytes = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact; SQ_INSTS_* are "
         "wave-instructions, SQ_*_CYCLES / SQ_WAIT_* quad-cycles summed over waves; means per launch")


def newest(pattern):
    hits = glob.glob(pattern, recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def short(name):
    return name.split("(")[0].replace("void ", "")


def counter_means(d, counter=None):
    f = newest(f"{root}/{d}/**/*counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        if counter is None or r["Counter_Name"] == counter:
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in out.items()}


def kernel_stats(d):
    f = newest(f"{root}/{d}/**/*kernel_stats.csv")
    if not f:
        return []
    rows = []
    for r in csv.DictReader(open(f)):
        if "mi355::" in r["Name"]:
            r["Name"] = short(r["Name"])
            rows.append(r)
    return rows


def bytes_of(means, kernel_needle):
    """(read bytes, write bytes, launches) per launch of the first kernel whose name holds the needle."""
    rd = wr = n = None
    for k, cs in means.items():
        if kernel_needle in k:
            if "FETCH_SIZE" in cs:
                rd, n = 2 * cs["FETCH_SIZE"]["mean"] * 1024, cs["FETCH_SIZE"]["launches"]
            if "WRITE_SIZE" in cs:
                wr = cs["WRITE_SIZE"]["mean"] * 1024
    return rd, wr, n


# ---- the multi-query pass: what a streamed launch moves beside its own pass
if newest(f"{root}/hm_FETCH_SIZE_stream/**/*counter_collection.csv"):
    rows = 10_000_000
    alg = (rows + 1) // 2 * 48
    parts = {}
    for mode in ("stream", "call"):
        f = counter_means(f"hm_FETCH_SIZE_{mode}", "FETCH_SIZE")
        w = counter_means(f"hm_WRITE_SIZE_{mode}", "WRITE_SIZE")
        merged = collections.defaultdict(dict)
        for src in (f, w):
            for k, cs in src.items():
                merged[k].update(cs)
        parts[mode] = merged
    s_rd, s_wr, s_n = bytes_of(parts["stream"], "scan_half_multi_kernel<true")
    c_rd, c_wr, c_n = bytes_of(parts["call"], "scan_half_multi_kernel<false")
    sm_rd, sm_wr, sm_n = bytes_of(parts["call"], "seed_half_multi_kernel")
    mg_rd, mg_wr, mg_n = bytes_of(parts["call"], "merge_kernel")
    out = {
        "what": "HBM bytes per launch of the multi-query pass over the fp16 replica, 10 M rows x 12 queries x top-100 (csrc/replica_multi.hip.h): "
                "a STREAMED launch (scanners + the previous batch's riding merges + the next batch's seed riders and neighbourhood "
                "workgroups) against the pieces of a call on its own, where the pass, the sample and the merges are three launches",
        "commands": {"stream": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 tools/run_half_multi.py --fp16 --only-stream 12 --calls 60",
                     "call": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 tools/run_half_multi.py --fp16 --sizes 12 --streams= --calls 60"},
        "units": UNITS,
        "algorithmic_bytes_per_pass": alg,
        "streamed_launch": {"read_bytes": s_rd, "write_bytes": s_wr, "launches": s_n,
                            "traffic_over_algorithmic": (s_rd + s_wr) / alg if s_rd and s_wr else None},
        "pass_alone": {"read_bytes": c_rd, "write_bytes": c_wr, "launches": c_n,
                       "traffic_over_algorithmic": (c_rd + c_wr) / alg if c_rd and c_wr else None,
                       "note": "own pass only: 24 B per row + the fp32 rows of its candidates; writes = its per-workgroup key lists"},
        "sample_launch_of_a_call": {"read_bytes": sm_rd, "write_bytes": sm_wr, "launches": sm_n,
                                    "note": "10 % of the rows for 12 queries + one neighbourhood (2048 fp32 rows) per query; a stream's riders "
                                            "sample 5 %"},
        "merge_launch_of_a_call": {"read_bytes": mg_rd, "write_bytes": mg_wr, "launches": mg_n,
                                   "note": "one workgroup per query over the pass's key lists"},
    }
    if None not in (s_rd, s_wr, c_rd, c_wr, sm_rd, mg_rd):
        extra = (s_rd + s_wr) - (c_rd + c_wr)
        out["split"] = {"streamed_minus_pass_alone_bytes": extra,
                        "riders_sample_estimate_bytes": sm_rd / 2,
                        "riding_merges_estimate_bytes": mg_rd,
                        "unexplained_bytes": extra - sm_rd / 2 - mg_rd,
                        "reading": "what a streamed launch moves above a pass on its own is the next batch's sample (about half the "
                                   "sample launch's bytes: 5 % of the rows instead of 10 %) plus the previous batch's merges; the rest "
                                   "is within the run-to-run spread of the counters"}
    sq = counter_means("hm_sq_stream")
    for k, cs in sq.items():
        if "scan_half_multi_kernel<true" in k:
            out["sq_counters_streamed_launch"] = {c: round(v["mean"], 1) for c, v in cs.items()}
    json.dump(out, open("profiles/r05_pmc_scan_half_multi.json", "w"), indent=1)
    print("r05_pmc_scan_half_multi.json", json.dumps({k: out[k] for k in ("streamed_launch", "pass_alone", "split") if k in out})[:900])
    keep = []
    for d, tag in (("hm_trace_stream", "stream of 12-query batches"), ("hm_trace_stream32", "stream of 32-query batches"),
                   ("hm_trace_call", "single calls of 12 queries")):
        for r in kernel_stats(d):
            if any(n in r["Name"] for n in ("scan_half_multi", "seed_half_multi", "merge_kernel")):
                r = dict(r)
                r["Run"] = tag
                keep.append(r)
    if keep:
        with open("profiles/r05_half_multi_kernel_stats.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
            w.writeheader()
            w.writerows(keep)
    if os.path.exists(f"{root}/half_multi.json"):
        shutil.copy(f"{root}/half_multi.json", "profiles/r05_half_multi.json")

# ---- the batched path (PART C)
if newest(f"{root}/bq_trace/**/*kernel_stats.csv"):
    keep = []
    for d, tag in (("bq_trace", "12.5 M rows x 1024 queries"), ("bq_trace10", "10 M rows x 1024 queries")):
        for r in kernel_stats(d):
            if "bq_" in r["Name"] or "scan_multi_queued" in r["Name"]:
                r = dict(r)
                r["Run"] = tag
                keep.append(r)
    with open("profiles/r05_batched_kernel_stats.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
        w.writeheader()
        w.writerows(keep)
    merged = collections.defaultdict(dict)
    for d in ("bq_sq_a", "bq_sq_b", "bq_fetch", "bq_write"):
        for k, cs in counter_means(d).items():
            if "bq_pass_kernel" in k:
                merged[k].update({c: v["mean"] for c, v in cs.items()})
    out = {"command": "rocprofv3 --pmc <one counter set per run> -- python3 tools/run_batched.py --rows 12500000 --batch 1024 --reps 6",
           "units": UNITS, "kernels": {}}
    for k, cs in merged.items():
        e = dict(cs)
        mf = cs.get("SQ_INSTS_MFMA")
        if mf:
            e["valu_per_mfma"] = cs.get("SQ_INSTS_VALU", 0) / mf
            e["salu_per_mfma"] = cs.get("SQ_INSTS_SALU", 0) / mf
            e["lds_per_mfma"] = cs.get("SQ_INSTS_LDS", 0) / mf
            if cs.get("SQ_BUSY_CYCLES"):
                # SQ_BUSY_CYCLES is summed over the SQs (one per CU... per XCD shader engine): cycles per MFMA per SIMD from wave cycles
                pass
            if cs.get("SQ_WAVE_CYCLES"):
                # quad-cycles summed over waves; 4 waves per SIMD resident: cycles per MFMA per SIMD = 4 * wave_cycles / (waves per SIMD) / mfma ... reported raw
                e["wave_quad_cycles_per_mfma"] = cs["SQ_WAVE_CYCLES"] / mf
            if cs.get("SQ_WAIT_INST_ANY") and cs.get("SQ_WAVE_CYCLES"):
                e["waiting_share"] = cs.get("SQ_WAIT_ANY", 0) / cs["SQ_WAVE_CYCLES"]
                e["issue_stalled_share"] = cs["SQ_WAIT_INST_ANY"] / cs["SQ_WAVE_CYCLES"]
        out["kernels"][k] = e
    json.dump(out, open("profiles/r05_pmc_batched_pass.json", "w"), indent=1)
    print("r05_pmc_batched_pass.json", {k: {c: round(v, 2) for c, v in e.items() if "per_mfma" in c or "share" in c} for k, e in out["kernels"].items()})

# ---- PART D
if os.path.exists(f"{root}/cl/single.jsonl"):
    with open("profiles/r05_clustered.jsonl", "w") as f:
        for name, what in (("single", "single-query routes (tools/run_replica.py)"), ("multi", "multi-query pass (tools/run_half_multi.py --fp16)"),
                           ("batched", "1024-query batch (tools/run_batched.py)")):
            for line in open(f"{root}/cl/{name}.jsonl"):
                d = json.loads(line)
                d["what"] = what
                f.write(json.dumps(d) + "\n")
    print("r05_clustered.jsonl written")
if os.path.exists(f"{root}/latency_10m.json"):
    with open("profiles/r05_latency.jsonl", "w") as f:
        for name in ("latency_10m.json", "latency_1m.json"):
            f.write(open(f"{root}/{name}").read().strip() + "\n")
for src, dst in (("other_configs.jsonl", "r05_other_configs.jsonl"),):
    if os.path.exists(f"{root}/{src}"):
        shutil.copy(f"{root}/{src}", f"profiles/{dst}")
if os.path.exists(f"{root}/virtual8.json"):
    with open("profiles/r05_virtual8.jsonl", "w") as f:
        for name in ("virtual8.json", "virtual8_replicated.json"):
            if os.path.exists(f"{root}/{name}"):
                f.write(open(f"{root}/{name}").read().strip() + "\n")
