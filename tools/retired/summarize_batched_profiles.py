#!/usr/bin/env python3
"""Turns the batched-path rocprofv3 outputs of tools/collect_profiles.sh into the files kept under profiles/:
  python tools/summarize_batched_profiles.py r02 gpurun_out/r2q
(<dir>/pmc_bq_a, <dir>/pmc_bq_b: two --pmc passes over tools/bqbench; <dir>/trace_bq: kernel trace of
tools/run_batched.py --rows 12500000 --batch 1024)."""
import collections
import csv
import glob
import json
import os
import sys

tag, root = sys.argv[1:3]


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "bq_pass_kernel<32" in k:
            a[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in d.items()} for k, d in a.items()}


kernels = agg(newest(f"{root}/pmc_bq_a/**/*counter_collection.csv"))
for k, d in agg(newest(f"{root}/pmc_bq_b/**/*counter_collection.csv")).items():
    kernels.setdefault(k, {}).update(d)
out = {
    "command": "rocprofv3 --pmc <7 + 5 SQ counters, 2 passes> --output-format csv -- tools/bqbench 12500000 4   (12.5 M rows x 1024 "
               "queries, 4 workgroups per CU, both passes over ALL 64-row tiles, pass 2 with thresholds nobody reaches; "
               "<32, *, 0, false> = rows from the fp32 matrix, <32, *, 0, true> = rows from the fp16 replica (the product's "
               "default); tools/collect_profiles.sh)",
    "units": "SQ_INSTS_* are wave-instructions, SQ_*_CYCLES / SQ_WAIT_* quad-cycles summed over waves, GRBM_GUI_ACTIVE "
             "cycles summed over the 8 XCDs; means per launch",
    "kernels": kernels, "reading": {},
}
for k, d in kernels.items():
    m = {c: v["mean"] for c, v in d.items()}
    if not m.get("SQ_INSTS_MFMA"):
        continue
    r = {"mfma_per_launch": m["SQ_INSTS_MFMA"], "valu_per_mfma": round(m["SQ_INSTS_VALU"] / m["SQ_INSTS_MFMA"], 2),
         "lds_per_mfma": round(m["SQ_INSTS_LDS"] / m["SQ_INSTS_MFMA"], 2),
         "salu_per_mfma": round(m["SQ_INSTS_SALU"] / m["SQ_INSTS_MFMA"], 2),
         "kernel_cycles": round(m["GRBM_GUI_ACTIVE"] / 8), "mfma_busy_cycles_per_simd": round(32 * m["SQ_INSTS_MFMA"] / 1024),
         "cycles_per_mfma_per_simd": round(m["GRBM_GUI_ACTIVE"] / 8 / (m["SQ_INSTS_MFMA"] / 1024), 1)}
    if "SQ_WAIT_INST_ANY" in m:
        r.update({"coexec_cycles_per_simd": round(m["SQ_VALU_MFMA_COEXEC_CYCLES"] / 1024),
                  "issue_stalled_share": round(m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                  "waiting_share": round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 3),
                  "lds_bank_conflict_cycles": m["SQ_LDS_BANK_CONFLICT"]})
    out["reading"][k] = r
out["conclusion"] = ("~9.5-10 (pass 1) / ~10.8-11.2 (pass 2, plus 2.3 scalar) VALU instructions per MFMA at 4 issue cycles each + "
                     "the MFMA's own 8 = 46-53 cycles of vector issue per 32 cycles of matrix pipe: VALU-issue bound; about half "
                     "of the wave time is issue-stalled; no LDS bank conflicts (one ds_read_b128 per two MFMAs). Reading the rows "
                     "from the fp16 replica removes ~0.5 VALU per MFMA (the per-row norm / scale / convert chain). The clock the "
                     "chip holds in this loop on random data is ~1.3-1.4 GHz (kernel_cycles / duration), and an all-zero catalogue "
                     "runs the same instruction stream in 0.72x the time (tools/bqbench): the passes are power-limited on top.")
json.dump(out, open(f"profiles/{tag}_pmc_batched_pass.json", "w"), indent=1)
keep = []
for r in csv.DictReader(open(newest(f"{root}/trace_bq/**/*kernel_stats.csv"))):
    if "mi355::" in r["Name"]:
        r["Name"] = r["Name"].split("(")[0].replace("void ", "")
        keep.append(r)
with open(f"profiles/{tag}_batched_12p5m_kernel_stats.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
    w.writeheader()
    w.writerows(keep)
for r in keep:
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
print(json.dumps(out["reading"], indent=1))
