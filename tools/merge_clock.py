#!/usr/bin/env python3
"""Where the merge launch of a query alone spends its time (merge_notify_kernel; a -DMI355REC_PHASE_CLOCK build, --lib):
  python3 tools/merge_clock.py --lib gpurun_out/q8/libmi355rec_phase.so --rows 1000000 --topn 10"""
import argparse, ctypes, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--topn", type=int, default=10)
ap.add_argument("--lib", required=True)
a = ap.parse_args()
import numpy as np
import torch
from spotify_recommender_amd import CosineEngine, capi
from spotify_recommender_amd.synth import synthetic_catalogue
capi.LIB_PATH = Path(a.lib).resolve()
lib = capi.lib()
fn = lib.mi355rec_debug_phase_clock
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
t = synthetic_catalogue(a.rows, seed=12345)
rows = []
with CosineEngine(t) as eng:
    for i in range(30):
        eng.query_row_topn((i * 7919) % a.rows, a.topn)
        buf = np.zeros(1024 * 8, dtype=np.uint64)
        assert fn(buf.ctypes.data, buf.size) == 0
        rows.append(buf.reshape(1024, 8)[1023].astype(np.int64))
r = np.array(rows[5:])
d = (r[:, 1:5] - r[:, 0:4]) / 100.0
print(json.dumps({"rows": a.rows, "topn": a.topn, "merge_phases_us_median": {
    "entry->first chunk + threshold + append": round(float(np.median(d[:, 0])), 2),
    "deeper rounds": round(float(np.median(d[:, 1])), 2), "final cut": round(float(np.median(d[:, 2])), 2),
    "rank": round(float(np.median(d[:, 3])), 2)}, "entry_to_ranked_us": round(float(np.median(r[:, 4] - r[:, 0])) / 100.0, 2)}))
