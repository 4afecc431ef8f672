#!/usr/bin/env python3
"""Where the ONE scan launch of a query alone (>= 4 M rows: scan + lone_tail) spends its time (a -DMI355REC_PHASE_CLOCK build,
--lib): python3 tools/lone_clock.py --lib gpurun_out/q8/libmi355rec_phase.so"""
import argparse, ctypes, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--topn", type=int, default=100)
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
res = []
with CosineEngine(t) as eng:
    for i in range(25):
        eng.query_row_topn((i * 7919) % a.rows, a.topn)
        buf = np.zeros(1024 * 8, dtype=np.uint64)
        assert fn(buf.ctypes.data, buf.size) == 0
        c = buf.reshape(1024, 8).astype(np.int64)[:1023]
        c = c[(c[:, 0] > 0) & (c[:, 4] > 0)]
        t0 = c[:, 0].min()
        res.append([np.median(c[:, 1] - t0), np.median(c[:, 2] - t0), np.median(c[:, 3] - t0), (c[:, 3] - t0).max(),
                    np.median(c[:, 4] - c[:, 3]), (c[:, 4] - t0).max(), len(c)])
r = np.array(res[5:], dtype=np.float64)
m = np.median(r, axis=0) / 100.0
print(json.dumps({"rows": a.rows, "topn": a.topn, "workgroups": int(r[0, 6]), "query_ready_median": round(m[0], 2),
                  "cutoff_ready_median": round(m[1], 2), "tiles_done_median": round(m[2], 2), "tiles_done_last": round(m[3], 2),
                  "store_wait_arrive_median": round(m[4], 2), "launch_end (the last workgroup: merged, fenced, word raised)": round(m[5], 2)}))
