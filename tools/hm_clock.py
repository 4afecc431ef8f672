#!/usr/bin/env python3
"""Where a streamed launch of the multi-query pass spends its time (scan_half_multi_kernel<true, .>; a
-DMI355REC_PHASE_CLOCK build, --lib): scanners stamp entry (0), fragment + cutoffs in place (1), steps done (2), last
candidates resolved (3), lists stored (4) — by wave 0 of each workgroup; mergers and seed riders entry (0) and exit (5).
  python3 tools/hm_clock.py --lib gpurun_out/q8/libmi355rec_phase.so --queries 12"""
import argparse, ctypes, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=12)
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
nb = a.queries
rows = [(k * 7919) % a.rows for k in range(64)]
q = t[torch.tensor(rows[:nb], device="cuda")].cpu().numpy()
ex = np.array(rows[:nb], dtype=np.int64)
with CosineEngine(t) as eng:
    eng.set_batch_path(capi.BATCH_HALF)
    ring = [torch.zeros(nb * 100, dtype=torch.int64, device="cuda") for _ in range(4)]
    for k in range(12):
        eng.enqueue_batch_keys_streamed(q, ex, 100, ring[k % 4])
    torch.cuda.synchronize()   # the last launch with riders and mergers is the one the LAST enqueue made
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.size) == 0
    eng.enqueue_flush()
    torch.cuda.synchronize()
c = buf.reshape(1024, 8).astype(np.int64)
c = c[c[:, 0] > 0][:600]
t0 = c[:, 0].min()
us = lambda x: round(float(x) / 100.0, 2)
scan = c[(c[:, 4] > 0) & (c[:, 5] <= 0)]
other = c[c[:, 5] > 0]
out = {"queries": nb, "workgroups": int(len(c)), "scanners": int(len(scan)), "others": int(len(other))}
for i, nm in enumerate(["entry", "prologue_done", "steps_done", "resolved", "lists_stored"]):
    v = scan[:, i] - t0
    out[nm] = {"first": us(v.min()), "median": us(np.median(v)), "last": us(v.max())}
if len(other):
    d = other[:, 5] - other[:, 0]
    out["mergers_and_riders"] = {"entry_first": us((other[:, 0] - t0).min()), "exit_median": us(np.median(other[:, 5] - t0)),
                                 "exit_last": us((other[:, 5] - t0).max()), "duration_median": us(np.median(d)), "duration_max": us(d.max())}
out["launch_span_us"] = us(max(scan[:, 4].max(), other[:, 5].max() if len(other) else 0) - t0)
print(json.dumps(out))
