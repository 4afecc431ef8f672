#!/usr/bin/env python3
"""Where a launch of the multi-query pass spends its time (scan_half_multi_kernel; a -DMI355REC_PHASE_CLOCK build, --lib):
scanners stamp entry (0), fragment + cutoffs in place (1), last candidates resolved (3), lists stored (4) — by wave 0 of each
workgroup — the moment the LAST wave of the workgroup has done its steps (2), the most any of its waves spent draining its
staging buffer / in exact steps (6) and the workgroup's number of exact steps (7); mergers and seed riders stamp entry (0)
and exit (5).
  bash tools/phase_build.sh gpurun_out/ph && python3 tools/hm_clock.py --lib gpurun_out/ph/libmi355rec_phase.so --queries 12
  ... --catalogue clustered --contiguous --clusters 3000 --spread 0.03 [--single]   (--single: one call on its own, no riders)"""
import argparse, ctypes, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=12)
ap.add_argument("--lib", required=True)
ap.add_argument("--single", action="store_true", help="time a call on its own (sample launch + pass + merge) instead of a streamed launch")
ap.add_argument("--catalogue", default="uniform", choices=["uniform", "clustered"])
ap.add_argument("--spread", type=float, default=0.03)
ap.add_argument("--contiguous", action="store_true")
ap.add_argument("--ramp", action="store_true")
ap.add_argument("--clusters", type=int, default=3000)
ap.add_argument("--spread-queries", action="store_true", help="query rows (k * 104729) mod N (spread over the shard) instead of (k * 7919) mod N (the first per cent of it)")
a = ap.parse_args()
import numpy as np
import torch
from spotify_recommender_amd import CosineEngine, capi
from spotify_recommender_amd.synth import clustered_catalogue, synthetic_catalogue
capi.LIB_PATH = Path(a.lib).resolve()
lib = capi.lib()
fn = lib.mi355rec_debug_phase_clock
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
if a.catalogue == "clustered":
    t = clustered_catalogue(a.rows, a.spread, clusters=a.clusters, contiguous=a.contiguous, ramp=a.ramp)
else:
    t = synthetic_catalogue(a.rows, seed=12345)
nb = a.queries
rows = [(k * (104729 if a.spread_queries else 7919)) % a.rows for k in range(64)]
q = t[torch.tensor(rows[:nb], device="cuda")].cpu().numpy()
ex = np.array(rows[:nb], dtype=np.int64)
with CosineEngine(t) as eng:
    eng.set_batch_path(capi.BATCH_HALF)
    ring = [torch.zeros(nb * 100, dtype=torch.int64, device="cuda") for _ in range(4)]
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    if a.single:
        for k in range(3):
            eng.enqueue_batch_keys(q, ex, 100, ring[k % 4])
        torch.cuda.synchronize()
        assert fn(buf.ctypes.data, buf.size) == 0
    else:
        for k in range(12):
            eng.enqueue_batch_keys_streamed(q, ex, 100, ring[k % 4])
        torch.cuda.synchronize()   # the last launch with riders and mergers is the one the LAST enqueue made
        assert fn(buf.ctypes.data, buf.size) == 0
        eng.enqueue_flush()
        torch.cuda.synchronize()
c = buf.reshape(1024, 8).astype(np.int64)
ids = np.nonzero(c[:, 0] > 0)[0][:600]
c = c[ids]
t0 = c[:, 0].min()
us = lambda x: round(float(x) / 100.0, 2)
is_scan = (c[:, 4] > 0) & (c[:, 5] <= 0)
scan, scan_ids = c[is_scan], ids[is_scan]
other = c[c[:, 5] > 0]
out = {"queries": nb, "workgroups": int(len(c)), "scanners": int(len(scan)), "others": int(len(other)), "catalogue": a.catalogue,
       "clusters": a.clusters if a.catalogue == "clustered" else None, "single_call": bool(a.single), "spread_queries": bool(a.spread_queries)}
for i, nm in ((0, "entry"), (1, "prologue_done"), (2, "steps_done"), (4, "lists_stored")):
    v = scan[:, i] - t0
    out[nm] = {"first": us(v.min()), "median": us(np.median(v)), "p95": us(np.percentile(v, 95)), "last": us(v.max())}
out["slowest_wave_drain_and_exact_us"] = {"median": us(np.median(scan[:, 6])), "p95": us(np.percentile(scan[:, 6], 95)), "max": us(scan[:, 6].max()),
                                   "workgroups_with_exact_steps": int((scan[:, 7] > 0).sum()), "exact_steps_max": int(scan[:, 7].max())}
late = np.argsort(-(scan[:, 4] - t0))[:8]
out["slowest_wave_mfma_part_of_overflowing_steps_us"] = {"p95": us(np.percentile(scan[:, 3], 95)), "max": us(scan[:, 3].max())}
out["slowest"] = [{"wg": int(scan_ids[i]), "steps_done": us(scan[i, 2] - t0), "hot_mfma_us": us(scan[i, 3]), "stored": us(scan[i, 4] - t0),
                   "drain_exact_us": us(scan[i, 6]), "exact_steps": int(scan[i, 7])} for i in late]
if len(other):
    d = other[:, 5] - other[:, 0]
    out["mergers_and_riders"] = {"entry_first": us((other[:, 0] - t0).min()), "exit_median": us(np.median(other[:, 5] - t0)),
                                 "exit_last": us((other[:, 5] - t0).max()), "duration_median": us(np.median(d)), "duration_max": us(d.max())}
out["launch_span_us"] = us(max(scan[:, 4].max(), other[:, 5].max() if len(other) else 0) - t0)
print(json.dumps(out))
