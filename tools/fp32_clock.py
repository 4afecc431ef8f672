#!/usr/bin/env python3
"""Where a streamed launch of the fp32 scan spends its time (scan_kernel<.., kWithMerge>; a -DMI355REC_PHASE_CLOCK build,
--lib): python3 tools/fp32_clock.py --lib gpurun_out/q8/libmi355rec_phase.so"""
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
ring = torch.zeros((4, a.topn), dtype=torch.int64, device="cuda")
with CosineEngine(t) as eng:
    eng.set_replica(capi.REPLICA_OFF)
    for i in range(30):
        eng.enqueue_row_keys_streamed((i * 7919 + 13) % a.rows, a.topn, ring[i % 4])
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.size) == 0
    eng.enqueue_flush()
    torch.cuda.synchronize()
c_all = buf.reshape(1024, 8).astype(np.int64)[:1023]
ok = (c_all[:, 0] > 0) & (c_all[:, 4] > 0)
bids = np.nonzero(ok)[0]
c = c_all[ok]
t0 = c[:, 0].min()
us = lambda x: round(float(x) / 100.0, 2)
out = {"rows": a.rows, "topn": a.topn, "scanners": int(len(c))}
for i, nm in ((0, "entry"), (1, "query_ready"), (3, "tiles_done"), (4, "list_stored")):
    v = c[:, i] - t0
    out[nm] = {"first": us(v.min()), "median": us(np.median(v)), "p95": us(np.percentile(v, 95)), "last": us(v.max())}
out["launch_span_us"] = us(c[:, 4].max() - t0)
# who finishes when: deciles of tiles_done, its median by dispatch order (thirds of blockIdx: with three workgroups per CU
# the dispatcher fills every CU once before it comes round again) and by XCD (blockIdx % 8)
td = c[:, 3] - t0
out["tiles_done_deciles"] = [us(np.percentile(td, p)) for p in range(0, 101, 10)]
third = (len(bids) + 2) // 3
out["tiles_done_median_by_third_of_blockIdx"] = [us(np.median(td[(bids >= k * third) & (bids < (k + 1) * third)])) for k in range(3)]
out["tiles_done_median_by_xcd"] = [us(np.median(td[bids % 8 == x])) for x in range(8)]
out["tiles_done_last_by_xcd"] = [us(td[bids % 8 == x].max()) for x in range(8)]
print(json.dumps(out))
