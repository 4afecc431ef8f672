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
ap.add_argument("--catalogue", default="uniform", choices=["uniform", "clustered"])
ap.add_argument("--spread", type=float, default=0.03)
ap.add_argument("--contiguous", action="store_true")
ap.add_argument("--ramp", action="store_true")
ap.add_argument("--clusters", type=int, default=3000)
ap.add_argument("--each", type=int, default=0, help="N streamed launches one by one (a device sync between them): per launch its span, the last scanner, the merger's, the riders' and the neighbourhood workgroup's exits")
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
ring = torch.zeros((4, a.topn), dtype=torch.int64, device="cuda")
def one_by_one(eng):
    rows = [(k * 7919 + 13) % a.rows for k in range(a.each + 24)]
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    for r in rows[:4]:
        eng.enqueue_row_keys_streamed(r, a.topn, ring[0])
    torch.cuda.synchronize()
    table = []
    for k, r in enumerate(rows[4:4 + a.each]):
        eng.enqueue_row_keys_streamed(r, a.topn, ring[k % 4])     # this launch: the scan of the query BEFORE, the merge of the one before that, the sample of r
        torch.cuda.synchronize()
        assert fn(buf.ctypes.data, buf.size) == 0
        c = buf.reshape(1024, 8).astype(np.int64)[:1023]
        t0 = c[c[:, 0] > 0, 0].min()
        sc = c[(c[:, 0] >= t0) & (c[:, 4] >= t0)]
        ot = c[(c[:, 0] >= t0) & (c[:, 5] >= t0)]
        ids = np.nonzero((c[:, 0] >= t0) & (c[:, 5] >= t0))[0]
        table.append({"scanned_row": rows[4 + k - 1], "span": round((max(sc[:, 4].max(), ot[:, 5].max()) - t0) / 100.0, 1),
                      "last_scanner": round((sc[:, 4].max() - t0) / 100.0, 1), "median_scanner": round(float(np.median(sc[:, 4] - t0)) / 100.0, 1),
                      "merger": round((ot[0, 5] - t0) / 100.0, 1), "nbhd": round((ot[-1, 5] - t0) / 100.0, 1),
                      "riders_last": round((ot[1:-1, 5].max() - t0) / 100.0, 1) if len(ot) > 2 else None})
    eng.enqueue_flush()
    torch.cuda.synchronize()
    spans = np.array([x["span"] for x in table])
    print(json.dumps({"each": a.each, "span_median": float(np.median(spans)), "span_mean": round(float(spans.mean()), 1), "span_max": float(spans.max()),
                      "slowest": sorted(table, key=lambda x: -x["span"])[:12], "fastest": sorted(table, key=lambda x: x["span"])[:3]}))


with CosineEngine(t) as eng:
    eng.set_replica(capi.REPLICA_OFF)
    if a.each:
        one_by_one(eng)
        sys.exit(0)
    for i in range(30):
        eng.enqueue_row_keys_streamed((i * 7919 + 13) % a.rows, a.topn, ring[i % 4])
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.size) == 0
    eng.enqueue_flush()
    torch.cuda.synchronize()
c_all = buf.reshape(1024, 8).astype(np.int64)[:1023]
ok = (c_all[:, 0] > 0) & (c_all[:, 4] > 0)
others = c_all[(c_all[:, 0] > 0) & (c_all[:, 5] > 0)]      # merger, seed riders, neighbourhood workgroup: entry (0), exit (5)
other_ids = np.nonzero((c_all[:, 0] > 0) & (c_all[:, 5] > 0))[0]
bids = np.nonzero(ok)[0]
c = c_all[ok]
t0 = c[:, 0].min()
us = lambda x: round(float(x) / 100.0, 2)
out = {"rows": a.rows, "topn": a.topn, "scanners": int(len(c))}
for i, nm in ((0, "entry"), (1, "query_ready"), (3, "tiles_done"), (4, "list_stored")):
    v = c[:, i] - t0
    out[nm] = {"first": us(v.min()), "median": us(np.median(v)), "p95": us(np.percentile(v, 95)), "last": us(v.max())}
out["launch_span_us"] = us(max(c[:, 4].max(), others[:, 5].max() if len(others) else 0) - t0)
out["catalogue"] = a.catalogue if a.catalogue == "uniform" else {"clusters": a.clusters, "spread": a.spread, "contiguous": a.contiguous, "ramp": a.ramp}
late = np.argsort(-(c[:, 4] - t0))[:8]
out["slowest_scanners"] = [{"wg": int(bids[i]), "query_ready": us(c[i, 1] - t0), "tiles_done": us(c[i, 3] - t0), "stored": us(c[i, 4] - t0)} for i in late]
out["merger_riders_nbhd"] = [{"wg": int(b), "entry": us(o[0] - t0), "exit": us(o[5] - t0)} for b, o in zip(other_ids, others)][:24]
# who finishes when: deciles of tiles_done, its median by dispatch order (thirds of blockIdx: with three workgroups per CU
# the dispatcher fills every CU once before it comes round again) and by XCD (blockIdx % 8)
td = c[:, 3] - t0
out["tiles_done_deciles"] = [us(np.percentile(td, p)) for p in range(0, 101, 10)]
third = (len(bids) + 2) // 3
out["tiles_done_median_by_third_of_blockIdx"] = [us(np.median(td[(bids >= k * third) & (bids < (k + 1) * third)])) for k in range(3)]
out["tiles_done_median_by_xcd"] = [us(np.median(td[bids % 8 == x])) for x in range(8)]
out["tiles_done_last_by_xcd"] = [us(td[bids % 8 == x].max()) for x in range(8)]
print(json.dumps(out))
