#!/usr/bin/env python3
"""1024-query batches (the two-pass matrix-core path of BASELINE configs[4]) dealt over L lanes of one handle, each on its own
stream: aggregate ms per batch against one handle.  The passes are VALU-bound with the chip full, but a batch also has ~60 us of small
latency-bound launches (prepare, select, finalize, the queue's) that another lane's passes can run beside.
  python3 tools/batched_lanes.py [--rows 12500000] [--lanes 2]"""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=12_500_000)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--topn", type=int, default=100)
ap.add_argument("--reps", type=int, default=24)
ap.add_argument("--lanes", type=int, default=2)
a = ap.parse_args()
import numpy as np
import torch
from spotify_recommender_amd import CosineEngine
from spotify_recommender_amd.synth import synthetic_catalogue
t = synthetic_catalogue(a.rows, seed=12345)
rows = torch.from_numpy((np.arange(a.batch, dtype=np.int64) * 104729) % a.rows).cuda()
q = t[rows].contiguous()
out = {"rows": a.rows, "batch": a.batch, "topn": a.topn}
eng = CosineEngine(t)
engs = [eng] + [eng.lane() for _ in range(a.lanes - 1)]
streams = [e.own_stream() for e in engs]
keys = [[torch.zeros(a.batch * a.topn, dtype=torch.int64, device="cuda") for _ in range(2)] for _ in engs]
torch.cuda.synchronize()
for nl in sorted({1, a.lanes}):
    def run(n):
        for k in range(n):
            l = k % nl
            engs[l].enqueue_batch_keys_dev(q, rows, a.topn, keys[l][(k // nl) % 2], stream=streams[l])
        torch.cuda.synchronize()
    run(2 * nl)
    t0 = time.perf_counter()
    run(a.reps)
    dt = (time.perf_counter() - t0) / a.reps
    out[f"lanes_{nl}"] = {"ms_per_batch": round(dt * 1e3, 4), "queries_per_s": round(a.batch / dt, 1)}
same = bool(torch.equal(keys[0][0], keys[-1][0]))
out["lanes_agree"] = same
print(json.dumps(out))
for e in reversed(engs):
    e.close()
