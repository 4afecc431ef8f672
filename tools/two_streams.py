#!/usr/bin/env python3
"""Does a second stream of single queries fill the ramps of the first?  K LANES of one handle (mi355rec_create_lane: the same rows and
replicas, own stream state; --separate: K independent handles, each with replicas of its own), each on its own HIP stream, streamed
queries dealt round-robin; aggregate queries/s against one handle alone.  python3 tools/two_streams.py [--handles 2] [--fp32]"""
import argparse, json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--topn", type=int, default=100)
ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--handles", type=int, default=2)
ap.add_argument("--separate", action="store_true")
ap.add_argument("--batch", type=int, default=0, help="streams of batches of this many queries (mi355rec_enqueue_batch_keys_streamed) instead of single queries")
ap.add_argument("--torch-streams", action="store_true", help="streams from torch's pool instead of the handles' own (they may share a hardware queue)")
ap.add_argument("--fp32", action="store_true", help="the fp32 rows (mi355rec_set_replica(OFF)) instead of the 8-bit replica")
a = ap.parse_args()
import numpy as np
import torch
from spotify_recommender_amd import CosineEngine
from spotify_recommender_amd.synth import synthetic_catalogue
t = synthetic_catalogue(a.rows, seed=12345)
rows = [(k * 7919 + 13) % a.rows for k in range(a.steps + 64)]
out = {"rows": a.rows, "topn": a.topn}
for nh in sorted({1, a.handles}):
    if a.separate:
        engs = [CosineEngine(t) for _ in range(nh)]
    else:
        engs = [CosineEngine(t)]
        engs += [engs[0].lane() for _ in range(nh - 1)]
    if a.fp32:
        from spotify_recommender_amd import capi
        for e in engs:
            e.set_replica(capi.REPLICA_OFF)
    streams = [torch.cuda.Stream() for _ in range(nh)] if a.torch_streams else [e.own_stream() for e in engs]
    rings = [[torch.zeros(a.topn, dtype=torch.int64, device="cuda") for _ in range(8)] for _ in range(nh)]
    if a.batch:
        rings = [[torch.zeros(a.batch * a.topn, dtype=torch.int64, device="cuda") for _ in range(8)] for _ in range(nh)]
        sel = np.array(rows[:a.batch], dtype=np.int64)
        qv = t[torch.from_numpy(sel).cuda()].cpu().numpy()

    def run(n0, n1):
        for k in range(n0, n1):
            h = k % nh
            if a.batch:
                engs[h].enqueue_batch_keys_streamed(qv, sel, a.topn, rings[h][(k // nh) % 8], stream=streams[h])
            else:
                engs[h].enqueue_row_keys_streamed(rows[k], a.topn, rings[h][(k // nh) % 8], stream=streams[h])
    run(0, 40)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(40, 40 + a.steps)
    for h in range(nh):
        engs[h].enqueue_flush(stream=streams[h])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[f"handles_{nh}"] = {"us_per_call": round(dt * 1e6 / a.steps, 2), "queries_per_s": round(a.steps * max(1, a.batch) / dt, 1)}
    for e in reversed(engs):
        e.close()
print(json.dumps(out))
