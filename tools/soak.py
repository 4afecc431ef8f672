#!/usr/bin/env python3
"""Soak of the lock-free hand-offs inside a launch (last seed rider -> next launch's cutoff, last workgroup -> merge
and completion word, last rider of a batch stream -> the next batch's cutoffs): thousands of streamed queries,
batches and lone queries over the 8-bit replica, every key compared on the device with the same query over the
fp32 rows (which use none of those hand-offs).  A lost or stale cutoff shows as a missing key.
  python3 tools/soak.py --rows 10000000 --queries 6000"""
import argparse
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=6000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--catalogue", default="uniform", choices=["uniform", "clustered"], help="clustered: tools/catalogues.py")
    ap.add_argument("--spread", type=float, default=0.03)
    ap.add_argument("--clusters", type=int, default=3000)
    ap.add_argument("--contiguous", action="store_true")
    ap.add_argument("--ramp", action="store_true", help="clustered + contiguous: features[11] = cluster / (clusters - 1), the genre ramp of a CSV grouped by genre (DataManager.cpp:244-250,299)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from spotify_recommender_amd import CosineEngine, capi
    from spotify_recommender_amd.synth import synthetic_catalogue

    n = args.rows
    rng = np.random.default_rng(args.seed)
    if args.catalogue == "clustered":
        sys.path.insert(0, str(Path(__file__).resolve().parent))
        from catalogues import clustered_catalogue
        t = clustered_catalogue(n, args.spread, clusters=args.clusters, contiguous=args.contiguous, ramp=args.ramp)
    else:
        t = synthetic_catalogue(n, seed=12345)
    rows = rng.integers(0, n, size=args.queries)
    topns = rng.choice([1, 10, 100, 100, 100, 500, 1000], size=args.queries)
    out = {"rows": n, "queries": args.queries}
    if args.catalogue == "clustered":
        out.update({"catalogue": "clustered", "clusters": args.clusters, "spread": args.spread, "contiguous": args.contiguous})
    with CosineEngine(t) as eng:
        ref = []
        got = []
        for mode, sink in ((capi.REPLICA_OFF, ref), (capi.REPLICA_ON, got)):
            eng.set_replica(mode)
            for i in range(args.queries):
                k = torch.zeros(int(topns[i]), dtype=torch.int64, device="cuda")
                eng.enqueue_row_keys_streamed(int(rows[i]), int(topns[i]), k)
                sink.append(k)
                if i % 997 == 996:
                    eng.enqueue_flush()       # a new stream every ~1000 queries
            eng.enqueue_flush()
            torch.cuda.synchronize()
        bad = [i for i in range(args.queries) if not torch.equal(ref[i], got[i])]
        out["streamed_mismatches"] = len(bad)
        out["first_bad"] = bad[:5]
        # streams of batches: 12 queries per call, same rows, top-100
        nb = 12
        calls = min(400, args.queries // nb)
        q = t[torch.tensor(rows[:calls * nb], device="cuda")].cpu().numpy()
        ex = rows[:calls * nb].astype(np.int64)
        eng.set_replica(capi.REPLICA_ON)
        outs = []
        for c in range(calls):
            k = torch.zeros(nb * 100, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys_streamed(q[c * nb:(c + 1) * nb], ex[c * nb:(c + 1) * nb], 100, k)
            outs.append(k)
        eng.enqueue_flush()
        torch.cuda.synchronize()
        eng.set_replica(capi.REPLICA_OFF)
        bad_b = 0
        single = torch.zeros(100, dtype=torch.int64, device="cuda")
        for c in range(0, calls, 7):          # every 7th batch checked, query by query
            for b in range(nb):
                eng.enqueue_row_keys(int(rows[c * nb + b]), 100, single)
                torch.cuda.synchronize()
                bad_b += not torch.equal(single, outs[c][b * 100:(b + 1) * 100])
        out["batch_stream_mismatches"] = int(bad_b)
        # batches on their OWN (the sample launch's last workgroups hand the cutoffs to the pass, round 4) and streams of
        # 2- and 32-query batches, sizes mixed; every batch checked against single queries over the fp32 rows
        eng.set_replica(capi.REPLICA_ON)
        sizes = rng.choice([2, 5, 12, 20, 32], size=120)
        lone_outs, stream_outs, offs = [], [], []
        off = 0
        for nbq in sizes:
            sl = slice(off, off + int(nbq))
            offs.append(sl)
            qv = t[torch.tensor(rows[sl], device="cuda")].cpu().numpy()
            k1 = torch.zeros(int(nbq) * 100, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys(qv, rows[sl].astype(np.int64), 100, k1)
            lone_outs.append(k1)
            k2 = torch.zeros(int(nbq) * 100, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys_streamed(qv, rows[sl].astype(np.int64), 100, k2)
            stream_outs.append(k2)
            off += int(nbq)
        eng.enqueue_flush()
        torch.cuda.synchronize()
        eng.set_replica(capi.REPLICA_OFF)
        bad_m = 0
        for c in range(0, len(sizes), 3):
            for b, r in enumerate(rows[offs[c]]):
                eng.enqueue_row_keys(int(r), 100, single)
                torch.cuda.synchronize()
                bad_m += not torch.equal(single, lone_outs[c][b * 100:(b + 1) * 100])
                bad_m += not torch.equal(single, stream_outs[c][b * 100:(b + 1) * 100])
        out["mixed_batch_mismatches"] = int(bad_m)
        # lone synchronous queries (scan + merge + completion word in one launch at this size)
        eng.set_replica(capi.REPLICA_ON)
        bad_l = 0
        for i in range(0, min(args.queries, 1500)):
            idx, _ = eng.query_row_topn(int(rows[i]), int(topns[i]))
            want = (~ref[i].cpu().numpy().view(np.uint64) & np.uint64(0xffffffff)).astype(np.int64)
            bad_l += idx.tolist() != want.tolist()
        out["lone_mismatches"] = int(bad_l)
        out["lone_fused_queries"] = int(eng.stats().lone_fused_queries)
    print(json.dumps(out), flush=True)
    return 1 if (out["streamed_mismatches"] or out["batch_stream_mismatches"] or out["mixed_batch_mismatches"] or out["lone_mismatches"]) else 0


if __name__ == "__main__":
    sys.exit(main())
