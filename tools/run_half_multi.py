#!/usr/bin/env python3
"""Driver for the multi-query pass over the replicas (csrc/replica_multi.hip.h): single calls and streams of
batches of 1 ... 32 queries, with the kernel's event time and the rows it sent to the exact chain; --fp16 for
the fp16 front end.
  python3 tools/run_half_multi.py --rows 10000000 --topn 100
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 tools/run_half_multi.py --only-stream 12
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--topn", type=int, default=100)
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--no-exclude", action="store_true", help="the queries exclude no row (no neighbourhood bound is taken: handoff.hip.h)")
    ap.add_argument("--only-stream", type=int, default=0, help="only a stream of batches of this many queries (profiling)")
    ap.add_argument("--fp16", action="store_true", help="rows from the fp16 replica (24 B/row: the default from three queries per pass up) instead of the 8-bit one")
    ap.add_argument("--lib", default=None, help="another build of the library (e.g. an MI355REC_EXPERIMENTS one under gpurun_out/)")
    ap.add_argument("--sizes", default="1,2,4,8,12,16,24,32", help="batch sizes of the single calls ('' = none)")
    ap.add_argument("--streams", default="2,12,32", help="batch sizes of the streams ('' = none)")
    ap.add_argument("--catalogue", default="uniform", choices=["uniform", "clustered"])
    ap.add_argument("--spread", type=float, default=0.03)
    ap.add_argument("--contiguous", action="store_true", help="clustered: a cluster's rows lie next to each other")
    ap.add_argument("--ramp", action="store_true", help="clustered + contiguous: features[11] = cluster / (clusters - 1), the genre ramp of a CSV grouped by genre (DataManager.cpp:244-250,299)")
    ap.add_argument("--clusters", type=int, default=3000)
    args = ap.parse_args()
    if args.lib:
        from spotify_recommender_amd import capi as _capi
        _capi.LIB_PATH = Path(args.lib).resolve()   # before the first capi.lib(): this process only
    import numpy as np
    import torch
    from spotify_recommender_amd import CosineEngine, capi
    from spotify_recommender_amd.synth import synthetic_catalogue

    n, topn = args.rows, args.topn
    if args.catalogue == "clustered":
        sys.path.insert(0, str(Path(__file__).resolve().parent))
        from catalogues import clustered_catalogue
        t = clustered_catalogue(n, args.spread, clusters=args.clusters, contiguous=args.contiguous, ramp=args.ramp)
    else:
        t = synthetic_catalogue(n, seed=12345)
    rows = [(k * 7919) % n for k in range(64)]
    q = t[torch.tensor(rows, device="cuda")].cpu().numpy()
    out = {"rows": n, "topn": topn, "single_calls": [], "streams": []}
    with CosineEngine(t) as eng:
        st0 = eng.stats()
        out["margin_single"], out["margin_multi"] = round(float(st0.replica_margin_single), 6), round(float(st0.replica_margin_multi), 6)
        # (the 8-bit front end is an A/B route of MI355REC_EXPERIMENTS builds since round 5: --lib such a build to time it)
        front_q8 = not args.fp16 and capi.has_experiments()
        eng.set_batch_path(capi.BATCH_Q8 if front_q8 else capi.BATCH_HALF)
        out["front_end"] = "8-bit replica, 12 B/row, fp16 re-check of the candidates" if front_q8 else "fp16 replica, 24 B/row"
        sizes = tuple(int(x) for x in args.sizes.split(",") if x) if not args.only_stream else ()
        for nb in sizes:
            keys = torch.zeros(nb * topn, dtype=torch.int64, device="cuda")
            ex = np.full(nb, -1, dtype=np.int64) if args.no_exclude else np.array(rows[:nb], dtype=np.int64)
            eng.enqueue_batch_keys(q[:nb], ex, topn, keys)
            torch.cuda.synchronize()
            b = eng.replica_counters()
            eng.set_timing(1)
            t0 = time.perf_counter()
            for _ in range(args.calls):
                eng.enqueue_batch_keys(q[:nb], ex, topn, keys)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.calls
            st = eng.stats()
            eng.set_timing(0)
            a = eng.replica_counters()
            out["single_calls"].append({"queries": nb, "us_per_call": round(dt * 1e6, 1), "queries_per_s": round(nb / dt),
                                        "pass_kernel_us": round(st.last_scan_ms * 1e3, 1), "merge_kernel_us": round(st.last_merge_ms * 1e3, 1),
                                        "rows_to_exact_chain_per_query": round((a["rescored_rows"] - b["rescored_rows"]) / args.calls / nb)})
        if sizes:   # the largest single call once more, against the single-query scan over the fp32 rows
            nb = max(sizes)
            keys = torch.zeros(nb * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys(q[:nb], np.array(rows[:nb], dtype=np.int64), topn, keys)
            eng.set_replica(capi.REPLICA_OFF)
            ref = torch.zeros((nb, topn), dtype=torch.int64, device="cuda")
            for i in range(nb):
                eng.enqueue_row_keys(rows[i], topn, ref[i])
            torch.cuda.synchronize()
            out["matches_single_query_fp32_scan"] = bool(torch.equal(ref, keys.view(nb, topn)))
            eng.set_replica(capi.REPLICA_AUTO)
        for nb in (tuple(int(x) for x in args.streams.split(",") if x) if not args.only_stream else (args.only_stream,)):
            ring = [torch.zeros(nb * topn, dtype=torch.int64, device="cuda") for _ in range(4)]
            ex = np.full(nb, -1, dtype=np.int64) if args.no_exclude else np.array(rows[:nb], dtype=np.int64)
            for k in range(6):
                eng.enqueue_batch_keys_streamed(q[:nb], ex, topn, ring[k % 4])
            eng.enqueue_flush()
            torch.cuda.synchronize()
            eng.set_timing(1)
            t0 = time.perf_counter()
            for k in range(args.calls):
                eng.enqueue_batch_keys_streamed(q[:nb], ex, topn, ring[k % 4])
            eng.enqueue_flush()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.calls
            st = eng.stats()
            eng.set_timing(0)
            out["streams"].append({"queries": nb, "us_per_call": round(dt * 1e6, 1), "queries_per_s": round(nb / dt),
                                   "launch_kernel_us": round(st.last_scan_ms * 1e3, 1)})
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
