#!/usr/bin/env python3
"""Timing / profiling driver for the batched matrix-core path (BASELINE configs[4] as one
GPU sees it): `--rows` catalogue rows, `--batch` device-resident queries per call.
  python3 tools/run_batched.py --rows 12500000 --batch 1024 --reps 10
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 tools/run_batched.py ...
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
    ap.add_argument("--rows", type=int, default=12_500_000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--topn", type=int, default=100)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--path", type=int, default=0, help="0 auto, 1 multi-query passes, 2 batched MFMA, 5 batched MFMA without tile skipping")
    ap.add_argument("--lib", default=None, help="another build of the library (e.g. an MI355REC_EXPERIMENTS one under gpurun_out/)")
    ap.add_argument("--host-queries", action="store_true")
    ap.add_argument("--catalogue", default="uniform", choices=["uniform", "clustered"])
    ap.add_argument("--spread", type=float, default=0.03)
    ap.add_argument("--contiguous", action="store_true", help="clustered: a cluster's rows lie next to each other")
    ap.add_argument("--ramp", action="store_true", help="clustered + contiguous: features[11] = cluster / (clusters - 1), the genre ramp of a CSV grouped by genre (DataManager.cpp:244-250,299)")
    ap.add_argument("--clusters", type=int, default=3000)
    ap.add_argument("--check", type=int, default=0, help="compare this many of the batch's queries with the single-query fp32 scan, key for key")
    args = ap.parse_args()
    import numpy as np
    import torch
    if args.lib:
        from spotify_recommender_amd import capi
        capi.LIB_PATH = Path(args.lib).resolve()   # before the first capi.lib(): this process only
    from spotify_recommender_amd import CosineEngine
    from spotify_recommender_amd.synth import synthetic_catalogue

    if args.catalogue == "clustered":
        sys.path.insert(0, str(Path(__file__).resolve().parent))
        from catalogues import clustered_catalogue
        t = clustered_catalogue(args.rows, args.spread, clusters=args.clusters, contiguous=args.contiguous, ramp=args.ramp)
    else:
        t = synthetic_catalogue(args.rows, seed=12345)
    rows = np.array([(k * 7919) % args.rows for k in range(args.batch)], dtype=np.int64)
    qd = t[torch.from_numpy(rows).cuda()].contiguous()
    ed = torch.from_numpy(rows).cuda()
    qh = qd.cpu().numpy()
    keys = torch.zeros(args.batch * args.topn, dtype=torch.int64, device="cuda")
    with CosineEngine(t) as eng:
        eng.set_batch_path(args.path)

        def call():
            if args.host_queries or args.path == 1:
                eng.enqueue_batch_keys(qh, rows, args.topn, keys)
            else:
                eng.enqueue_batch_keys_dev(qd, ed, args.topn, keys)

        call()
        torch.cuda.synchronize()
        eng.set_timing(1)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        st = eng.stats()
        out = {"rows": args.rows, "batch": args.batch, "topn": args.topn, "path": args.path,
               "ms_per_batch": round(dt * 1e3, 4), "queries_per_s": round(args.batch / dt, 1),
               "pass_kernel_ms": round(float(st.last_pass_ms), 4),
               "effective_tflops_24flop_per_pair": round(24.0 * args.rows * args.batch / dt / 1e12, 2)}
        if args.path != 1:
            out.update(eng.batched_last_counters())
            out.update(eng.batched_pass2_pairs())
        if args.check:
            from spotify_recommender_amd import capi
            eng.set_replica(capi.REPLICA_OFF)
            ref = torch.zeros((args.check, args.topn), dtype=torch.int64, device="cuda")
            for i in range(args.check):
                eng.enqueue_row_keys(int(rows[i]), args.topn, ref[i])
            torch.cuda.synchronize()
            out["matches_single_query_fp32_scan"] = bool(torch.equal(ref, keys.view(args.batch, args.topn)[:args.check]))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
