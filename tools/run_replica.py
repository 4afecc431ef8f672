#!/usr/bin/env python3
"""A/B driver for the single-query scans over the replicas (8-bit: csrc/replica_q8.hip.h, fp16:
csrc/replica.hip.h) against the fp32 scan: same queries through all three, keys compared bit for bit,
streamed step time, rows re-scored per query and lone-query latency of each.
  python3 tools/run_replica.py --rows 10000000 --topn 100 --steps 300
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/x -- python3 tools/run_replica.py ...
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from catalogues import clustered_catalogue   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--topn", type=int, default=100)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--check", type=int, default=64, help="queries compared between the two paths")
    ap.add_argument("--only", type=int, default=-1, help="time only this mode (1 fp32 rows, 2 8-bit replica, 3 fp16 replica)")
    ap.add_argument("--lib", default=None, help="another build of the library (e.g. an MI355REC_EXPERIMENTS one under gpurun_out/)")
    ap.add_argument("--catalogue", default="uniform", choices=["uniform", "clustered"],
                    help="clustered: min-max normalised audio-feature look-alike (discrete key / mode / genre columns, a few thousand "
                         "tight clusters, exact duplicates) instead of uniform noise: what the replicas cannot rule out grows")
    ap.add_argument("--spread", type=float, default=0.03, help="clustered: standard deviation of a cluster")
    ap.add_argument("--contiguous", action="store_true", help="clustered: a cluster's rows lie next to each other")
    ap.add_argument("--ramp", action="store_true", help="clustered + contiguous: features[11] = cluster / (clusters - 1), the genre ramp of a CSV grouped by genre (DataManager.cpp:244-250,299)")
    ap.add_argument("--clusters", type=int, default=3000, help="clustered: number of clusters")
    ap.add_argument("--by-value", action="store_true",
                    help="the timed legs pass the same catalogue rows BY VALUE with nothing excluded (mi355rec_enqueue_query_keys_streamed / "
                         "mi355rec_query_topn(q, -1)): the neighbourhood bound then comes from the query's anchor (csrc/handoff.hip.h)")
    args = ap.parse_args()
    if args.lib:
        from spotify_recommender_amd import capi as _capi
        _capi.LIB_PATH = Path(args.lib).resolve()   # before the first capi.lib(): this process only
    import numpy as np
    import torch
    from spotify_recommender_amd import CosineEngine, capi
    from spotify_recommender_amd.synth import synthetic_catalogue

    if args.catalogue == "clustered":
        t = clustered_catalogue(args.rows, args.spread, clusters=args.clusters, contiguous=args.contiguous, ramp=args.ramp)
    else:
        t = synthetic_catalogue(args.rows, seed=12345)
    rows = [(k * 7919 + 13) % args.rows for k in range(max(args.check, args.steps + 20, 200))]   # 200: the latency loop below
    out = {"rows": args.rows, "topn": args.topn, "by_value": bool(args.by_value)}
    qvecs = t[torch.tensor(rows, device=t.device)].cpu().numpy() if args.by_value else None
    with CosineEngine(t) as eng:
        st = eng.stats()
        out["replica_build_ms"] = round(float(st.replica_build_ms), 3)
        out["replica_grid_blocks"] = int(st.replica_grid_blocks)
        keys = {m: torch.zeros((args.check, args.topn), dtype=torch.int64, device="cuda") for m in (1, 2, 3)}
        # (single queries over the fp16 replica: an A/B route of MI355REC_EXPERIMENTS builds since round 5 — use --lib)
        fp16 = capi.has_experiments()
        if args.only < 0:
            for mode in (capi.REPLICA_OFF, capi.REPLICA_ON) + ((capi.REPLICA_FP16,) if fp16 else ()):
                eng.set_replica(mode)
                for i in range(args.check):
                    eng.enqueue_row_keys(rows[i], args.topn, keys[mode][i])
                torch.cuda.synchronize()
            for mode, name in ((2, "q8"),) + (((3, "fp16"),) if fp16 else ()):
                same = bool(torch.equal(keys[1], keys[mode]))
                out[f"keys_identical_{name}"] = same
                if not same:
                    bad = (keys[1] != keys[mode]).any(dim=1).nonzero().flatten().tolist()
                    out[f"first_bad_queries_{name}"] = bad[:8]
                # streamed too
                sk = torch.zeros((args.check, args.topn), dtype=torch.int64, device="cuda")
                eng.set_replica(mode)
                for i in range(args.check):
                    eng.enqueue_row_keys_streamed(rows[i], args.topn, sk[i])
                eng.enqueue_flush()
                torch.cuda.synchronize()
                out[f"streamed_identical_{name}"] = bool(torch.equal(sk, keys[1]))
        ring = torch.zeros((64, args.topn), dtype=torch.int64, device="cuda")
        for mode, name in ((capi.REPLICA_OFF, "fp32_rows"), (capi.REPLICA_FP16, "replica_fp16"), (capi.REPLICA_ON, "replica_q8")):
            if (args.only >= 0 and args.only != mode) or (mode == capi.REPLICA_FP16 and not fp16):
                continue
            eng.set_replica(mode)
            def step(i, k):
                if args.by_value:
                    eng.enqueue_query_keys_streamed(qvecs[i], -1, args.topn, ring[k % 64])
                else:
                    eng.enqueue_row_keys_streamed(rows[i], args.topn, ring[k % 64])
            for i in range(20):
                step(i, i)
            eng.enqueue_flush()
            torch.cuda.synchronize()
            eng.set_timing(8)
            c0 = eng.replica_counters()
            t0 = time.perf_counter()
            for i in range(args.steps):
                step(20 + i, i)
            eng.enqueue_flush()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            st = eng.stats()
            c1 = eng.replica_counters()
            eng.set_timing(0)
            out[name] = {"us_per_step": round(dt * 1e6, 2), "queries_per_s": round(1.0 / dt, 1),
                         "scan_kernel_us": round(float(st.last_scan_ms) * 1e3, 2),
                         "rescored_per_query": round((c1["rescored_rows"] - c0["rescored_rows"]) / args.steps, 1),
                         "grid_blocks": int(st.replica_grid_blocks)}
            # one query alone, synchronous API
            lat = []
            for i in range(200):
                t1 = time.perf_counter()
                if args.by_value:
                    eng.query_topn(qvecs[i], -1, args.topn)
                else:
                    eng.query_row_topn(rows[i], args.topn)
                lat.append(time.perf_counter() - t1)
            lat.sort()
            out[name]["p50_us"] = round(lat[len(lat) // 2] * 1e6, 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
