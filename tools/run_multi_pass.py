#!/usr/bin/env python3
"""Profiling driver: micro-batched multi-query passes over a 10 M x 12 catalogue
(the `microbatch` leg of bench.py on its own), so that rocprofv3 --pmc passes see
mi355::scan_multi_kernel without the rest of the benchmark.
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS ... --output-format csv -d gpurun_out/x -- python3 tools/run_multi_pass.py
"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--topn", type=int, default=100)
    ap.add_argument("--queries", type=int, default=72)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--single", type=int, default=5, help="single-query passes as well (scan_kernel)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from spotify_recommender_amd import CosineEngine
    from spotify_recommender_amd.synth import synthetic_catalogue

    t = synthetic_catalogue(args.rows, seed=12345)
    rows = np.array([(k * 7919) % args.rows for k in range(args.queries)], dtype=np.int64)
    q = t[torch.from_numpy(rows).cuda()].cpu().numpy()
    keys = torch.zeros(args.queries * args.topn, dtype=torch.int64, device="cuda")
    with CosineEngine(t) as eng:
        eng.set_batch_path(1)   # the exact multi-query passes (AUTO would send 72 queries down the batched path)
        for _ in range(args.reps):
            eng.enqueue_batch_keys(q, rows, args.topn, keys)
        for k in range(args.single):
            eng.enqueue_row_keys(int(rows[k]), args.topn, keys[: args.topn])
        torch.cuda.synchronize()
    print("done", flush=True)


if __name__ == "__main__":
    main()
