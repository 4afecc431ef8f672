#!/usr/bin/env python3
"""Do two streams of single queries overlap on the device?  Two handles over the same borrowed matrix, each
on its own HIP stream, queries alternating between them, against one handle taking all of them: if the
pair is faster, the fixed cost of a launch (dispatch gap, prologue, list store) hides behind the other
stream's scan, and a handle that alternates between two internal lanes would gain the same."""
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
    ap.add_argument("--steps", type=int, default=600)
    args = ap.parse_args()
    import torch
    from spotify_recommender_amd import CosineEngine
    from spotify_recommender_amd.synth import synthetic_catalogue

    t = synthetic_catalogue(args.rows, seed=12345)
    rows = [(k * 7919 + 13) % args.rows for k in range(args.steps + 40)]
    ring = torch.zeros((64, args.topn), dtype=torch.int64, device="cuda")
    out = {"rows": args.rows, "topn": args.topn}
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with CosineEngine(t) as a, CosineEngine(t) as b:
        for name, lanes in (("one_handle", [(a, s1)]), ("two_handles_two_streams", [(a, s1), (b, s2)]),
                            ("two_handles_one_stream", [(a, s1), (b, s1)])):
            for rep in range(2):   # first pass warms up
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    eng, st = lanes[i % len(lanes)]
                    eng.enqueue_row_keys_streamed(rows[i], args.topn, ring[i % 64], stream=st)
                for eng, st in lanes:
                    eng.enqueue_flush(stream=st)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.steps
            out[name] = {"us_per_query": round(dt * 1e6, 2), "queries_per_s": round(1 / dt, 1)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
