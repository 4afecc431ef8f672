#!/usr/bin/env python3
"""Where a streamed single-query launch over the 8-bit replica spends its time.  Needs a library built
with -DMI355REC_PHASE_CLOCK, passed with --lib (tools/phase_clock.sh builds it under gpurun_out/, never over the
product's libmi355rec.so):
every workgroup stamps a 100 MHz wall clock at entry (0), query ready (1), launch-wide cutoff ready (2),
tiles done (3), list stored (4); the merger and the seed riders stamp entry (0) and exit (5).
Printed: per phase, the median and the latest workgroup relative to the first workgroup's entry, for the
LAST launch of a stream of --steps queries."""
import argparse
import ctypes
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--topn", type=int, default=10)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--lib", required=True, help="the instrumented library (tools/phase_clock.sh builds it)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from spotify_recommender_amd import CosineEngine, capi
    from spotify_recommender_amd.synth import synthetic_catalogue

    capi.LIB_PATH = Path(args.lib).resolve()   # before the first capi.lib(): this process only
    lib = capi.lib()
    fn = lib.mi355rec_debug_phase_clock
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    t = synthetic_catalogue(args.rows, seed=12345)
    rows = [(k * 7919 + 13) % args.rows for k in range(args.steps + 2)]
    ring = torch.zeros((4, args.topn), dtype=torch.int64, device="cuda")
    with CosineEngine(t) as eng:
        eng.set_replica(capi.REPLICA_ON)
        for i in range(args.steps):
            eng.enqueue_row_keys_streamed(rows[i], args.topn, ring[i % 4])
        # the last launch with riders is the one made by the LAST enqueue (query steps-2); the flush launches
        # query steps-1 without riders and would overwrite the stamps, so read before flushing
        torch.cuda.synchronize()
        buf = np.zeros(1024 * 8, dtype=np.uint64)
        assert fn(buf.ctypes.data, buf.size) == 0
        eng.enqueue_flush()
        torch.cuda.synchronize()
        grid = int(eng.stats().replica_grid_blocks)
    c = buf.reshape(1024, 8).astype(np.int64)
    live = c[:, 0] > 0
    c = c[live]
    t0 = c[:, 0].min()
    scan = c[(c[:, 4] > 0) & (c[:, 5] == 0)]      # workgroups that scan from the start
    other = c[c[:, 5] > 0]                         # the merger, then the seed riders (they scan after their job)
    us = lambda x: round(float(x) / 100.0, 2)
    out = {"rows": args.rows, "topn": args.topn, "workgroups": int(live.sum()), "scanners": len(scan), "grid_blocks": grid}
    names = ["entry", "query_ready", "cutoff_ready", "tiles_done", "list_stored"]
    for i, nm in enumerate(names):
        v = scan[:, i] - t0
        out[nm] = {"first": us(v.min()), "median": us(np.median(v)), "last": us(v.max())}
    if (scan[:, 6] > 0).any():   # (builds that stamp it: the workgroup's sample values have arrived)
        v = scan[scan[:, 6] > 0][:, 6] - t0
        out["sample_arrived"] = {"first": us(v.min()), "median": us(np.median(v)), "last": us(v.max())}
    d = scan[:, 1:5] - scan[:, 0:4]
    out["per_workgroup_phase_us_median"] = {f"{names[i]}->{names[i + 1]}": us(np.median(d[:, i])) for i in range(4)}
    # does a workgroup's speed depend on where it runs?  (blockIdx % 8 = XCD under the round-robin dispatch)
    ids = np.nonzero(live)[0][(c[:, 4] > 0) & (c[:, 5] == 0)]
    tiles = scan[:, 3] - scan[:, 2]
    out["tile_phase_us_by_bid_mod8"] = [us(np.median(tiles[ids % 8 == x])) for x in range(8)]
    out["tile_phase_us_by_bid_mod8_max"] = [us(tiles[ids % 8 == x].max()) for x in range(8)]
    out["tile_phase_us_by_bid_quarter"] = [us(np.median(tiles[(ids * 4 // (ids.max() + 1)) == x])) for x in range(4)]
    out["tile_phase_us_percentiles_5_25_50_75_95"] = [us(np.percentile(tiles, p)) for p in (5, 25, 50, 75, 95)]
    if len(other):
        # merger = the first non-scanner; riders after it
        out["merger"] = {"entry": us(other[0, 0] - t0), "exit": us(other[0, 5] - t0)}
        if len(other) > 1:
            r = other[1:]
            out["riders"] = {"n": len(r), "entry_first": us((r[:, 0] - t0).min()), "exit_last": us((r[:, 5] - t0).max()),
                             "duration_median": us(np.median(r[:, 5] - r[:, 0]))}
    out["launch_span_us"] = us(max(c[:, 4].max(), c[:, 5].max()) - t0)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
