#!/usr/bin/env python3
"""Headline benchmark: queries/sec of cosine top-100 over a 10 M x 12 fp32
synthetic catalogue (BASELINE.json configs[2]; configs[3] when --gpus > 1).

A step = one query = one fused streaming pass over the catalogue (row-sharded
over the ranks when N > 1, merged with ONE all-gather of 100 packed keys per
rank) + the device merge.  The catalogue is resident in HBM before the timed
region; results stay on the device (800 B of keys per query).  Outside the
timed region the same line carries `microbatch` (12 queries per exact pass) and
`batched` (BASELINE configs[4]'s 1024-query batches on the matrix-core path,
with its compute roofline).

  python bench.py --gpus 1 --steps 300 --warmup 30
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
`roofline` (scan kernel vs the 8 TB/s HBM peak, HIP-event timed inside the
timed region) and `cpu_baseline` (the oracle on this box's host cores,
N=1 only, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from spotify_recommender_amd.benchlegs import (BYTES_PER_ROW, FLOP_PER_PAIR, FP16_MFMA_PEAK_TFLOPS, FP32_PEAK_TFLOPS,  # noqa: E402
                                               HBM_PEAK_GBPS, clustered_object, config0_object, host_threads, preflight,
                                               peer_access_matrix, probe_concurrent_gbps, probe_gbps_of, rank_group_report, roofline_bound,
                                               stream_leg, traffic_fields)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--topn", type=int, default=100)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="skip the per-kernel HIP events (roofline.achieved becomes null)")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="HIP events around every k-th scan launch of the UNTIMED pass that fills roofline.avg_kernel_ms "
                         "(0 = choose so that 16 launches are timed, 3 in a short run)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is run this many times back to back (each between barrier + synchronize pairs); "
                         "`value` is the median run, `value_runs` lists them all")
    ap.add_argument("--batch", type=int, default=1024, help="queries per call of the batched (configs[4]) leg")
    ap.add_argument("--no-batched", action="store_true")
    ap.add_argument("--no-c5-shard", action="store_true",
                    help="skip the second batched leg (a 12.5 M-row shard of BASELINE configs[4], its own catalogue)")
    ap.add_argument("--placement", choices=["sharded", "replicated"], default="sharded",
                    help="N > 1, single process: rows split over the devices (north_star), or every device holds all rows "
                         "and serves whole windows of the stream (include/mi355rec.h, PLACEMENT)")
    ap.add_argument("--window", type=int, default=16,
                    help="N > 1: single queries whose per-rank keys share one all-gather")
    ap.add_argument("--no-streamed", action="store_true",
                    help="merge every query in its own launch instead of inside the next query's scan launch")
    ap.add_argument("--replica-fp16", action="store_true",
                    help="A/B (an MI355REC_EXPERIMENTS build of the library only): single queries scan the 24 B/row fp16 "
                         "replica instead of the 12 B/row 8-bit one")
    ap.add_argument("--no-replica", action="store_true",
                    help="scan the fp32 rows (48 B/row, the reference's own traffic) instead of a replica")
    ap.add_argument("--latency-queries", type=int, default=1000)
    ap.add_argument("--catalogue", choices=["uniform", "clustered-contiguous"], default="uniform",
                    help="clustered-contiguous: the `clustered` object (every route over a catalogue sorted by genre, "
                         "spotify_recommender_amd/synth.py) covers 3000 and 300 contiguous clusters with and without the genre "
                         "ramp instead of the one shape the default run carries")
    ap.add_argument("--no-clustered", action="store_true", help="skip the `clustered` object")
    ap.add_argument("--lanes", type=int, default=0,
                    help="lanes of the handle the timed stream of single queries is dealt over (mi355rec_create_lane: the same rows and "
                         "replicas, own stream state, own HIP stream; 0 = 2 for the single-process stream, 1 = one handle as until round 4)")
    ap.add_argument("--no-single-lane", action="store_true",
                    help="skip the `single_lane` leg (the same stream through ONE handle): for profiling runs whose kernel averages must be "
                         "those of the timed region alone")
    ap.add_argument("--no-config0", action="store_true",
                    help="skip the `config0` object (BASELINE configs[0]: a 114 000-track CSV -> songs_data.bin -> top-10 on the GPU "
                         "path and on the product's CPU backend)")
    ap.add_argument("--preflight", action="store_true",
                    help="--gpus N, single process: print the peer-access matrix, the placement AUTO would choose and the device "
                         "memory per shard, run ONE query per transport against the oracle, and exit before any timing")
    ap.add_argument("--virtual-shards", type=int, default=0,
                    help="drive the product's single-process row-sharded engine (mi355rec_create_sharded_on) with this "
                         "many shards of ONE GPU: the multi-GPU orchestration rehearsed on a one-GPU box")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: resolve the launch mode, load the C-ABI library, print the plan as one JSON line")
    ap.add_argument("--transport", choices=["auto", "peer", "rccl"], default="auto",
                    help="N > 1, single process: how the per-shard key lists meet (default: peer stores where every "
                         "device can map the first one's memory, else one ncclAllGather per rank)")
    return ap.parse_args()


def workload_label(n: int, topn: int) -> str:
    """Which BASELINE.json config a (rows, topn) pair is, if any."""
    if n == 10_000_000 and topn == 100:
        return "BASELINE configs[2], HBM-roofline run"
    if n == 1_000_000 and topn == 10:
        return "BASELINE configs[1]"
    return "not a BASELINE config: same path, other size"


def metric_label(n: int, topn: int) -> str:
    rows = f"{n // 1_000_000}M" if n % 1_000_000 == 0 else str(n)
    return f"queries/sec, cosine top-{topn} over a {rows} x 12 fp32 catalogue"


def replica_desc(st):
    """What a single query streams, from the handle's stats: (bytes per row, algorithmic bytes per launch,
    kernel, its name in the PMC file, dtype label, short replica label)."""
    rb = int(st.replica_single_row_bytes) if st.replica_active else 0
    if rb == 12:
        return (12, int(st.replica_single_bytes_per_query), "mi355::scan_q8_kernel<Q8Cfg<512,4,2>, {targs}>",
                "scan_q8_kernel", "f32 (every score is the exact fp32 chain; a 12 B/row 8-bit replica only rules rows out)",
                "8-bit replica (12 B/row")
    if rb == 24:
        return (24, int(st.replica_single_bytes_per_query), "mi355::scan_half_kernel<HalfCfg<512,4,2>, {targs}>",
                "scan_half_kernel", "f32 (every score is the exact fp32 chain; a 24 B/row fp16 replica only rules rows out)",
                "fp16 replica (24 B/row")
    return (BYTES_PER_ROW, int(st.bytes_per_query), "mi355::scan_kernel<ScanCfg<512,1,6,2>, {targs}, 0, true>",
            "scan_kernel", "f32", "fp32 rows (48 B/row")


def cpu_baseline(feats_host, topn, query_rows):
    """The oracle timed on the host cores: B1 (OpenMP rows + per-thread top-N) is the reported value — the portable
    build the checker uses (gcc -O3, no -march: the reference's own flags, Makefile:9) — with the same source built
    -march=native beside it and B0 (the reference's serial loop + heap over its AoS layout) riding along.  Bounded
    samples: ~8 + 4 + 6 s."""
    from oracle import oracle

    threads, host = host_threads(oracle.max_threads())
    oracle.recommend_omp(feats_host, query_rows[0], topn, threads)  # touch pages / spin up the team
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < 8.0:          # ~8 s of all-core CPU work
        oracle.recommend_omp(feats_host, query_rows[done % len(query_rows)], topn, threads)
        done += 1
    omp_qps = done / (time.perf_counter() - t0)
    native = None
    try:
        run_native, native_flags = oracle.native_recommend_omp()
        run_native(feats_host, query_rows[0], topn, threads)
        t0 = time.perf_counter()
        nd = 0
        while time.perf_counter() - t0 < 4.0:
            run_native(feats_host, query_rows[nd % len(query_rows)], topn, threads)
            nd += 1
        native = {"value": round(nd / (time.perf_counter() - t0), 3), "unit": "queries/s", "cores": threads,
                  "flags": native_flags, "sample": f"{nd} queries (4 s)"}
    except Exception as e:   # no compiler on the box: the portable figure stands alone
        native = {"unavailable": str(e)[:200]}
    # B0: the reference's serial loop over its AoS layout —
    # features at a 152-byte stride inside vector<Song> (Song.h:21-32)
    import numpy as np
    aos = np.zeros((feats_host.shape[0], 38), dtype=np.float32)
    aos[:, 25:37] = feats_host
    aos_feats = aos[:, 25:37]
    oracle.recommend_by_index(aos_feats, query_rows[0], topn)
    t0 = time.perf_counter()
    serial = 0
    while time.perf_counter() - t0 < 6.0:          # ~6 s of single-core CPU work
        oracle.recommend_by_index(aos_feats, query_rows[serial % len(query_rows)], topn)
        serial += 1
    serial_qps = serial / (time.perf_counter() - t0)
    return {
        "value": round(omp_qps, 3), "unit": "queries/s", "cores": threads, "kind": "port",
        "sample": f"{done} queries ({8.0:.0f} s) x {feats_host.shape[0]} rows top-{topn}, OpenMP rows + per-thread top-N "
                  f"(oracle/cosine_oracle.c)",
        "flags": "gcc -std=c11 -O3 -fopenmp -ffp-contract=off, no -march (the reference's own: Makefile:9)",
        "host": host,
        "march_native": native,
        "serial_reference_loop_qps": round(serial_qps, 3),
        "serial_sample": f"{serial} queries (6 s), 1 core, 152-byte AoS row stride (Recommender.cu:256-318 restated)",
    }


def run_node(args, json_fd):
    """`--gpus N` launched plainly (or `--virtual-shards G`): ONE process drives every shard through the
    product's row-sharded C-ABI handle (mi355rec_create_sharded / _on, csrc/sharded.hip) — the engine the
    C++ Recommender uses in place of the reference's cudaSetDevice(0) (Recommender.cu:124).  A step is one
    query = one streamed scan launch on EVERY shard; the per-shard key lists of `--window` queries share one
    exchange (peer stores or one ncclAllGather per rank) and one batched merge whose results land in host
    memory.  The flush of the last window and the wait for the last result are inside the timed region."""
    import numpy as np
    import torch

    from spotify_recommender_amd import capi
    from spotify_recommender_amd.engine import CosineEngine, NodeEngine
    from spotify_recommender_amd.synth import synthetic_catalogue

    virtual = args.virtual_shards > 1
    g = args.virtual_shards if virtual else args.gpus
    visible = torch.cuda.device_count()
    if not virtual and visible < g:
        raise SystemExit(f"--gpus {g} but only {visible} device(s) visible (use --virtual-shards {g} to rehearse on one)")
    devices = [0] * g if virtual else list(range(g))
    n, topn = args.rows, args.topn
    total_q = args.warmup + args.steps
    q_rows = [(k * 7919) % n for k in range(total_q + args.latency_queries)]
    dev0 = torch.device("cuda", 0)
    full = synthetic_catalogue(n, seed=args.seed, device=dev0)
    feats_host = full.cpu().numpy()
    if not virtual:
        del full
        torch.cuda.empty_cache()

    if args.preflight:
        from oracle import oracle   # (the checker, handed to the leg: the package itself never imports it)
        return preflight(args, json_fd, torch, np, capi, NodeEngine, feats_host, devices, virtual, oracle)
    replicated = args.placement == "replicated"
    node = NodeEngine(feats_host, devices=devices,
                      placement=capi.PLACEMENT_REPLICATED if replicated else capi.PLACEMENT_SHARDED)
    if node.placement() == capi.PLACEMENT_CPU:   # (cannot happen with an explicit device list; never measure the CPU backend)
        raise RuntimeError("the node handle is served by the CPU backend: no HIP device is visible to libmi355rec.so")
    info = node.info()
    node.set_window(args.window)
    # `value` is measured with one streamed scan launch per shard per QUERY (each query its own pass over
    # every shard: the N = 1 line's step, sharded); the same stream with BATCHED windows (a window of
    # queries = one multi-query pass per shard) is reported beside it as `batched_windows`.
    node.set_window_mode(False)
    # Which transport `value` is measured on.  north_star names "a single RCCL all-gather over xGMI" for BASELINE configs[3]
    # (10 M rows x top-100 row-sharded over real devices): there `value` is the RCCL transport unless --transport says
    # otherwise; everywhere else it is the handle's default (peer stores where every device maps the first one's memory).
    # BOTH transports are timed and reported as first-class objects (`transport.rccl`, `transport.peer`).
    configs3 = (not virtual) and g > 1 and (not replicated) and n == 10_000_000 and topn == 100
    transport_note = None
    if args.transport == "rccl" or (args.transport == "auto" and configs3):
        try:
            node.set_transport(capi.TRANSPORT_RCCL)
        except capi.Mi355Error as e:
            if args.transport == "rccl":
                raise
            transport_note = f"the RCCL transport is not available here ({e}): `value` is the default transport"
    elif args.transport == "peer":
        node.set_transport(capi.TRANSPORT_PEER)
    replica_mode = capi.REPLICA_FP16 if args.replica_fp16 else capi.REPLICA_AUTO
    if args.no_replica:
        node.set_replica(capi.REPLICA_OFF)
    elif args.replica_fp16:
        node.set_replica(replica_mode)
    transport = node.info()["transport"]

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)

    def stream(lo_k, hi_k):
        t = -1
        for k in range(lo_k, hi_k):
            t = node.enqueue_row(q_rows[k], topn)
        node.enqueue_flush()
        return node.wait(t, topn)

    def timed_stream(events=True):
        """One K-step region between synchronisations (no HIP events inside it), then — `events` — an UNTIMED pass of the same
        stream with events around every stride-th scan launch of every shard, for the kernels' mean durations."""
        stream(0, args.warmup)
        sync_all()
        st0 = node.stream_stats()
        t0 = time.perf_counter()
        last = stream(args.warmup, total_q)
        sync_all()
        dt = time.perf_counter() - t0
        st1 = node.stream_stats()
        shard_ms = []
        if events and not args.no_kernel_events:
            stride = args.event_stride if args.event_stride > 0 else max(1, -(-args.steps // (16 if args.steps >= 64 else 3)))
            node.set_timing(stride)
            stream(args.warmup, total_q)
            sync_all()
            shard_ms = [float(node.shard_stats(r).last_scan_ms) for r in range(g) if info["shard_rows"][r] > 0]
            node.set_timing(0)
        host_us = (st1["host_ns"] - st0["host_ns"]) / max(1, st1["queries"] - st0["queries"]) / 1e3
        return dt, last, shard_ms, host_us, st1["exchanges"] - st0["exchanges"]

    # the timed region is run `--repeats` times back to back; `value` is the median run (as at N = 1)
    runs = [timed_stream(events=(i == 0)) for i in range(max(1, args.repeats))]
    run_times = [r[0] for r in runs]
    mid = sorted(range(len(runs)), key=lambda i: run_times[i])[len(runs) // 2]
    elapsed, last_result, _, host_us, exchanges = runs[mid]
    shard_ms = runs[0][2]
    st_headline = node.shard_stats(0)   # which copy of the rows the headline's queries scan
    replica = bool(st_headline.replica_active)
    rows_local = max(info["shard_rows"])

    # one query alone, end to end, through the synchronous call the C++ Recommender makes
    lat = []
    for k in range(total_q, total_q + args.latency_queries):
        t1 = time.perf_counter()
        node.query_row_topn(q_rows[k], topn)
        lat.append((time.perf_counter() - t1) * 1e3)
    lat.sort()

    # the same stream over the fp32 rows (SURVEY.md §8(d)'s 48 B per row), for the survey-priced roofline
    fp32_rows = None
    if replica:
        node.set_replica(capi.REPLICA_OFF)
        dt32, _, ms32, host32, _ = timed_stream()
        node.set_replica(replica_mode)
        k_ms = sum(ms32) / len(ms32) if ms32 else 0.0
        fp32_rows = {"ms_per_step": round(dt32 / args.steps * 1e3, 5), "value": round(args.steps / dt32, 2), "unit": "queries/s",
                     "host_enqueue_us_per_query": round(host32, 2),
                     "roofline": {"bound": roofline_bound(rows_local * BYTES_PER_ROW), "algorithmic_bytes_per_launch": rows_local * BYTES_PER_ROW,
                                  "avg_kernel_ms": round(k_ms, 5),
                                  "achieved": round(rows_local * BYTES_PER_ROW / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(rows_local * BYTES_PER_ROW / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if k_ms > 0 else None}}

    # the same stream with batched windows: a window of queries is ONE multi-query pass per shard
    batched_windows = None
    if topn <= 128 and args.window >= 2:
        node.set_window_mode(True)
        dtb, lastb, msb, hostb, exb = timed_stream()
        node.set_window_mode(False)
        kb = sum(msb) / len(msb) if msb else 0.0
        alg_b = (rows_local + 1) // 2 * 48
        batched_windows = {
            "value": round(args.steps / dtb, 2), "unit": "queries/s", "ms_per_step": round(dtb / args.steps * 1e3, 5),
            "queries_per_pass": min(args.window, 32), "launches_per_shard_per_window": 3 * ((args.window + 31) // 32),
            "host_enqueue_us_per_query": round(hostb, 2), "exchanges": exb,
            "kernel": "mi355::scan_half_multi_kernel (fp16 matrix-core pre-filter over the 24 B/row replica, up to 32 queries per pass)",
            "avg_pass_kernel_ms": round(kb, 5),
            "pass_gbps": round(alg_b / (kb * 1e-3) / 1e9, 1) if kb > 0 else None,
            "note": "throughput mode: a query starts when its window closes; results identical (tests/test_gpu_node.py)"}
        last_batched = lastb

    # BOTH transports as first-class objects (real placements only: RCCL wants one device per shard); the one `value` was
    # measured on keeps the headline's figures, the other is timed here over the same stream
    def transport_name(t):
        return "peer" if t == capi.TRANSPORT_PEER else "rccl"

    transports = {"value_is": transport_name(transport), "rccl": None, "peer": None, "note": transport_note,
                  "value_rule": "BASELINE configs[3] (10 M x top-100, row-sharded over real devices): RCCL, as north_star words it; "
                                "elsewhere the handle's default; --transport overrides"}
    transports[transport_name(transport)] = {"value": round(args.steps / elapsed, 2), "unit": "queries/s", "ms_per_step": round(elapsed / args.steps * 1e3, 5),
                                             "exchanges": exchanges, "host_enqueue_us_per_query": round(host_us, 2)}
    if not virtual and g > 1 and not replicated:
        other_t = capi.TRANSPORT_RCCL if transport == capi.TRANSPORT_PEER else capi.TRANSPORT_PEER
        try:
            node.set_transport(other_t)
            dt2, _, _, host2, ex2 = timed_stream()
            transports[transport_name(other_t)] = {"value": round(args.steps / dt2, 2), "unit": "queries/s", "ms_per_step": round(dt2 / args.steps * 1e3, 5),
                                                   "exchanges": ex2, "host_enqueue_us_per_query": round(host2, 2)}
        except capi.Mi355Error as e:
            transports[transport_name(other_t)] = {"unavailable": str(e)}
        node.set_transport(transport)
    if transports["rccl"] is not None and "value" in transports["rccl"]:
        # what RCCL itself says: ncclCommCount of the communicators the exchange ran on (mi355rec_sharded_rccl_ranks)
        transports["rccl"].update(node.rccl_ranks())
        transports["rccl"]["via"] = "one ncclAllGather per rank per window, issued by that rank's worker thread (ncclCommInitAll)"

    # virtual shards: the whole catalogue through ONE single-device handle on the same GPU, same stream of
    # queries — the difference is what the orchestration (G launches per query, exchange, second merge) costs
    single = None
    if virtual:
        eng = CosineEngine(full)
        if args.no_replica:
            eng.set_replica(capi.REPLICA_OFF)
        ring = [torch.zeros(topn, dtype=torch.int64, device=dev0) for _ in range(4)]
        for k in range(args.warmup):
            eng.enqueue_row_keys_streamed(q_rows[k], topn, ring[k % 4])
        eng.enqueue_flush()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(args.warmup, total_q):
            eng.enqueue_row_keys_streamed(q_rows[k], topn, ring[k % 4])
        eng.enqueue_flush()
        torch.cuda.synchronize()
        sdt = (time.perf_counter() - t1) / args.steps
        lat1 = []
        for k in range(total_q, total_q + min(200, args.latency_queries)):
            t1 = time.perf_counter()
            eng.query_row_topn(q_rows[k], topn)
            lat1.append((time.perf_counter() - t1) * 1e3)
        lat1.sort()
        eng.close()
        single = {"ms_per_step": round(sdt * 1e3, 5), "value": round(1.0 / sdt, 2), "p50_ms": round(lat1[len(lat1) // 2], 4) if lat1 else None,
                  "orchestration_ms_per_query": round(elapsed / args.steps * 1e3 - sdt * 1e3, 5),
                  "note": "the same 10 M-row stream through one mi355rec handle on the same GPU; orchestration = sharded step - this"}

    # results of the run against the oracle (checker only): the last streamed query of the timed region,
    # plus a few through the synchronous call
    from oracle import oracle
    ok, checked = True, 0
    rows_to_check = [(q_rows[total_q - 1], last_result)]
    if batched_windows is not None:
        rows_to_check.append((q_rows[total_q - 1], last_batched))
    for k in (0, total_q // 2, total_q + args.latency_queries - 1):
        rows_to_check.append((q_rows[k], node.query_row_topn(q_rows[k], topn)))
    for row, (idx, sc) in rows_to_check:
        want = oracle.scores(feats_host, feats_host[row], threads=0)
        ci, cs = oracle.topn_canonical(want, row, topn)
        ok = ok and idx.tolist() == ci.tolist() and bool(np.array_equal(sc, cs + np.float32(0)))
        checked += 1

    k_ms = sum(shard_ms) / len(shard_ms) if shard_ms else 0.0
    row_bytes, alg, kernel_label, kernel_name, dtype_label, replica_label = replica_desc(st_headline)
    kernel_label = kernel_label.format(targs="true, true") + f" ({replica_label})"
    achieved = alg / (k_ms * 1e-3) / 1e9 if k_ms > 0 else None
    line = {
        "metric": metric_label(n, topn), "value": round(args.steps / elapsed, 2), "unit": "queries/s",
        "n_gpus": 1 if virtual else g, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None,
        "dtype": dtype_label,
        "data": "synthetic",
        "value_runs": [round(args.steps / t, 1) for t in run_times],
        "value_note": f"median of {len(run_times)} back-to-back {args.steps}-step regions, each between synchronisations of every device",
        "config": {
            "workload": f"{n} synthetic tracks x 12 fp32 features, top-{topn}, "
                        + ("REPLICATED on " if replicated else "row-sharded over ")
                        + (f"{g} VIRTUAL {'replicas' if replicated else 'shards'} of one MI355X (orchestration rehearsal)" if virtual else f"{g} MI355X")
                        + (" (BASELINE configs[3])" if (n == 10_000_000 and topn == 100 and not replicated) else ""),
            "engine": "ONE process, the product's C-ABI: mi355rec_create_placed (csrc/sharded.hip)",
            "placement": args.placement,
            "placement_note": ("every device holds all rows and serves whole windows of the stream, round-robin: no exchange"
                               if replicated else
                               "rows split into contiguous blocks, one per device; one exchange + one merge per window (north_star)"),
            "rows": n, "topn": topn, "shards": g, "virtual_shards": virtual, "devices": devices,
            "rows_per_shard": rows_local, "queries_per_step": 1,
            "transport": transport_name(transport),
            "window": args.window,
            "window_mode": ("streamed: one scan launch per query on the window's replica" if replicated
                            else "streamed: one scan launch per shard per query"), "exchanges_in_timed_region": exchanges,
            "rows_by_pointer": node.rows_by_pointer(), "note": node.note(),
            "merge": "per shard inside the next query's scan launch (streamed); one exchange + one batched merge per window; "
                     "last window flushed and its result awaited inside the timed region",
            "seed": args.seed, "generator": "torch.rand(seed) uniform[0,1) on device 0, sharded from host memory",
        },
        "p50_ms": round(lat[len(lat) // 2], 4) if lat else None,
        "p99_ms": round(lat[min(len(lat) - 1, int(len(lat) * 0.99))], 4) if lat else None,
        "roofline": {   # (<= 20 scalar keys; per launch of one shard, mean over the shards; every shard's own figures: `per_gpu`)
            "bound": "hbm",
            "kernel": kernel_label,
            "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4) if achieved else None,
            "survey_frac": None, "survey_achieved": None, "survey_kernel_us": None,
            "traffic": None,
            "algorithmic_bytes_per_launch": alg, "bytes_per_row": row_bytes,
            "avg_kernel_ms": round(k_ms, 5) if k_ms else None,
            "cache_resident": bool(alg <= 128 * 2**20),
            "per": "shard launch (mean over shards)",
        },
        "roofline_notes": {"survey": "survey_* = the same stream over the fp32 rows (48 B/row, SURVEY.md §8(d)): the `fp32_rows` object",
                           **traffic_fields()},
        "host": {"enqueue_us_per_query": round(host_us, 2), "launches_per_query": 1 if replicated else g,
                 "note": "wall time of the enqueue/flush calls on the one host thread (all shards), per query"},
        "verified_against_oracle": bool(ok), "verified_queries": checked,
    }
    if fp32_rows is not None:
        line["fp32_rows"] = fp32_rows
        line["roofline"]["survey_frac"] = fp32_rows["roofline"]["frac"]
        line["roofline"]["survey_achieved"] = fp32_rows["roofline"]["achieved"]
        line["roofline"]["survey_kernel_us"] = round(fp32_rows["roofline"]["avg_kernel_ms"] * 1e3, 2)
    elif not replica:
        line["roofline"]["survey_frac"] = line["roofline"]["frac"]
        line["roofline"]["survey_achieved"] = line["roofline"]["achieved"]
        line["roofline"]["survey_kernel_us"] = round(k_ms * 1e3, 2) if k_ms else None
    if batched_windows is not None:
        line["batched_windows"] = batched_windows
    line["transport"] = transports
    line["peer_access"] = peer_access_matrix(torch, devices)
    line["all_pairs_peer"] = all(all(v.values()) for v in line["peer_access"].values())
    # every shard's own scan kernel against the HBM peak (HIP events on the shard's stream)
    live = [r for r in range(g) if info["shard_rows"][r] > 0]
    line["per_gpu"] = [{"shard": r, "device": devices[r], "rows": info["shard_rows"][r], "avg_kernel_ms": round(m, 5),
                        "achieved_gbps": round(alg / (m * 1e-3) / 1e9, 1) if m > 0 else None,
                        "frac": round(alg / (m * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if m > 0 else None}
                       for r, m in zip(live, shard_ms)]
    if single is not None:
        line["single_engine_same_gpu"] = single
    node.close()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(line) + "\n").encode())


def main():
    args = parse_args()
    # The contract is ONE JSON line on stdout.  RCCL prints its version banner (and warnings)
    # to the process's stdout from C, so file descriptor 1 is pointed at stderr for the whole
    # run and the JSON line is written to the saved descriptor at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    # How this HOST waits for the GPU: completion signals are POLLED, not interrupt-driven (HSA_ENABLE_INTERRUPT=0, what a
    # latency-sensitive serving host sets; must be in the environment before the runtime initialises, the caller's own setting
    # wins).  It shortens the way back from torch.cuda.synchronize() by ~10 us — nothing in a 300-step region, 3-7 % of the
    # driver's 20-step one (0.39 ms: profiles/r06_region_timeline_20steps.txt; A/B: docs/LAB_NOTES.md R6.6) — and is reported
    # in the line (`config.host_wait`).  The library's own synchronous calls spin on a word in pinned memory either way.
    # (One process on one or more GPUs only: under torch.distributed.run the runtime's default stays — RCCL's own threads wait
    # on the same signals, and that combination has never run on this build's boxes.)
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("BENCH_FORCE_SHARDED") != "1":
        os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    from spotify_recommender_amd import CosineEngine, ShardedEngine, shard_bounds
    from spotify_recommender_amd.engine import unpack_keys
    from spotify_recommender_amd.synth import synthetic_catalogue

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_run:
        from spotify_recommender_amd import capi
        capi.lib()   # the product library loads and exports every declared symbol
        mode = ("node: one process, mi355rec_create_placed (%s) over %d %s" % (
                    args.placement, args.virtual_shards if args.virtual_shards > 1 else args.gpus,
                    "virtual shards of device 0" if args.virtual_shards > 1 else "devices")
                if (world == 1 and (args.gpus > 1 or args.virtual_shards > 1)) else
                ("rank: one process per GPU over torch.distributed (RCCL), WORLD_SIZE=%d" % world if world > 1 else
                 "single: one mi355rec handle on device 0"))
        os.write(json_fd, (json.dumps({"dry_run": True, "mode": mode, "metric": metric_label(args.rows, args.topn),
                                       "rows": args.rows, "topn": args.topn, "gpus": args.gpus,
                                       "devices_visible": torch.cuda.device_count()}) + "\n").encode())
        return
    if world == 1 and (args.gpus > 1 or args.virtual_shards > 1 or args.preflight):   # (--preflight on one GPU: the node handle over device 0)
        # Launched plainly: ONE process drives every GPU through the product's C-ABI
        # (mi355rec_create_sharded: what the C++ Recommender uses).  Under torch.distributed.run
        # (WORLD_SIZE = N) the one-process-per-GPU path below runs instead.
        return run_node(args, json_fd)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1 and args.placement != "sharded":
        raise SystemExit("--placement replicated is a mode of the single-process node handle (launch bench.py plainly); "
                         "one process per GPU under torch.distributed is north_star's row sharding + one all-gather")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_sharded = os.environ.get("BENCH_FORCE_SHARDED") == "1"  # exercise the RCCL path on one GPU
    if world > 1 or force_sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    n, topn = args.rows, args.topn
    total_q = args.warmup + args.steps
    q_rows = [(k * 7919) % n for k in range(total_q + args.latency_queries)]

    # every rank generates the same catalogue from the same seed, keeps its rows
    full = synthetic_catalogue(n, seed=args.seed, device=dev)
    q_vecs = full[torch.tensor(q_rows, device=dev)].cpu().numpy()
    lo, hi = shard_bounds(n, world, rank)
    feats_host = full.cpu().numpy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    if world > 1:
        shard = full[lo:hi].clone()
        del full
    else:
        shard = full
    torch.cuda.empty_cache()

    eng = CosineEngine(shard, row_base=lo)
    from spotify_recommender_amd import capi
    if args.replica_fp16 and not capi.has_experiments():
        raise SystemExit("--replica-fp16 needs an MI355REC_EXPERIMENTS build of libmi355rec.so (single queries over the fp16 replica "
                         "are an A/B route since round 5)")
    replica_mode = capi.REPLICA_FP16 if args.replica_fp16 else capi.REPLICA_AUTO
    if args.no_replica:
        eng.set_replica(capi.REPLICA_OFF)
    elif args.replica_fp16:
        eng.set_replica(replica_mode)
    st_headline = eng.stats()       # which copy of the rows the headline's queries scan
    replica = bool(st_headline.replica_active)
    # (one process per GPU: every rank deals its windowed stream over lanes of its engine as well — ShardedEngine(lanes=))
    # (from 2 M rows per rank: below that a rank's step is bound by the host's per-query work, which a second lane adds to —
    # measured on one GPU through the same path (BENCH_FORCE_SHARDED=1), one lane / two: 10 M rows 39.6 k / 45.7 k queries/s, 5 M 54.2 / 67.2 k,
    # 2.5 M 72.0 / 77.0 k, 1.25 M 73.5 / 59.1 k)
    rank_lanes = args.lanes if args.lanes > 0 else (2 if (hi - lo) >= 2_000_000 else 1)
    sharded = ShardedEngine(eng, max_topn=topn, always_gather=force_sharded, lanes=rank_lanes) if (world > 1 or force_sharded) else None
    out_keys = torch.zeros(topn, dtype=torch.int64, device=dev)

    # A stream of single queries: the merge of query k rides in the scan launch of query
    # k + 1 (mi355rec_enqueue_row_keys_streamed), the last one is flushed INSIDE the timed
    # region; every step is still one full pass over the catalogue for one query.
    streamed = sharded is None and not args.no_streamed and topn <= 1024
    # LANES: a handle is one chain of launches (query k + 1's launch needs query k's lists and sample), so its launches never
    # overlap and the chip idles while one drains and the next ramps up; lanes (mi355rec_create_lane) are further handles
    # over the same rows and replicas, each on its own HIP stream, and the stream of queries is dealt over them in turn.
    n_lanes = args.lanes if args.lanes > 0 else (2 if (streamed and world == 1) else 1)
    if not streamed:
        n_lanes = 1
    lanes = [eng] + [eng.lane() for _ in range(n_lanes - 1)]
    if n_lanes > 1 and args.no_replica:
        for ln in lanes[1:]:
            ln.set_replica(capi.REPLICA_OFF)
    # (each lane on the stream the library created with it: those sit on different hardware queues — two streams from torch's
    # pool, created after the handles, landed on ONE queue and the lanes ran at a single handle's rate)
    lane_streams = [ln.own_stream() for ln in lanes] if n_lanes > 1 else [None]
    rings = [[torch.zeros(topn, dtype=torch.int64, device=dev) for _ in range(4)] for _ in lanes]
    torch.cuda.synchronize()

    def ring_of(k):
        return rings[k % n_lanes][(k // n_lanes) % 4]

    # (bound once, as a C host would call it: the row and the output pointer are all that changes per query)
    import ctypes
    ring_ptrs = [[ctypes.c_void_p(t.data_ptr()) for t in rs] for rs in rings]
    lane_calls = [ln.bound_enqueue_row_keys_streamed(topn, ls) for ln, ls in zip(lanes, lane_streams)] if streamed else []

    def step(k):
        if sharded is None:
            if streamed:
                lane_calls[k % n_lanes](q_rows[k], ring_ptrs[k % n_lanes][(k // n_lanes) % 4])
            else:
                eng.enqueue_row_keys(q_rows[k], topn, out_keys)
        else:
            # N > 1: the exchange is amortised over a window of queries (one all-gather and
            # one batched merge per WINDOW single-query passes), closed inside the timed region
            sharded.enqueue_query_windowed(q_vecs[k], q_rows[k], topn, window=args.window)

    def flush():
        if streamed:
            for ln, ls in zip(lanes, lane_streams):
                ln.enqueue_flush(stream=ls)
        if sharded is not None:
            sharded.flush_window()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    flush()
    fence()

    def lane_counters():
        cs = [ln.replica_counters() for ln in lanes]
        return {k: sum(c[k] for c in cs) for k in cs[0]}

    # THE TIMED REGION: exactly K steps between barrier + synchronize pairs, the flush of the last query inside it, no HIP
    # events inside it (an event pair costs a launch ~6 us of stream time).  A K-step region of 20 steps is
    # 0.4 ms — one sample of it is a noisy thing to steer by — so the region is run REPEATS times back to back, each bracketed
    # the same way: `value` is the MEDIAN run, `value_runs` all of them (item 1d).
    def timed_region():
        fence()
        t0 = time.perf_counter()
        for k in range(args.warmup, total_q):
            step(k)
        flush()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fence()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    rc_before = lane_counters() if replica else None
    run_times = [timed_region() for _ in range(max(1, args.repeats))]
    rc_after = lane_counters() if replica else None
    elapsed = sorted(run_times)[len(run_times) // 2]
    # ... and the kernel's duration from an UNTIMED pass of the same stream (same lanes, same queries): HIP events around every
    # stride-th scan launch of lane 0, on the stream the kernel is launched on
    stride = args.event_stride if args.event_stride > 0 else max(1, -(-args.steps // (16 if args.steps >= 64 else 3)))
    st = None
    if not args.no_kernel_events:
        eng.set_timing(stride)
        for k in range(args.warmup, total_q):
            step(k)
        flush()
        fence()
        st = eng.stats()
        eng.set_timing(False)
    # N > 1: what the process group the exchange ran on says about itself — every rank calls this (collectives inside)
    group = None
    if world > 1 or force_sharded:
        props = torch.cuda.get_device_properties(local_rank)
        row_bytes_h, alg_h = replica_desc(st_headline)[:2]
        group = rank_group_report(dist, torch, rank, world, dev, f"cuda:{local_rank} {props.name}", str(getattr(props, "uuid", local_rank)),
                                  float(st.last_scan_ms) if st is not None else 0.0, alg_h)
    # what the timed stream itself produced for its last two queries (checked against the oracle below)
    timed_tail = [(q_rows[k], ring_of(k).clone()) for k in (total_q - 2, total_q - 1)] if streamed and args.steps >= 2 else []
    # the same stream through ONE handle (what `value` was until round 4): a launch that has the chip to itself
    single_lane = None
    if n_lanes > 1 and not args.no_single_lane and rank == 0:
        single_lane, _ = stream_leg(eng, torch, q_rows, topn, min(args.steps, 300), min(max(args.warmup, 4), 20))

    # the same queries BY VALUE with nothing excluded (mi355rec_enqueue_query_keys_streamed(q, -1), mi355rec_query_topn(q, -1): a new
    # track) through one handle: their launch-wide bound comes from the neighbourhood of their anchor
    by_value = by_value_last = None
    if sharded is None and streamed and rank == 0 and not args.no_single_lane:
        nv = min(256, len(q_rows))
        v_vecs = np.ascontiguousarray(q_vecs[:nv], dtype=np.float32)
        by_value, by_value_last = stream_leg(eng, torch, q_rows[:nv], topn, min(args.steps, 300), min(max(args.warmup, 4), 20), vectors=v_vecs)
        lat_v = []
        for k in range(min(100, nv)):
            t1 = time.perf_counter()
            eng.query_topn(v_vecs[k], -1, topn)
            lat_v.append((time.perf_counter() - t1) * 1e3)
        lat_v.sort()
        by_value["one_query_alone_p50_ms"] = round(lat_v[len(lat_v) // 2], 4)
        by_value["note"] = "one handle; compare `single_lane` (the same rows by index) and `p50_ms`"

    # single-query latency (submit -> result on host), outside the timed region
    lat = []
    host_idx = None
    sync_query = eng.bound_query_row_topn(topn) if sharded is None else None   # caller-owned result buffers, bound once
    for k in range(total_q, total_q + args.latency_queries):
        t1 = time.perf_counter()
        if sharded is None:
            # one query end to end through the synchronous C-ABI call the C++ Recommender uses
            # (mi355rec_query_row_topn: scan + merge, ids and scores in host memory on return)
            r_idx, r_sc, r_count = sync_query(q_rows[k])
            res = (r_idx[:r_count], r_sc[:r_count])
        else:
            sharded.enqueue_query(q_vecs[k], q_rows[k], topn)   # one query end to end: its own all-gather
            res = sharded.out_keys[:topn].cpu()
        lat.append((time.perf_counter() - t1) * 1e3)
        host_idx = res
    if host_idx is not None and sharded is None:
        host_idx = (host_idx[0].copy(), host_idx[1].copy())   # (views of the bound buffers)
    lat.sort()

    # The same stream of single queries over the fp32 rows (the reference's own 48 B per row,
    # SURVEY.md §8(d)), same protocol — warm-up, K steps with the flush inside the timed region,
    # kernel events — outside the headline's timed region: what the replica buys, measured in the
    # same process, and a complete headline of its own for a reader who prices a query at 48 B/row.
    fp32_rows = None
    if replica and sharded is None and streamed:
        for ln in lanes:
            ln.set_replica(capi.REPLICA_OFF)
        for k in range(args.warmup):
            step(k)
        flush()
        fence()
        dt = sorted(timed_region() for _ in range(3))[1] / args.steps   # (the median of three K-step regions, no events in them)
        eng.set_timing(stride)
        for k in range(args.warmup, total_q):
            step(k)
        flush()
        fence()
        st32 = eng.stats()
        eng.set_timing(False)
        lat32 = []
        for k in range(total_q, total_q + min(200, args.latency_queries)):
            t1 = time.perf_counter()
            eng.query_row_topn(q_rows[k], topn)
            lat32.append((time.perf_counter() - t1) * 1e3)
        lat32.sort()
        alone32 = None
        if n_lanes > 1 and not args.no_single_lane:
            alone32, _ = stream_leg(eng, torch, q_rows, topn, min(args.steps, 100), min(max(args.warmup, 4), 10))
        for ln in lanes:
            ln.set_replica(replica_mode)
        k_ms = float(st32.last_scan_ms)
        fp32_rows = {"kernel": "mi355::scan_kernel<ScanCfg<512,1,6,2>, true, false, 0, true>", "steps": args.steps,
                     "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 5), "value": round(1.0 / dt, 2),
                     "unit": "queries/s", "p50_ms": round(lat32[len(lat32) // 2], 4) if lat32 else None,
                     "roofline": {"bound": roofline_bound((hi - lo) * BYTES_PER_ROW), "algorithmic_bytes_per_launch": (hi - lo) * BYTES_PER_ROW,
                                  "avg_kernel_ms": round(k_ms, 5),
                                  "launches_in_flight": n_lanes,
                                  "achieved": (round((hi - lo) * BYTES_PER_ROW / dt / 1e9, 1) if n_lanes > 1 else
                                               (round((hi - lo) * BYTES_PER_ROW / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None)),
                                  "achieved_per_launch": round((hi - lo) * BYTES_PER_ROW / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None,
                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": (round((hi - lo) * BYTES_PER_ROW / dt / 1e9 / HBM_PEAK_GBPS, 4) if n_lanes > 1 else
                                           (round((hi - lo) * BYTES_PER_ROW / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if k_ms > 0 else None)),
                                  "achieved_note": (f"{n_lanes} lanes: `achieved` = steps x algorithmic bytes / elapsed (value x 48 B x rows, launch gaps included); "
                                                    "ALGORITHMIC bytes: the lanes scan the same rows a fraction of a launch apart, so part of the second lane's rows may come "
                                                    "from the Infinity Cache rather than HBM — FETCH_SIZE counts L2 misses, in front of that cache, and cannot tell "
                                                    "(1.02x the algorithmic bytes per launch either way, profiles/*_pmc_hbm_traffic.json); "
                                                    "the kernel alone, every byte from HBM: `single_lane`" if n_lanes > 1 else None),
                                  **traffic_fields()}}
        if alone32 is not None:
            a_us = alone32["scan_kernel_us"]
            alone32["roofline"] = {"achieved": round((hi - lo) * BYTES_PER_ROW / (a_us * 1e-6) / 1e9, 1) if a_us > 0 else None, "peak": HBM_PEAK_GBPS,
                                   "unit": "GB/s", "frac": round((hi - lo) * BYTES_PER_ROW / (a_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if a_us > 0 else None,
                                   "note": "one handle: algorithmic bytes / the kernel's mean duration with the chip to itself"}
            fp32_rows["single_lane"] = alone32

    # micro-batched throughput, outside the timed region:
    # 12 queries share one pass over the catalogue, seed/final merge launches shared by 36

    def timed(fn, reps, rounds=1):
        """One untimed warm call of `fn`, then `rounds` timed rounds of `reps` calls each (synchronised on both sides): the MEDIAN
        round's time per call; every round's is left in `timed.runs` (a leg that stalled once shows there, not in its figure)."""
        fn()
        fence()
        per_call = []
        for _ in range(rounds):
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            fence()
            if world > 1:
                tt = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            per_call.append(dt / reps)
        timed.runs = per_call
        return sorted(per_call)[len(per_call) // 2]

    # micro-batched throughput, outside the timed region: a dozen queries
    # per call.  With a replica such a call takes the batched matrix-core path (one block of <= 32
    # queries); the exact 12-query pass (mi355::scan_multi_kernel) is timed beside it.
    micro = None
    if topn <= 128:
        def batch_leg(nb, calls, path, streamed_batches=False, over_lanes=False):
            b_rows = np.array(q_rows[:nb], dtype=np.int64)
            rings = [torch.zeros(nb * topn, dtype=torch.int64, device=dev) for _ in range(8)]
            eng.set_batch_path(path)
            state = {"k": 0}
            use = lanes if (over_lanes and streamed_batches and n_lanes > 1) else [eng]
            use_streams = lane_streams if len(use) > 1 else [None]

            def batch_step():
                k = state["k"]
                b_keys = rings[k % 8]
                state["k"] += 1
                if sharded is None:
                    if streamed_batches:
                        use[k % len(use)].enqueue_batch_keys_streamed(q_vecs[:nb], b_rows, topn, b_keys, stream=use_streams[k % len(use)])
                    else:
                        eng.enqueue_batch_keys(q_vecs[:nb], b_rows, topn, b_keys)
                else:
                    sharded.enqueue_batch(q_vecs[:nb], b_rows, topn)

            # the flush of a stream of batches is inside the timed region
            def run():
                for _ in range(calls):
                    batch_step()
                if streamed_batches:
                    for ln, ls in zip(use, use_streams):
                        ln.enqueue_flush(stream=ls)

            # (the warm call is a whole run(): whatever a handle allocates or sizes on its first batch of this shape happens there;
            # three timed rounds, the median reported — VERDICT r5 weak 4: one leg of one driver run took 1 ms per call once)
            dt = timed(run, 1, rounds=3) / calls
            batch_leg.worst = max(timed.runs) / calls
            eng.set_batch_path(capi.BATCH_AUTO)
            return dt, rings[(state["k"] - 1) % 8]

        nb = 12 if sharded is None else 72
        stream_ok = replica and sharded is None
        dt, b_keys = batch_leg(nb, 40 if sharded is None else 6, capi.BATCH_AUTO, stream_ok, over_lanes=True)
        micro = {"queries_per_call": nb, "value": round(nb / dt, 1), "unit": "queries/s",
                 "ms_per_call": round(dt * 1e3, 5), "lanes": n_lanes if stream_ok else 1,
                 "note": ("a STREAM of 12-query batches (mi355rec_enqueue_batch_keys_streamed, the flush inside the timed "
                          "region): one launch per batch = one 24 B/row pass over the fp16 replica with an fp16 matrix-core "
                          "pre-filter (mi355::scan_half_multi_kernel<true>), the previous batch's merges and the next batch's "
                          "sample riding in it"
                          if stream_ok else
                          "one call = the path mi355rec_enqueue_batch_keys picks for this batch size; "
                          + ("single GPU" if sharded is None else "one all-gather per call"))}
        if stream_ok:
            eng.set_timing(1)
            dt12, _ = batch_leg(12, 20, capi.BATCH_AUTO, True)
            st_m = eng.stats()
            eng.set_timing(False)
            k_ms = float(st_m.last_scan_ms)
            alg12 = int(st_m.replica_bytes_per_query)
            micro["roofline"] = {"bound": roofline_bound(alg12), "kernel": "mi355::scan_half_multi_kernel<true> (up to 32 queries per 24 B/row pass)",
                                 "algorithmic_bytes_per_launch": alg12, "avg_kernel_ms": round(k_ms, 5),
                                 "achieved": round(alg12 / (k_ms * 1e-3) / 1e9, 1) if k_ms > 0 else None, "peak": HBM_PEAK_GBPS,
                                 "unit": "GB/s", "frac": round(alg12 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if k_ms > 0 else None}
            for name, nq in (("two_queries", 2), ("thirty_two_queries", 32)):
                d, _ = batch_leg(nq, 40, capi.BATCH_AUTO, True, over_lanes=True)
                micro[name] = {"ms_per_call": round(d * 1e3, 5), "value": round(nq / d, 1), "unit": "queries/s"}
            if n_lanes > 1:   # the same streams of batches through ONE handle (what these figures were until round 4)
                micro["single_lane"] = {}
                for name, nq in (("twelve_queries", 12), ("thirty_two_queries", 32)):
                    d, _ = batch_leg(nq, 40, capi.BATCH_AUTO, True)
                    micro["single_lane"][name] = {"ms_per_call": round(d * 1e3, 5), "value": round(nq / d, 1), "unit": "queries/s",
                                                  "worst_round_ms_per_call": round(batch_leg.worst * 1e3, 5)}
            d, _ = batch_leg(12, 20, capi.BATCH_AUTO, False)
            micro["single_call_12"] = {"ms_per_call": round(d * 1e3, 5), "value": round(12 / d, 1), "unit": "queries/s",
                                       "note": "one batch alone (mi355rec_enqueue_batch_keys): sample launch + pass + merge launch"}
            d, _ = batch_leg(12, 20, capi.BATCH_MFMA, False)
            micro["two_pass_matrix_core_path_12"] = {"ms_per_call": round(d * 1e3, 5), "value": round(12 / d, 1), "unit": "queries/s",
                                                     "note": "the same 12 queries forced through the two-pass batched path (round 2's route)"}
        if sharded is None:
            eng.enqueue_row_keys(q_rows[0], topn, out_keys)
        else:
            sharded.enqueue_query(q_vecs[0], q_rows[0], topn)
        torch.cuda.synchronize()
        a, _ = unpack_keys((out_keys if sharded is None else sharded.out_keys[:topn]).cpu().numpy())
        b, _ = unpack_keys((b_keys[:topn] if sharded is None else sharded.batch_keys[0]).cpu().numpy())
        micro["matches_single_query_path"] = bool(a.tolist() == b.tolist())
        if sharded is None:
            dt, _ = batch_leg(72, 6, capi.BATCH_MULTI)   # two chains of 36 = six exact passes of 12 queries
            micro["exact_multi_query_pass"] = {"queries_per_pass": 12, "queries_per_call": 72, "value": round(72 / dt, 1),
                                               "unit": "queries/s", "ms_per_pass": round(dt / 6 * 1e3, 5),
                                               "kernel": "mi355::scan_multi_kernel (fp32 rows, exact pre-filter)"}

    # the batched path of BASELINE configs[4] (csrc/batched.hip.h): `--batch` queries per call,
    # queries resident in HBM, fp16 matrix-core pre-filter + exact fp32 re-score
    batched = None
    if topn <= 128 and not args.no_batched and (hi - lo) >= 65536:
        bq = args.batch
        bq_rows = np.array([(k * 104729) % n for k in range(bq)], dtype=np.int64)
        bq_dev = torch.from_numpy(bq_rows).to(dev)
        if world == 1:
            q_dev = shard[bq_dev].contiguous()
        q_host = None
        if sharded is not None:
            # every rank needs the query VECTORS: rows live on different ranks, so take them from the seed
            q_host = synthetic_catalogue(n, seed=args.seed, device=dev)[bq_dev].cpu().numpy()
        bq_keys = torch.zeros(bq * topn, dtype=torch.int64, device=dev)

        def bq_step():
            if sharded is None:
                eng.enqueue_batch_keys_dev(q_dev, bq_dev, topn, bq_keys)
            else:
                sharded.enqueue_batch(q_host, bq_rows, topn)

        def measure_batched(engine, step_fn, rows_local):
            step_fn()
            fence()
            dt = timed(step_fn, 10, rounds=3)   # (the median of three rounds of ten calls; no HIP events inside them)
            engine.set_timing(1)                # ... the passes' mean duration from an untimed evented round
            for _ in range(4):
                step_fn()
            fence()
            pass_ms = float(engine.stats().last_pass_ms)
            engine.set_timing(False)
            diag = engine.batched_last_counters()
            pairs = engine.batched_pass2_pairs()
            flops = FLOP_PER_PAIR * float(rows_local) * bq
            # Matrix-core work really issued: v_mfma_f32_32x32x16_f16 = 2*32*32*16 flop, two of them per (64-row tile,
            # block of 32 queries) pair.  Pass 1 looks at every 4th tile; pass 2 at the pairs the library counted
            # (mi355rec_batched_pass2_pairs: all of them without tile skipping, ~78 % with it).
            tiles64 = (rows_local + 63) // 64
            blocks = 1
            while blocks * 32 < min(bq, 1024):
                blocks *= 2
            chunks = (bq + 1023) // 1024
            per_pair = 2 * 2.0 * 32 * 32 * 16
            pass1_pairs = ((tiles64 + 3) // 4) * blocks
            issued = per_pair * chunks * (pass1_pairs + pairs["pairs_done"])
            unskipped = per_pair * chunks * (pass1_pairs + pairs["pairs_total"])
            return {
                "queries_per_call": bq, "value": round(bq / dt, 1),
                "unit": "queries/s", "ms_per_call": round(dt * 1e3, 4), "rows_per_gpu": rows_local,
                "roofline": {
                    "bound": "mfma", "kernel": "mi355::bq_pass_kernel (pass 1 + pass 2), v_mfma_f32_32x32x16_f16",
                    "issued_mfma_flops_per_call_per_gpu": issued,
                    "achieved": round(issued / dt / 1e12, 1), "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(issued / dt / 1e12 / FP16_MFMA_PEAK_TFLOPS, 3),
                    "pass2_pairs_done": pairs["pairs_done"], "pass2_pairs_total": pairs["pairs_total"],
                    "pass2_pairs_skipped_frac": round(1.0 - pairs["pairs_done"] / max(1, pairs["pairs_total"]), 4),
                    "frac_if_nothing_were_skipped": round(unskipped / dt / 1e12 / FP16_MFMA_PEAK_TFLOPS, 3),
                    "avg_pass_kernel_ms": round(pass_ms, 4),
                    "binding_unit": "VALU issue (8 v_max3_i32 per MFMA = 1024 outputs), the hit path (the blocks that hold a "
                                    "candidate cost ~3x a block without one) and chip power; 62 of the 81 cycles a SIMD spends per MFMA are vector issue (profiles/r05_pmc_batched_pass.json, DESIGN.md §5.5)",
                    "note": "the pre-filter runs on the fp16 matrix cores, so its roofline is the dense fp16 MFMA peak and `achieved` "
                            "counts the MFMA flops really issued; pass 2 SKIPS the (tile, block) pairs that the maxima pass 1 left "
                            "behind rule out, which lowers this fraction while the call gets faster — "
                            "`frac_if_nothing_were_skipped` prices the same call at the MFMAs a skip-less pass 2 would issue",
                    "survey_priced": {"algorithmic_flops_per_call_per_gpu": flops, "tflops": round(flops / dt / 1e12, 1),
                                      "fp32_peak": FP32_PEAK_TFLOPS, "ratio": round(flops / dt / 1e12 / FP32_PEAK_TFLOPS, 3),
                                      "note": "SURVEY.md §8(d)'s 24 flop per (row, query) pair against the fp32 peak: NOT a "
                                              "roofline fraction (> 1 because the exact fp32 chain only touches the candidates)"},
                },
                "candidates_per_query": round(diag["candidates_total"] / max(1, min(bq, 1024) - diag["queued_queries"]), 1),
                "queued_to_exact_scan": diag["queued_queries"], "special_rows": diag["special_rows"],
            }

        def batched_over_lanes(engine, q_d, rows_d, ref_keys, reps=16, lane=None):
            """The same 1024-query batches dealt alternately over TWO LANES of the handle, each on its own stream: a batch's
            passes are VALU-bound with the chip full, but its small latency-bound launches (prepare, select, finalize, the
            queue's: ~60 us of 600) run beside the other lane's passes.  Aggregate figures; one batch's own latency is
            `ms_per_call` of the parent object."""
            ln = lane if lane is not None else engine.lane()
            pair, pst = [engine, ln], [engine.own_stream(), ln.own_stream()]
            lk = [[torch.zeros(bq * topn, dtype=torch.int64, device=dev) for _ in range(2)] for _ in pair]
            torch.cuda.synchronize()

            def run(count):
                for k in range(count):
                    pair[k & 1].enqueue_batch_keys_dev(q_d, rows_d, topn, lk[k & 1][(k >> 1) & 1], stream=pst[k & 1])
                torch.cuda.synchronize()
            run(4)
            t1 = time.perf_counter()
            run(reps)
            d2 = (time.perf_counter() - t1) / reps
            same = bool(torch.equal(lk[1][0], ref_keys)) and bool(torch.equal(lk[0][0], ref_keys))
            status = ln.lane_status()
            if lane is None:
                ln.close()
            return {"lanes": 2, "ms_per_call": round(d2 * 1e3, 4), "value": round(bq / d2, 1), "unit": "queries/s",
                    "matches_one_handle": same, "lane_stream": status,
                    "note": "aggregate over two lanes of the handle (mi355rec_create_lane), batches dealt alternately, each lane on its own stream"}

        batched = measure_batched(eng, bq_step, hi - lo)
        if sharded is None and world == 1 and n_lanes > 1:
            batched["two_lanes"] = batched_over_lanes(eng, q_dev, bq_dev, bq_keys, lane=lanes[1])
        # the first and last query of the batch against the single-query path
        ok = True
        for k in (0, bq - 1):
            if sharded is None:
                eng.enqueue_row_keys(int(bq_rows[k]), topn, out_keys)
                torch.cuda.synchronize()
                a = out_keys.cpu().numpy()
                b = bq_keys[k * topn:(k + 1) * topn].cpu().numpy()
            else:
                sharded.enqueue_query(q_host[k], int(bq_rows[k]), topn)
                torch.cuda.synchronize()
                a = sharded.out_keys[:topn].cpu().numpy()
                b = sharded.batch_keys[k].cpu().numpy()
            ok = ok and bool(np.array_equal(a, b))
        batched["matches_single_query_path"] = ok

        # BASELINE configs[4] as ONE of its eight GPUs sees it: a 12.5 M-row shard of the 100 M catalogue, 1024 queries
        # per call (the driver-run record carried the 10 M figure only).  Its own catalogue and handle,
        # built after the headline's legs; skipped for other sizes / ranks.
        if world == 1 and n == 10_000_000 and topn == 100 and not args.no_c5_shard:
            c5_rows = 12_500_000
            c5 = synthetic_catalogue(c5_rows, seed=args.seed + 5, device=dev)
            with CosineEngine(c5) as eng5:
                q5_rows = torch.from_numpy(np.array([(k * 104729) % c5_rows for k in range(bq)], dtype=np.int64)).to(dev)
                q5 = c5[q5_rows].contiguous()
                k5 = torch.zeros(bq * topn, dtype=torch.int64, device=dev)
                batched["configs4_shard"] = measure_batched(eng5, lambda: eng5.enqueue_batch_keys_dev(q5, q5_rows, topn, k5), c5_rows)
                batched["configs4_shard"]["workload"] = ("one 12.5 M-row shard of BASELINE configs[4] (100 M rows over 8 GPUs), 1024 "
                                                         "device-resident queries per call, top-100")
                if n_lanes > 1:
                    batched["configs4_shard"]["two_lanes"] = batched_over_lanes(eng5, q5, q5_rows, k5)
                # two of its queries against the single-query path of the same handle
                ok5 = True
                one = torch.zeros(topn, dtype=torch.int64, device=dev)
                for k in (0, bq - 1):
                    eng5.enqueue_row_keys(int(q5_rows[k].item()), topn, one)
                    torch.cuda.synchronize()
                    ok5 = ok5 and bool(torch.equal(one, k5[k * topn:(k + 1) * topn]))
                batched["configs4_shard"]["matches_single_query_path"] = ok5
            del c5, q5, k5
            torch.cuda.empty_cache()

    # The read ceiling of EVERY buffer a timed kernel streams, taken over that very buffer in the state the timed stream
    # leaves it in (a plain read-only kernel, csrc/kernels.hip.h stream_probe_kernel): the 480 MB fp32 matrix comes from
    # HBM, the 120 MB 8-bit replica sits in the 256 MiB Infinity Cache between two passes — a kernel's rate is only ever
    # held against the probe of ITS buffer.
    probe_gbps = probe_q8 = probe_fp16 = None
    if not args.no_kernel_events:
        rows_l = hi - lo
        probe_gbps = probe_gbps_of(eng, torch, capi.PROBE_FP32_ROWS, rows_l * BYTES_PER_ROW, dev)
        if replica:
            probe_q8 = probe_gbps_of(eng, torch, capi.PROBE_Q8_REPLICA, (rows_l + 3) // 4 * 48, dev)
            probe_fp16 = probe_gbps_of(eng, torch, capi.PROBE_FP16_REPLICA, (rows_l + 1) // 2 * 48, dev)
    probe_lanes = None   # the buffer the headline kernel streams, read by every lane at once
    if not args.no_kernel_events and n_lanes > 1:
        rows_l = hi - lo
        if replica:
            probe_lanes = probe_concurrent_gbps(lanes, lane_streams, torch, capi.PROBE_Q8_REPLICA, (rows_l + 3) // 4 * 48, dev)
        else:
            probe_lanes = probe_concurrent_gbps(lanes, lane_streams, torch, capi.PROBE_FP32_ROWS, rows_l * BYTES_PER_ROW, dev)

    if rank == 0:
        qps = args.steps / elapsed
        scan_ms = float(st.last_scan_ms) if st is not None else 0.0
        row_bytes, alg_bytes, kernel_fmt, kernel_pmc, dtype_label, replica_label = replica_desc(st_headline)
        per_launch = (alg_bytes / (scan_ms * 1e-3) / 1e9) if scan_ms > 0 else None
        # With lanes the launches of this kernel OVERLAP (by how much varies along the stream), so no single launch's duration says
        # what the chip sustains on it: `achieved` is then the sustained rate of the timed region itself — every step's algorithmic
        # bytes / the elapsed time `value` is made of, launch gaps included — held against the kernel-level rate of as many plain
        # read streams at once (no gaps in it: the fraction is on the safe side).  One handle: bytes / the kernel's mean duration.
        achieved = (qps * alg_bytes / 1e9) if n_lanes > 1 else per_launch
        # a pass only finds its bytes in the 256 MiB Infinity Cache if the whole buffer survives one
        # full pass of itself plus the fp32 fetches: half the cache is the most that can be hoped for
        cache_resident = alg_bytes <= 128 * 2**20
        own_probe = {12: probe_q8, 24: probe_fp16}.get(row_bytes, probe_gbps)   # the plain read of the buffer THIS kernel streams
        if n_lanes > 1 and row_bytes == 24:
            probe_lanes = None   # (measured over the 8-bit replica or the fp32 rows only)
        if n_lanes > 1 and probe_lanes and cache_resident:
            peak, peak_source = probe_lanes, f"measured: {n_lanes} plain read streams at once over the same cache-resident buffer (probes.note)"
        elif cache_resident and own_probe:
            peak, peak_source = own_probe, "measured: one plain read stream over the same cache-resident buffer (probes.note)"
        else:
            peak, peak_source = HBM_PEAK_GBPS, "MI355X_MICROARCH.md: HBM3E spec peak"
        if replica:
            targs = "true, true" if streamed else ("true, false" if sharded is None else "false, true")
            kernel_name = kernel_fmt.format(targs=targs) + " over the " + replica_label + ")"
        else:
            kernel_name = ("mi355::scan_kernel<ScanCfg<512,1,6,2>, true, false, 0, true>" if streamed else
                           ("mi355::scan_kernel<ScanCfg<512,1,6,2>, true, false>" if world == 1 else
                            "mi355::scan_kernel<ScanCfg<512,1,6,2>, false, false, 0, true>"))
        # SURVEY.md §8(d) prices a query at 48 B/row: that is the fp32 scan's roofline, measured in this run — the kernel ALONE
        # (one handle, every byte from HBM) when the lanes leg ran, else the fp32 stream's own events
        survey = None
        if fp32_rows is not None:
            alone32 = fp32_rows.get("single_lane")
            k_us = alone32["scan_kernel_us"] if alone32 else fp32_rows["roofline"]["avg_kernel_ms"] * 1e3
            if k_us and k_us > 0:
                g = (hi - lo) * BYTES_PER_ROW / (k_us * 1e-6) / 1e9
                survey = {"kernel_us": round(k_us, 2), "achieved": round(g, 1), "frac": round(g / HBM_PEAK_GBPS, 4)}
        elif not replica and per_launch:
            survey = {"kernel_us": round(scan_ms * 1e3, 2), "achieved": round(per_launch, 1), "frac": round(per_launch / HBM_PEAK_GBPS, 4)}
        rescored = (round((rc_after["rescored_rows"] - rc_before["rescored_rows"]) / max(1, rc_after["scans"] - rc_before["scans"]), 1)
                    if replica else None)
        value_runs = [round(args.steps / t, 1) for t in run_times]
        line = {
            "metric": metric_label(n, topn),
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": dtype_label, "data": "synthetic",
            "value_runs": value_runs,
            "value_note": f"median of {len(run_times)} back-to-back {args.steps}-step regions, each between barrier + synchronize pairs",
            "config": {
                "workload": f"{n} synthetic tracks x 12 fp32 features, top-{topn}, "
                            + (f"1 MI355X ({workload_label(n, topn)})" if world == 1 else
                               f"row-sharded across {world} MI355X, one process per GPU, one all-gather of {topn} keys/rank "
                               + ("(BASELINE configs[3])" if (n == 10_000_000 and topn == 100) else "")),
                "rows": n, "topn": topn, "rows_per_gpu": hi - lo, "queries_per_step": 1,
                "lanes": n_lanes if sharded is None else rank_lanes,
                "lane_streams": ([ln.lane_status() for ln in lanes[1:]] if (sharded is None and n_lanes > 1) else None),
                "lanes_note": (f"the stream is dealt over {n_lanes} lanes of one handle (mi355rec_create_lane: same rows and replicas, own "
                               "stream state and HIP stream); `single_lane` is the same stream through one handle" if n_lanes > 1 else None),
                "merge": ("inside the next query's scan launch (streamed), last one flushed in the timed region"
                          if streamed else ("own launch per query" if sharded is None else
                                            f"local merge streamed; one all-gather + one batched merge per {args.window} queries")),
                "seed": args.seed, "generator": "torch.rand(seed) uniform[0,1) on device",
                "host_wait": ("polled completion signals (HSA_ENABLE_INTERRUPT=0, set by bench.py)" if os.environ.get("HSA_ENABLE_INTERRUPT") == "0"
                              else ("the runtime's default (interrupt-driven)" if os.environ.get("HSA_ENABLE_INTERRUPT") is None
                                    else f"HSA_ENABLE_INTERRUPT={os.environ.get('HSA_ENABLE_INTERRUPT')} (the caller's)")),
            },
            "p50_ms": round(lat[len(lat) // 2], 4) if lat else None,
            "p99_ms": round(lat[min(len(lat) - 1, int(len(lat) * 0.99))], 4) if lat else None,
            # <= 20 SCALAR keys, short strings (the driver's record keeps a bounded number of keys and characters): everything
            # longer lives in the top-level `roofline_notes` / `probes` objects
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": round(achieved, 1) if achieved else None, "peak": round(peak, 1), "unit": "GB/s",
                "frac": round(achieved / peak, 4) if achieved else None,
                "survey_frac": survey["frac"] if survey else None,
                "survey_achieved": survey["achieved"] if survey else None,
                "survey_kernel_us": survey["kernel_us"] if survey else None,
                "traffic": None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "bytes_per_row": row_bytes,
                "avg_kernel_ms": round(scan_ms, 5) if scan_ms else None,
                "launches_in_flight": n_lanes,
                "achieved_per_launch": round(per_launch, 1) if per_launch else None,
                "frac_of_hbm_peak": round(achieved / HBM_PEAK_GBPS, 4) if achieved else None,
                "cache_resident": bool(cache_resident),
                "rescored_rows_per_query": rescored,
                "peak_source": peak_source,
            },
            "roofline_notes": {
                "frac": ("bytes this kernel streams (" + replica_label + ")): " if replica else "") +
                        (f"{n_lanes} lanes, launches overlap: achieved = value x algorithmic bytes (sustained, launch gaps included); "
                         "peak = kernel-level rate of as many plain read streams at once (no gaps)" if n_lanes > 1 else
                         "achieved = algorithmic bytes / avg_kernel_ms"),
                "survey": "SURVEY.md §8(d) prices a query at 48 B/row = the fp32 scan (mi355::scan_kernel) ALONE on one handle, every byte "
                          "from HBM: survey_achieved = rows x 48 B / survey_kernel_us, survey_frac = that / 8000 GB/s; same run, HIP events; "
                          "the `fp32_rows` object is that stream's own headline",
                "avg_kernel_ms": "HIP events on the launching stream around every stride-th scan launch of lane 0, in an UNTIMED pass of the same "
                                 "stream right after the timed regions; with lanes = one launch's duration while the other lane's is in flight "
                                 "(what a rocprofv3 kernel trace of this command averages); the kernel with the chip to itself: `single_lane`",
                "cache": ("the buffer this kernel streams fits the 256 MiB Infinity Cache with room to spare and stays there between "
                          "queries: `peak` is the measured plain read of that same buffer, `frac_of_hbm_peak` the same rate against 8000 GB/s"
                          if cache_resident else None),
                "prefilter_margin": (("per query: l1(Q)/(254 S) + sqrt(12)/(2 S) + 3e-5 (<= 0.0137), integer dot" if row_bytes == 12
                                      else round(float(st.replica_margin_single), 6)) if (replica and st is not None) else None),
                "merge_kernel_ms": round(float(st.last_merge_ms), 5) if st is not None else None,
                **traffic_fields(),
            },
            # plain reads of each buffer, GB/s (a kernel is only ever compared with the probe of ITS buffer)
            "probes": {"fp32_rows": round(probe_gbps, 1) if probe_gbps else None,
                       "q8_replica": round(probe_q8, 1) if probe_q8 else None,
                       "fp16_replica": round(probe_fp16, 1) if probe_fp16 else None,
                       "own_buffer_all_lanes_at_once": round(probe_lanes, 1) if probe_lanes else None,
                       "note": "csrc/kernels.hip.h stream_probe_kernel, HIP events: one stream = 20 launches; all lanes at once = lanes x bytes "
                               "/ the mean duration of a probe launch while the other lane's is in flight (30 launches per lane); "
                               "MI355X_MICROARCH.md measures 7.4-8.6 TB/s for Infinity-Cache-served reads, 6.29 TB/s for an HBM copy"},
        }
        if group is not None:
            # (north_star: "a single RCCL all-gather over xGMI to merge per-shard top-N candidates" — under the driver's
            # torch.distributed.run launch that is torch.distributed's all_gather_into_tensor on backend nccl = RCCL)
            line["transport"] = {"value_is": "rccl", "rccl": {"via": "torch.distributed all_gather_into_tensor, one per window of "
                                                                     f"{args.window} queries, {topn} keys per query and rank",
                                                              "backend": group["backend"], "ranks": group["ranks"],
                                                              "ranks_counted": group["ranks_counted"], "value": round(qps, 2), "unit": "queries/s"},
                                 "peer": None, "peer_note": "peer stores are a transport of the ONE-process node handle (bench.py --gpus N launched plainly)"}
            line["group"] = group
            line["peer_access"] = peer_access_matrix(torch, list(range(torch.cuda.device_count())))
        if single_lane is not None:
            k_us = single_lane["scan_kernel_us"]
            alone = (alg_bytes / (k_us * 1e-6) / 1e9) if k_us > 0 else None
            alone_probe = {12: probe_q8, 24: probe_fp16}.get(row_bytes, probe_gbps)
            single_lane["roofline"] = {"achieved": round(alone, 1) if alone else None, "unit": "GB/s",
                                       "peak": round(alone_probe, 1) if (alone_probe and cache_resident) else HBM_PEAK_GBPS,
                                       "frac": (round(alone / (alone_probe if (alone_probe and cache_resident) else HBM_PEAK_GBPS), 4) if alone else None),
                                       "note": "one handle, one chain of launches: algorithmic bytes / the kernel's mean duration with the chip "
                                               "to itself, against the ONE plain read stream over the same buffer"}
            line["single_lane"] = single_lane
        if by_value is not None:
            line["by_value"] = by_value
        if fp32_rows is not None:
            line["fp32_rows"] = fp32_rows
            tgt = (fp32_rows.get("single_lane") or {}).get("roofline") or fp32_rows["roofline"]   # (ONE kernel's rate against ONE plain stream's)
            if probe_gbps and tgt.get("achieved"):
                tgt["stream_probe_gbps"] = round(probe_gbps, 1)
                tgt["frac_of_stream_probe"] = round(tgt["achieved"] / probe_gbps, 4)
        if micro is not None:
            if micro.get("roofline") and probe_fp16 and micro["roofline"].get("achieved"):
                mr = micro["roofline"]
                mr["stream_probe_gbps"] = round(probe_fp16, 1)
                mr["frac_of_own_buffer_probe"] = round(mr["achieved"] / probe_fp16, 4)
                mr["peak_note"] = ("`frac` is against the HBM spec peak: the 240 MB fp16 replica does not survive a pass in the 256 MiB "
                                   "Infinity Cache; `frac_of_own_buffer_probe` is against the plain read of that same buffer")
            line["microbatch"] = micro
        if batched is not None:
            line["batched"] = batched
        if world == 1 and not args.no_clustered and (hi - lo) >= 4_000_000:
            shapes = ([(3000, 0.03, False), (3000, 0.03, True), (300, 0.01, False), (300, 0.01, True)]
                      if args.catalogue == "clustered-contiguous" else [(3000, 0.03, True), (300, 0.01, True)])   # (both shapes in every line: VERDICT r5 item 6)
            shapes = [(max(2, c * n // 10_000_000), sp, r) for c, sp, r in shapes]   # (clusters of ~3300 / ~33000 rows whatever --rows is)
            from oracle import oracle as _checker   # (handed to the leg as its checker)
            line["clustered"] = clustered_object(args, torch, np, dev, shapes, _checker)
        if world == 1 and not args.no_config0:
            try:
                from oracle import oracle as _checker
                line["config0"] = config0_object(torch, np, _checker)
            except Exception as e:   # (a box without g++ for the shim, ...): the headline stands without it
                line["config0"] = {"unavailable": repr(e)[:300]}
        if feats_host is not None:
            line["cpu_baseline"] = cpu_baseline(feats_host, topn, q_rows[:64])
            # several queries of the run, checked against the oracle (checker only)
            from oracle import oracle
            ok = True
            checked = 0
            for k in (0, total_q // 3, total_q // 2, total_q + args.latency_queries - 1):
                eng.enqueue_row_keys(q_rows[k], topn, out_keys)
                torch.cuda.synchronize()
                rows_got, sc_got = unpack_keys(out_keys.cpu().numpy())
                want = oracle.scores(feats_host, feats_host[q_rows[k]], threads=0)
                ci, cs = oracle.topn_canonical(want, q_rows[k], topn)
                ok = ok and rows_got.tolist() == ci.tolist() and bool(np.array_equal(sc_got, cs + np.float32(0)))
                checked += 1
            for row, keys_t in timed_tail:      # results of the TIMED stream (merge riding in the next scan launch)
                rows_got, sc_got = unpack_keys(keys_t.cpu().numpy())
                want = oracle.scores(feats_host, feats_host[row], threads=0)
                ci, cs = oracle.topn_canonical(want, row, topn)
                ok = ok and rows_got.tolist() == ci.tolist() and bool(np.array_equal(sc_got, cs + np.float32(0)))
                checked += 1
            if host_idx is not None and sharded is None:   # the last synchronous latency query (ids + scores on the host)
                row = q_rows[total_q + args.latency_queries - 1]
                want = oracle.scores(feats_host, feats_host[row], threads=0)
                ci, cs = oracle.topn_canonical(want, row, topn)
                ok = ok and host_idx[0].tolist() == ci.tolist() and bool(np.array_equal(host_idx[1], cs + np.float32(0)))
                checked += 1
            if by_value_last is not None:                 # the last query of the by-value stream: nothing excluded
                vec = v_vecs[-1 - by_value_last[0]]
                rows_got, sc_got = unpack_keys(by_value_last[1].cpu().numpy())
                want = oracle.scores(feats_host, np.ascontiguousarray(vec), threads=0)
                ci, cs = oracle.topn_canonical(want, -1, topn)
                ok = ok and rows_got.tolist() == ci.tolist() and bool(np.array_equal(sc_got, cs + np.float32(0)))
                checked += 1
            if batched is not None:
                for k in (1, args.batch // 2):
                    row = int((k * 104729) % n)
                    rows_got, sc_got = unpack_keys(bq_keys[k * topn:(k + 1) * topn].cpu().numpy())
                    want = oracle.scores(feats_host, feats_host[row], threads=0)
                    ci, cs = oracle.topn_canonical(want, row, topn)
                    ok = ok and rows_got.tolist() == ci.tolist() and bool(np.array_equal(sc_got, cs + np.float32(0)))
                    checked += 1
            line["verified_against_oracle"] = bool(ok)
            line["verified_queries"] = checked
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    eng.close()
    if world > 1 or force_sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
