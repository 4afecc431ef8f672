"""The legs of bench.py that are not the timed region: probes of the plain read rate of a handle's buffers, a stream of single
queries through one handle, the `clustered` object (sorted catalogues), the `config0` object (BASELINE configs[0] from CSV) and the
`--preflight` report of the node handle.  Moved out of bench.py in round 6; no behaviour change.

Nothing here imports `oracle/`: the legs that check results take the checker module as an argument from bench.py (the oracle is
test infrastructure — bench.py's verification and `cpu_baseline` legs are its only users outside tests/)."""
from __future__ import annotations

import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_ROW = 48      # algorithmic bytes per catalogue row per query
FP32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: fp32 vector = fp32 matrix peak (configs[4]'s roofline)
FP16_MFMA_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (never the 2:1-sparsity figure)
FLOP_PER_PAIR = 24          # 12 mul + 12 add per (row, query) pair of the batched path


def traffic_fields():
    """`roofline.traffic` is null in the line: HBM bytes come from PMC counters (separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes over this same command), which the timed process cannot read about itself — earlier rounds copied
    the figure of a committed profile into the line, which is a citation, not a measurement.  The profiles are the
    evidence."""
    files = sorted((ROOT / "profiles").glob("*_pmc_hbm_traffic.json"))
    return {"traffic": None,
            "traffic_note": "not measurable from inside the run; per-launch HBM bytes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                            "FETCH doubled per the guide's gfx950 correction) are in profiles/"
                            + (files[-1].name if files else "*_pmc_hbm_traffic.json")}


def roofline_bound(alg_bytes: int) -> str:
    """A buffer that a pass can still find in the 256 MiB Infinity Cache when it comes round again is not an HBM
    stream: its rate is bound by the cache / fabric, and FETCH_SIZE counts cache hits as well (guide, HBM section)."""
    return "infinity-cache" if alg_bytes <= 128 * 2**20 else "hbm"


def host_threads(omp_max: int):
    """Threads the CPU baseline may really use, and why: the affinity mask, the cgroup quota and BENCH_CPU_THREADS
    (default 16 = the GPU box's CPU share per GPU) bound it, not the socket's core count."""
    info = {"nproc": os.cpu_count(), "omp_max_threads": omp_max}
    n = omp_max
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
        n = min(n, info["affinity"])
    except Exception:
        info["affinity"] = None
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        info["cgroup_cpus"] = None if quota == "max" else round(int(quota) / int(period), 2)
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        info["cgroup_cpus"] = None
    info["thread_cap"] = int(os.environ.get("BENCH_CPU_THREADS", "16"))
    info["thread_cap_source"] = "BENCH_CPU_THREADS" if "BENCH_CPU_THREADS" in os.environ else "default 16: a GPU box's CPU share per GPU"
    n = max(1, min(n, info["thread_cap"]))
    return n, info


def probe_gbps_of(eng, torch, which, n_bytes, dev):
    """The plain read-only stream over one buffer of the handle (csrc/kernels.hip.h, stream_probe_kernel), 20 timed
    launches behind 3 untimed ones: GB/s, or None when the handle has no such buffer."""
    from spotify_recommender_amd import capi
    sink = torch.zeros(4096, dtype=torch.int32, device=dev)
    try:
        for _ in range(3):
            eng.enqueue_stream_probe(sink, which=which)
    except capi.Mi355Error:
        return None
    torch.cuda.synchronize()
    eng.set_timing(True)
    for _ in range(20):
        eng.enqueue_stream_probe(sink, which=which)
    torch.cuda.synchronize()
    ms = float(eng.stats().last_scan_ms)
    eng.set_timing(False)
    return n_bytes / (ms * 1e-3) / 1e9 if ms > 0 else None


def probe_concurrent_gbps(lanes, lane_streams, torch, which, n_bytes, dev):
    """The same plain read-only stream from EVERY lane at once, each on its own HIP stream, over the one buffer they share: the
    aggregate rate while they overlap = lanes x bytes / the mean duration of a probe launch with the others' in flight (lane 0's
    HIP events over 30 launches) — a kernel-level ceiling without launch gaps, so a sustained rate held against it is on the safe
    side.  (A wall-clock version of this probe is bounded by how fast Python can enqueue 19 us kernels, not by the chip: 6.4 TB/s
    where the events say 8.4 and MI355X_MICROARCH.md measures 7.4-8.6 TB/s for Infinity-Cache-served reads.)  None when the handle
    has no such buffer."""
    from spotify_recommender_amd import capi
    sinks = [torch.zeros(4096, dtype=torch.int32, device=dev) for _ in lanes]
    torch.cuda.synchronize()
    try:
        for _ in range(3):
            for ln, ls, sk in zip(lanes, lane_streams, sinks):
                ln.enqueue_stream_probe(sk, stream=ls, which=which)
    except capi.Mi355Error:
        return None
    torch.cuda.synchronize()
    lanes[0].set_timing(True)
    for _ in range(30):
        for ln, ls, sk in zip(lanes, lane_streams, sinks):
            ln.enqueue_stream_probe(sk, stream=ls, which=which)
    torch.cuda.synchronize()
    ms = float(lanes[0].stats().last_scan_ms)
    lanes[0].set_timing(False)
    return len(lanes) * n_bytes / (ms * 1e-3) / 1e9 if ms > 0 else None


def stream_leg(eng, torch, rows, topn, steps, warmup, vectors=None):
    """A stream of single queries through ONE handle (merge riding in the next launch, flush inside the timed region, no HIP
    events in it): us per step and rows sent to the exact chain per query; then the scan kernel's mean duration from an UNTIMED
    pass of the same stream with the library's HIP events around every launch; the last result of the timed pass.
    `vectors` (len(rows) x 12 floats, host): the queries are passed BY VALUE with nothing excluded
    (mi355rec_enqueue_query_keys_streamed(q, -1)) instead of by row — the last result then carries exclude = -1."""
    ring = [torch.zeros(topn, dtype=torch.int64, device="cuda") for _ in range(4)]
    by_value = None
    if vectors is not None:
        import ctypes
        from spotify_recommender_amd import capi
        fn, h = eng._lib.mi355rec_enqueue_query_keys_streamed, eng._h   # (bound once: the conversions of the generic wrapper are ~10 us per call)
        ptrs = [ctypes.c_void_p(t.data_ptr()) for t in ring]
        qptr = [vectors[i].ctypes.data_as(ctypes.c_void_p) for i in range(len(vectors))]
        stream0 = eng._stream_ptr(None)

        def by_value(i, k):
            capi.check(fn(h, qptr[i], -1, topn, ptrs[k % 4], stream0), h)

    def run(k0, count):
        for k in range(count):
            if by_value:
                by_value((k0 + k) % len(rows), k)
            else:
                eng.enqueue_row_keys_streamed(int(rows[(k0 + k) % len(rows)]), topn, ring[k % 4])
        eng.enqueue_flush()
        torch.cuda.synchronize()

    run(0, warmup)
    c0 = eng.replica_counters()
    t0 = time.perf_counter()
    run(warmup, steps)
    dt = (time.perf_counter() - t0) / steps
    c1 = eng.replica_counters()
    last = (int(rows[(warmup + steps - 1) % len(rows)]) if vectors is None else -1 - ((warmup + steps - 1) % len(rows)), ring[(steps - 1) % 4].clone())
    eng.set_timing(1)
    run(warmup, min(steps, 64))
    k_ms = float(eng.stats().last_scan_ms)
    eng.set_timing(False)
    return {"us_per_step": round(dt * 1e6, 2), "queries_per_s": round(1.0 / dt, 1), "scan_kernel_us": round(k_ms * 1e3, 2),
            "rows_to_exact_chain_per_query": round((c1["rescored_rows"] - c0["rescored_rows"]) / steps, 1)}, last


def clustered_object(args, torch, np, dev, shapes, oracle):
    """Every route over catalogues whose similar rows lie NEXT TO EACH OTHER (a CSV grouped by genre: DataManager.cpp:244-250,299)
    — where an evenly spaced sample misses the query's own cluster and the launch-wide bound has to come from the excluded
    row's neighbourhood (csrc/handoff.hip.h).  Same sizes as the headline (rows x top-N), its own catalogue and handle per
    shape; a few results of each are checked against the oracle (`oracle`: the checker module, handed in by bench.py —
    nothing in this package imports it)."""
    import ctypes
    from spotify_recommender_amd import CosineEngine, capi
    from spotify_recommender_amd.engine import unpack_keys
    from spotify_recommender_amd.synth import clustered_catalogue
    n, topn = args.rows, args.topn
    out = {"rows": n, "topn": topn, "generator": "spotify_recommender_amd.synth.clustered_catalogue(contiguous=True), seed 777",
           "queries": "catalogue rows (k * 104729) mod N: spread over the whole shard", "shapes": []}
    q_rows = [(k * 104729) % n for k in range(2048)]
    for clusters, spread, ramp in shapes:
        t = clustered_catalogue(n, spread, clusters=clusters, contiguous=True, ramp=ramp, device=dev)
        shape = {"clusters": clusters, "rows_per_cluster": n // clusters, "spread": spread, "genre_ramp": ramp}
        checks = []
        with CosineEngine(t) as eng:
            eng.set_replica(capi.REPLICA_OFF)
            shape["fp32_rows_stream"], last = stream_leg(eng, torch, q_rows, topn, 100, 10)
            checks.append(last)
            eng.set_replica(capi.REPLICA_AUTO)
            shape["replica_q8_stream"], last = stream_leg(eng, torch, q_rows, topn, 200, 20)
            checks.append(last)
            # the same rows passed BY VALUE with nothing excluded (a new track): the launch-wide bound then comes from the
            # neighbourhood of the query's ANCHOR (csrc/handoff.hip.h) — VERDICT r5 "missing" 3
            v_rows = q_rows[:256]
            v_vecs = np.ascontiguousarray(t[torch.tensor(v_rows, device=dev)].cpu().numpy())
            eng.set_replica(capi.REPLICA_OFF)
            shape["fp32_rows_stream_by_value"], _ = stream_leg(eng, torch, v_rows, topn, 100, 10, vectors=v_vecs)
            eng.set_replica(capi.REPLICA_AUTO)
            shape["replica_q8_stream_by_value"], last_v = stream_leg(eng, torch, v_rows, topn, 200, 20, vectors=v_vecs)
            checks.append((("by value", v_vecs[-1 - last_v[0]]), last_v[1]))
            # the same stream over two lanes of the handle, each on its own stream
            lane = eng.lane()
            pair = [eng, lane]
            calls = [e.bound_enqueue_row_keys_streamed(topn, e.own_stream()) for e in pair]
            lrings = [[torch.zeros(topn, dtype=torch.int64, device=dev) for _ in range(4)] for _ in pair]
            lptrs = [[ctypes.c_void_p(tt.data_ptr()) for tt in rs] for rs in lrings]
            torch.cuda.synchronize()

            def lanes_run(k0, k1):
                for k in range(k0, k1):
                    calls[k & 1](int(q_rows[k % len(q_rows)]), lptrs[k & 1][(k >> 1) & 3])
                for e in pair:
                    e.enqueue_flush(stream=e.own_stream())
                torch.cuda.synchronize()
            lanes_run(0, 40)
            t1 = time.perf_counter()
            lanes_run(40, 440)
            dt2 = (time.perf_counter() - t1) / 400
            checks.append((int(q_rows[439 % len(q_rows)]), lrings[1][(439 >> 1) & 3].clone()))
            lane.close()
            shape["replica_q8_stream_two_lanes"] = {"us_per_step": round(dt2 * 1e6, 2), "queries_per_s": round(1.0 / dt2, 1)}
            lat = []
            for k in range(100):
                t1 = time.perf_counter()
                res = eng.query_row_topn(q_rows[300 + k], topn)
                lat.append((time.perf_counter() - t1) * 1e6)
            lat.sort()
            shape["one_query_alone_p50_us"] = round(lat[len(lat) // 2], 1)
            checks.append((q_rows[399], res))
            lat = []
            for k in range(100):
                t1 = time.perf_counter()
                eng.query_topn(v_vecs[k], -1, topn)
                lat.append((time.perf_counter() - t1) * 1e6)
            lat.sort()
            shape["one_query_alone_by_value_p50_us"] = round(lat[len(lat) // 2], 1)
            if topn <= 128:
                for nb in (12, 32):
                    sel = np.array(q_rows[400:400 + nb], dtype=np.int64)
                    qv = t[torch.from_numpy(sel).to(dev)].cpu().numpy()
                    rings = [torch.zeros(nb * topn, dtype=torch.int64, device=dev) for _ in range(4)]
                    def run(calls):
                        for k in range(calls):
                            eng.enqueue_batch_keys_streamed(qv, sel, topn, rings[k % 4])
                        eng.enqueue_flush()
                        torch.cuda.synchronize()
                    run(4)
                    # (no HIP events inside the timed rounds — an event pair costs a launch ~6 us of stream time, which round 5's
                    # figure for this leg included; the median of three rounds of 20 calls, flush inside; then the kernel's mean
                    # duration from an untimed pass with events)
                    rounds = []
                    for _ in range(3):
                        t1 = time.perf_counter()
                        run(20)
                        rounds.append((time.perf_counter() - t1) / 20)
                    dt = sorted(rounds)[1]
                    eng.set_timing(1)
                    run(20)
                    k_ms = float(eng.stats().last_scan_ms)
                    eng.set_timing(False)
                    shape[f"pass_of_{nb}_streamed"] = {"us_per_call": round(dt * 1e6, 1), "queries_per_s": round(nb / dt, 1),
                                                       "launch_kernel_us": round(k_ms * 1e3, 1)}
                    checks.append((int(sel[nb - 1]), rings[3][(nb - 1) * topn:nb * topn].clone()))
                bq = min(args.batch, 1024)
                bsel = torch.from_numpy(np.array(q_rows[500:500 + bq], dtype=np.int64)).to(dev)
                qd = t[bsel].contiguous()
                keys = torch.zeros(bq * topn, dtype=torch.int64, device=dev)
                eng.enqueue_batch_keys_dev(qd, bsel, topn, keys)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    eng.enqueue_batch_keys_dev(qd, bsel, topn, keys)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 5
                d = eng.batched_last_counters()
                shape[f"batch_of_{bq}"] = {"ms_per_call": round(dt * 1e3, 4), "queries_per_s": round(bq / dt, 1),
                                           "candidates_per_query": round(d["candidates_total"] / max(1, bq - d["queued_queries"]), 1),
                                           "candidates_max": d["candidates_max"], "queued_to_exact_scan": d["queued_queries"]}
                checks.append((q_rows[500 + bq // 2], keys[(bq // 2) * topn:(bq // 2 + 1) * topn].clone()))
        host = t.cpu().numpy()
        ok = True
        for row, got in checks:
            if isinstance(got, tuple):
                idx, sc = got
            else:
                idx, sc = unpack_keys(got.cpu().numpy())
            if isinstance(row, tuple):   # ("by value", vector): nothing excluded
                want = oracle.scores(host, np.ascontiguousarray(row[1]), threads=0)
                row = -1
            else:
                want = oracle.scores(host, host[row], threads=0)
            ci, cs = oracle.topn_canonical(want, row, topn)
            ok = ok and np.asarray(idx).tolist() == ci.tolist() and bool(np.array_equal(np.asarray(sc), cs + np.float32(0)))
        shape["verified_against_oracle"] = bool(ok)
        shape["verified_queries"] = len(checks)
        out["shapes"].append(shape)
        del t, host
        torch.cuda.empty_cache()
    return out


def config0_object(torch, np, oracle):
    """BASELINE configs[0]: a 114 000-track Spotify-shaped CSV (114 genres x 1000 tracks, grouped by genre like the Kaggle
    file) -> DataManager preprocessing -> songs_data.bin -> top-10, on the GPU path (a query alone and a stream) and — in a
    FRESH CHILD PROCESS started with the devices hidden, never a re-exec of this one — on the product's own CPU backend
    (csrc/cpu_backend.cpp), with the oracle's OpenMP port on the same features beside them and the load + initialize times
    of SURVEY.md §8(f) rank 3."""
    import ctypes
    import subprocess
    import tempfile
    from spotify_recommender_amd import CosineEngine, build, capi
    build.build_shim()
    L = ctypes.CDLL(str(build.LIB_SHIM))
    L.shim_preprocess.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.shim_fast_load.argtypes = [ctypes.c_char_p]
    L.shim_fast_load.restype = ctypes.c_void_p
    L.shim_fast_initialize.argtypes = [ctypes.c_void_p]
    L.shim_fast_free.argtypes = [ctypes.c_void_p]
    L.shim_load.argtypes = [ctypes.c_char_p]
    L.shim_load.restype = ctypes.c_void_p
    L.shim_free.argtypes = [ctypes.c_void_p]
    L.shim_song_features.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    n, topn = 114_000, 10
    out = {"workload": "BASELINE configs[0]: 114 000-track CSV (114 genres x 1000, grouped by genre), 12 features, top-10",
           "rows": n, "topn": topn}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        rng = np.random.default_rng(114)
        cols = ("track_id,track_name,artists,danceability,energy,key,loudness,mode,speechiness,"
                "acousticness,instrumentalness,liveness,valence,tempo,track_genre")
        lines = [cols]
        for g in range(114):
            block = rng.random((1000, 9))
            for i in range(1000):
                r = block[i]
                k = g * 1000 + i
                lines.append(f"t{k:06d},Track {k},Artist {k % 5000},{r[0]:.4f},{r[1]:.4f},{int(r[2] * 12)},"
                             f"{-60 * r[3]:.3f},{int(r[4] * 2)},{r[5]:.4f},{r[6]:.5f},{r[7] ** 6:.6f},"
                             f"{r[8]:.4f},{(r[0] + r[1]) / 2:.4f},{60 + 140 * r[2]:.3f},genre{g:03d}")
        csv = tmp / "dataset.csv"
        csv.write_text("\n".join(lines) + "\n")
        binf = tmp / "songs_data.bin"
        t0 = time.perf_counter()
        assert L.shim_preprocess(str(csv).encode(), str(binf).encode()) == 1
        out["preprocess_csv_s"] = round(time.perf_counter() - t0, 3)
        out["songs_data_bin_bytes"] = binf.stat().st_size
        t0 = time.perf_counter()
        fast = L.shim_fast_load(str(binf).encode())
        assert fast and L.shim_fast_initialize(fast) == 1
        out["load_and_initialize_s"] = {"loadCatalogue + initialize(matrix)": round(time.perf_counter() - t0, 4)}
        L.shim_fast_free(fast)
        t0 = time.perf_counter()
        slow = L.shim_load(str(binf).encode())
        L.shim_initialize.argtypes = [ctypes.c_void_p]
        assert slow and L.shim_initialize(slow) == 1
        out["load_and_initialize_s"]["loadData + initialize(vector<Song>) (the reference's path)"] = round(time.perf_counter() - t0, 4)
        feats = np.zeros((n, 12), np.float32)
        gid = ctypes.c_int(0)
        for i in range(n):
            L.shim_song_features(slow, i, feats[i].ctypes.data, ctypes.byref(gid))
        L.shim_free(slow)
        np.save(tmp / "feats.npy", feats)
        q_rows = [(k * 7919) % n for k in range(1200)]
        # (i) the GPU path
        with CosineEngine(feats) as eng:
            st = eng.stats()
            for k in range(50):
                eng.query_row_topn(q_rows[k], topn)
            lat = []
            for k in range(500):
                t1 = time.perf_counter()
                res = eng.query_row_topn(q_rows[50 + k], topn)
                lat.append((time.perf_counter() - t1) * 1e6)
            lat.sort()
            leg, last = stream_leg(eng, torch, q_rows, topn, 1000, 100)
            # ... and the same stream over two lanes of the handle (mi355rec_create_lane), bound calls, each lane on its own stream
            lane = eng.lane()
            pair = [eng, lane]
            calls = [e.bound_enqueue_row_keys_streamed(topn, e.own_stream()) for e in pair]
            lrings = [[torch.zeros(topn, dtype=torch.int64, device="cuda") for _ in range(4)] for _ in pair]
            lptrs = [[ctypes.c_void_p(t.data_ptr()) for t in rs] for rs in lrings]
            torch.cuda.synchronize()

            def lanes_run(k0, k1):
                for k in range(k0, k1):
                    calls[k & 1](int(q_rows[k % len(q_rows)]), lptrs[k & 1][(k >> 1) & 3])
                for e in pair:
                    e.enqueue_flush(stream=e.own_stream())
                torch.cuda.synchronize()
            lanes_run(0, 100)
            t1 = time.perf_counter()
            lanes_run(100, 2100)
            dt2 = (time.perf_counter() - t1) / 2000
            lane.close()
            out["gpu"] = {"two_lanes_streamed_queries_per_s": round(1.0 / dt2, 1), "two_lanes_streamed_us_per_query": round(dt2 * 1e6, 2),
                          "route": "fp32 rows (48 B/row; shards below 1 M rows are launch-bound either way)" if not st.replica_active else "replica",
                          "one_query_alone_p50_us": round(lat[len(lat) // 2], 1), "one_query_alone_p99_us": round(lat[int(len(lat) * 0.99)], 1),
                          "streamed_queries_per_s": leg["queries_per_s"], "streamed_us_per_query": leg["us_per_step"],
                          "scan_kernel_us": leg["scan_kernel_us"]}
            want = oracle.scores(feats, feats[q_rows[549]], threads=0)
            ci, cs = oracle.topn_canonical(want, q_rows[549], topn)
            out["gpu"]["verified_against_oracle"] = bool(res[0].tolist() == ci.tolist() and np.array_equal(res[1], cs + np.float32(0)))
        # (ii) the product's CPU backend, in a child that never sees a device
        child = ("import sys, time, json, numpy as np\n"
                 f"sys.path.insert(0, {str(ROOT)!r})\n"
                 "from spotify_recommender_amd import capi\n"
                 "from spotify_recommender_amd.engine import NodeEngine\n"
                 f"f = np.load({str(tmp / 'feats.npy')!r})\n"
                 "assert capi.lib().mi355rec_device_count() == 0, 'the child must not see a device'\n"
                 "node = NodeEngine(f, n_devices=0, placement=capi.PLACEMENT_AUTO)\n"
                 "assert node.placement() == capi.PLACEMENT_CPU\n"
                 f"rows = [(k * 7919) % {n} for k in range(4000)]\n"
                 f"for k in range(20): node.query_row_topn(rows[k], {topn})\n"
                 "lat = []\n"
                 "t0 = time.perf_counter(); done = 0\n"
                 "while time.perf_counter() - t0 < 4.0:\n"
                 "    t1 = time.perf_counter()\n"
                 f"    r = node.query_row_topn(rows[20 + done % 3000], {topn})\n"
                 "    lat.append(time.perf_counter() - t1); done += 1\n"
                 "dt = time.perf_counter() - t0\n"
                 "lat.sort()\n"
                 f"last = node.query_row_topn(rows[7], {topn})\n"
                 "print(json.dumps({'queries_per_s': round(done / dt, 1), 'p50_us': round(lat[len(lat) // 2] * 1e6, 1), 'queries': done,\n"
                 "                  'note': node.note(), 'idx': last[0].tolist(), 'score_bits': np.asarray(last[1]).view(np.uint32).tolist()}))\n")
        env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
        p = subprocess.run([sys.executable, "-c", child], capture_output=True, text=True, env=env, timeout=180)
        if p.returncode == 0:
            r = json.loads(p.stdout.strip().splitlines()[-1])
            want = oracle.scores(feats, feats[q_rows[7]], threads=0)
            ci, cs = oracle.topn_canonical(want, q_rows[7], topn)
            import re
            m = re.search(r"\((\d+) ", r["note"])
            threads = int(m.group(1)) if m else None
            out["cpu_backend"] = {"queries_per_s": r["queries_per_s"], "one_query_p50_us": r["p50_us"], "sample": f"{r['queries']} queries (4 s)",
                                  "threads": threads, "note": r["note"], "process": "fresh child, HIP_VISIBLE_DEVICES='' before any GPU call",
                                  "verified_against_oracle": bool(r["idx"] == ci.tolist() and
                                                                  r["score_bits"] == (cs + np.float32(0)).view(np.uint32).tolist())}
        else:
            out["cpu_backend"] = {"unavailable": (p.stderr or p.stdout)[-400:]}
        # (iii) the oracle's OpenMP port on the same features (the checker, timed as in cpu_baseline)
        threads, host = host_threads(oracle.max_threads())
        oracle.recommend_omp(feats, q_rows[0], topn, threads)
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < 3.0:
            oracle.recommend_omp(feats, q_rows[done % 1000], topn, threads)
            done += 1
        out["oracle_port"] = {"queries_per_s": round(done / (time.perf_counter() - t0), 1), "threads": threads, "sample": f"{done} queries (3 s)",
                              "kind": "port (oracle/cosine_oracle.c, OpenMP rows + per-thread top-N)"}
        oracle.recommend_by_index(feats, q_rows[0], topn)
        t0 = time.perf_counter()
        done = 0
        while time.perf_counter() - t0 < 2.0:
            oracle.recommend_by_index(feats, q_rows[done % 1000], topn)
            done += 1
        out["oracle_port"]["serial_reference_loop_qps"] = round(done / (time.perf_counter() - t0), 1)
    return out


def preflight(args, json_fd, torch, np, capi, NodeEngine, feats_host, devices, virtual, oracle):
    """`--gpus N --preflight`: what a first contact with N real GPUs should say BEFORE anything is timed — which device can
    map which (the PEER transport stores keys through those mappings and reads query rows through them), the placement and
    shard count AUTO would choose for this catalogue, the device memory a shard needs against what each device has free —
    and ONE query per transport and placement through the product's node handle against the oracle, so that a
    misconfigured node fails here with a sentence, not in the timed stream with a hang."""
    n, topn = args.rows, args.topn
    g = len(devices)
    real = sorted(set(devices))
    report = {"preflight": True, "gpus": g, "virtual": bool(virtual), "rows": n, "topn": topn, "devices_visible": torch.cuda.device_count()}
    peer = {}
    for a in real:
        peer[str(a)] = {str(b): (bool(torch.cuda.can_device_access_peer(a, b)) if a != b else True) for b in real}
    report["peer_access"] = peer
    report["all_pairs_peer"] = all(all(v.values()) for v in peer.values())
    auto = int(capi.lib().mi355rec_auto_shards(n, torch.cuda.device_count()))
    report["auto_placement"] = {"shards": auto, "rule": "SHARDED over clamp(rows // 4 M, 1, visible devices) devices (include/mi355rec.h, PLACEMENT)",
                                "rows_per_shard": -(-n // max(1, auto))}
    rows_per = -(-n // g)
    report["memory"] = {"bytes_per_row_resident": 84, "note": "48 B fp32 row + 24 B fp16 replica + 12 B 8-bit replica",
                        "sharded_bytes_per_device": rows_per * 84, "replicated_bytes_per_device": n * 84,
                        "free_bytes": {str(d): int(torch.cuda.mem_get_info(d)[0]) for d in real}}
    row = (7 * 7919) % n
    want = oracle.scores(feats_host, feats_host[row], threads=0)
    ci, cs = oracle.topn_canonical(want, row, topn)
    checks = []
    ok_all = True
    plan = [(capi.PLACEMENT_SHARDED, capi.TRANSPORT_PEER, "sharded / peer stores"),
            (capi.PLACEMENT_SHARDED, capi.TRANSPORT_RCCL, "sharded / one ncclAllGather per rank"),
            (capi.PLACEMENT_REPLICATED, None, "replicated (no exchange)")]
    for placement, transport, label in plan:
        entry = {"what": label}
        try:
            if transport == capi.TRANSPORT_RCCL and virtual:
                raise RuntimeError("RCCL wants one device per rank: not with virtual shards of one GPU")
            node = NodeEngine(feats_host, devices=devices, placement=placement)
            try:
                if transport is not None:
                    node.set_transport(transport)
                idx, sc = node.query_row_topn(row, topn)             # the synchronous call
                t = node.enqueue_row(row, topn)                      # ... and the ticketed stream
                node.enqueue_flush()
                idx2, sc2 = node.wait(t, topn)
                good = (idx.tolist() == ci.tolist() and bool(np.array_equal(sc, cs + np.float32(0))) and
                        idx2.tolist() == ci.tolist() and bool(np.array_equal(sc2, cs + np.float32(0))))
                entry.update({"matches_oracle": bool(good), "rows_by_pointer": node.rows_by_pointer(), "note": node.note(),
                              "shard_rows": node.info()["shard_rows"]})
                ok_all = ok_all and good
            finally:
                node.close()
        except Exception as e:
            entry["failed"] = str(e)[:300]
            if not (transport == capi.TRANSPORT_RCCL and virtual):
                ok_all = False
        checks.append(entry)
    report["one_query_per_transport"] = checks
    report["ok"] = bool(ok_all)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(report) + "\n").encode())
    if not ok_all:
        raise SystemExit("preflight FAILED: " + "; ".join(f"{c['what']}: {c.get('failed', 'result differs from the oracle')}"
                                                          for c in checks if c.get("failed") or c.get("matches_oracle") is False))



def rank_group_report(dist, torch, rank: int, world: int, tensor_device, device_label: str, device_uuid: str,
                      kernel_ms: float, alg_bytes: int, peak_gbps: float = HBM_PEAK_GBPS):
    """What an N > 1 bench line says about the process group it ran on, so that a first record from a real node answers "did
    the collective see N ranks, on N different GPUs" by itself.  Called by EVERY rank (it contains
    collectives); returns the object on every rank.
      ranks                   what the process group reports (dist.get_world_size)
      ranks_counted           an all-reduce of ones over that group — MEASURED, not read back from the launcher's environment
      backend                 "nccl" (= RCCL on ROCm) under the driver's torch.distributed.run launch, "gloo" in the CPU tests
      devices / distinct_devices   each rank's device (name + uuid), gathered; N distinct ones on a real node
      per_gpu                 each rank's own scan kernel: mean duration (HIP events), algorithmic GB/s and its fraction of `peak_gbps`
    """
    ones = torch.ones(1, dtype=torch.int64, device=tensor_device)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    mine = {"rank": rank, "device": device_label, "uuid": device_uuid, "avg_kernel_ms": round(float(kernel_ms), 5) if kernel_ms else None,
            "algorithmic_bytes_per_launch": int(alg_bytes)}
    if kernel_ms and kernel_ms > 0:
        g = alg_bytes / (kernel_ms * 1e-3) / 1e9
        mine["achieved_gbps"] = round(g, 1)
        mine["frac"] = round(g / peak_gbps, 4)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    uuids = [g["uuid"] for g in gathered]
    return {"backend": str(dist.get_backend()), "ranks": int(dist.get_world_size()), "ranks_counted": int(ones.item()),
            "launcher_world_size": world, "devices": [f'{g["device"]} [{g["uuid"]}]' for g in gathered],
            "distinct_devices": len(set(uuids)), "per_gpu": gathered,
            "note": "ranks_counted = an all-reduce of ones over the group the exchange uses; per_gpu = each rank's own scan kernel "
                    "(HIP events) against the HBM peak"}


def peer_access_matrix(torch, devices):
    """Which device can map which (what the PEER transport stores keys through): {a: {b: bool}} over `devices`."""
    real = sorted(set(devices))
    return {str(a): {str(b): (bool(torch.cuda.can_device_access_peer(a, b)) if a != b else True) for b in real} for a in real}
