"""The product's own CPU backend (csrc/cpu_backend.cpp) behind the node-level C-ABI — what
mi355rec_create_sharded / mi355rec_create_placed hand out on a host WITHOUT a HIP device, as the reference falls
back to its CPU loop (Recommender.cu:117-127,176-181,256-273).  BASELINE configs[0]: "114 k tracks, top-10,
CPU cosine path only".

Runs only where no GPU is visible (the build container; on the GPU box the backend is never taken — that is
asserted by tests/test_gpu_node.py).  Checked against the oracle (tests may use it; the product does not): score
vectors bit for bit, top-N tie-aware with the heap order as the reference order, the tie fixture, the 114 000-row
CSV end to end through the C++ classes and through the CLI.
"""
import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_canonical_order, assert_topn_matches

ROOT = Path(__file__).resolve().parents[1]


def _gpu_visible():
    from spotify_recommender_amd import capi
    return capi.lib().mi355rec_device_count() > 0


pytestmark = pytest.mark.skipif(_gpu_visible(), reason="a GPU is visible: the CPU backend is never taken here")


@pytest.fixture(scope="module")
def Node(engine_lib):
    from spotify_recommender_amd.engine import NodeEngine
    return NodeEngine


def test_the_node_handle_is_served_by_the_cpu_backend(Node, golden_dir):
    from spotify_recommender_amd import capi
    g = np.load(golden_dir / "catalogue4096.npz")
    f = np.ascontiguousarray(g["feats"])
    with Node(f, placement=capi.PLACEMENT_AUTO) as node:
        assert node.placement() == capi.PLACEMENT_CPU
        info = node.info()
        assert info["n_shards"] == 0 and info["rows"] == 4096
        assert "CPU backend" in node.note() and not node.rows_by_pointer()
        # the private calculateSimilarities mirror: every score, bit for bit
        for q in (0, 17, 4095):
            assert np.array_equal(node.scores_row(q).view(np.uint32), oracle.scores(f, f[q]).view(np.uint32))
        for i, q in enumerate(g["queries"]):
            want = oracle.scores(f, f[int(q)])
            for topn in (1, 10, 100, 5000):
                idx, sc = node.query_row_topn(int(q), topn)
                assert_topn_matches(idx, sc, want, int(q), topn, ref_idx=oracle.topn_heap(want, int(q), topn))
                assert_canonical_order(idx, want)
        # an arbitrary vector, nothing excluded; a batch
        rng = np.random.default_rng(3)
        vec = rng.random(12, dtype=np.float32)
        idx, sc = node.query_topn(vec, -1, 25)
        assert_topn_matches(idx, sc, oracle.scores(f, vec), -1, 25)
        qrows = rng.integers(0, 4096, size=9)
        bi, bs, counts = node.query_batch_topn(f[qrows], qrows, 20)
        for b, r in enumerate(qrows):
            assert_topn_matches(bi[b][:counts[b]], bs[b][:counts[b]], oracle.scores(f, f[r]), int(r), 20)
        # the same answers as the library's key helpers would order them (canonical order = key order)
        L = capi.lib()
        want = oracle.scores(f, f[5])
        keys = sorted((L.mi355rec_pack_key(float(want[r]), r) for r in range(4096) if r != 5), reverse=True)[:50]
        idx, _ = node.query_row_topn(5, 50)
        assert idx.tolist() == [L.mi355rec_key_row(k) for k in keys]
        with pytest.raises(capi.Mi355Error):
            node.query_row_topn(4096, 3)
        with pytest.raises(capi.Mi355Error):
            node.query_row_topn(0, 0)
        node.set_transport(capi.TRANSPORT_RCCL)       # nothing to exchange: accepted and ignored
        node.set_replica(capi.REPLICA_OFF)
    # an explicit device list cannot be honoured without a device; the single-device C-ABI has no CPU path
    with pytest.raises(capi.Mi355Error):
        Node(f, devices=[0])
    with pytest.raises(capi.Mi355Error):
        Node(f, n_devices=2)


def test_ties_and_degenerate_rows_in_canonical_order(Node):
    from spotify_recommender_amd import capi
    # the survey's 9-row tie case (tests/test_oracle_pins.py pins the reference's heap order on it): rows 1..8
    # identical to the query row 0 up to scale, so every score is exactly 1.0
    base = np.array([0.3, 0.1, 0.9, 0.5, 0.2, 0.7, 0.4, 0.6, 0.8, 0.15, 0.35, 0.55], np.float32)
    f = np.stack([base] * 9).astype(np.float32)
    with Node(f) as node:
        idx, sc = node.query_row_topn(0, 8)
        assert idx.tolist() == [1, 2, 3, 4, 5, 6, 7, 8] and np.all(sc == sc[0])
        idx, _ = node.query_row_topn(0, 3)
        assert idx.tolist() == [1, 2, 3]               # canonical: row ascending among equal scores
    # zero rows / zero query score exactly 0; NaN and inf follow the reference's comparisons
    rng = np.random.default_rng(8)
    f = rng.random((300, 12), dtype=np.float32) - np.float32(0.5)
    f[7] = 0.0
    f[9] = np.float32(1e-30)
    f[11, 3] = np.nan
    f[13, 0] = np.inf
    with Node(f) as node:
        for q in (0, 7, 9, 11, 13, 299):
            want = oracle.scores(f, f[q])
            assert np.array_equal(node.scores_row(q).view(np.uint32), want.view(np.uint32)), q
            idx, sc = node.query_row_topn(q, 40)
            assert_topn_matches(idx, sc, want, q, 40)
    one = rng.random((1, 12), dtype=np.float32)
    with Node(one) as node:
        idx, sc = node.query_row_topn(0, 10)            # nothing but the query itself
        assert len(idx) == 0


def test_the_stream_keeps_its_tickets_and_windows(Node):
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(5)
    n = 50_000
    f = rng.random((n, 12), dtype=np.float32)
    with Node(f) as node:
        node.set_window(2)
        with pytest.raises(capi.Mi355Error):
            node.wait(0, 10)
        tickets = [node.enqueue_row(int(r), 10) for r in range(24)]    # 12 windows of 2: a ring of 4 is kept
        assert tickets == list(range(24))
        node.enqueue_flush()
        with pytest.raises(capi.Mi355Error, match="no longer kept"):
            node.wait(tickets[0], 10)
        for r in (16, 21, 23):
            idx, sc = node.wait(tickets[r], 10)
            want = oracle.scores(f, f[r])
            assert_topn_matches(idx, sc, want, r, 10, ref_idx=oracle.topn_heap(want, r, 10))
        t = node.enqueue_row(3, 10)
        node.enqueue_flush()                                            # a flush rounds the next ticket up to a window
        t2 = node.enqueue_query(f[4], 4, 10)
        assert t == 24 and t2 == 26
        idx, sc = node.wait(t2, 10)
        assert_topn_matches(idx, sc, oracle.scores(f, f[4]), 4, 10)
        with pytest.raises(capi.Mi355Error):
            node.enqueue_row(n, 10)
        with pytest.raises(capi.Mi355Error):
            node.enqueue_row(0, 2000)
        st = node.stream_stats()
        assert st["queries"] == 26


@pytest.fixture(scope="module")
def shim(engine_lib):
    from spotify_recommender_amd import build
    build.build_shim()
    L = ctypes.CDLL(str(build.LIB_SHIM))
    L.shim_from_matrix.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.shim_from_matrix.restype = ctypes.c_void_p
    L.shim_load.argtypes = [ctypes.c_char_p]
    L.shim_load.restype = ctypes.c_void_p
    L.shim_preprocess.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    for name in ("shim_free", "shim_initialize", "shim_is_initialized", "shim_is_gpu_enabled", "shim_get_song_count"):
        getattr(L, name).argtypes = [ctypes.c_void_p]
    L.shim_recommend_by_index.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.shim_recommend_by_index.restype = ctypes.c_int64
    L.shim_recommend.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.shim_recommend.restype = ctypes.c_int64
    L.shim_song_features.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    return L


def _rec(fn, h, key, topn):
    out = np.full(max(topn, 1), -1, np.int32)
    sc = np.zeros(max(topn, 1), np.float32)
    n = fn(h, key, topn, out.ctypes.data, sc.ctypes.data, max(topn, 1))
    return out[:max(n, 0)], sc[:max(n, 0)]


def _config1_csv(path):
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
    path.write_text("\n".join(lines) + "\n")


def test_config1_114k_csv_on_the_cpu_backend(shim, tmp_path, capfd):
    """BASELINE configs[0]: the 114 000-row Spotify-shaped CSV -> preprocess -> songs_data.bin -> Recommender,
    top-10 — on the CPU backend, as the reference runs this config on its CPU path.  initialize() succeeds, says
    what the reference says in this situation and reports isGPUEnabled() == false."""
    csv = tmp_path / "dataset.csv"
    _config1_csv(csv)
    out = tmp_path / "songs_data.bin"
    assert shim.shim_preprocess(str(csv).encode(), str(out).encode()) == 1
    h = shim.shim_load(str(out).encode())
    assert h
    try:
        capfd.readouterr()
        assert shim.shim_initialize(h) == 1
        said = capfd.readouterr()
        assert "Operating in CPU fallback mode (cosine similarity on CPU)." in said.out      # Recommender.cu:177
        assert "Falling back to CPU similarity computation." in said.err                     # Recommender.cu:121
        assert shim.shim_is_initialized(h) == 1 and shim.shim_is_gpu_enabled(h) == 0
        assert shim.shim_get_song_count(h) == 114_000
        feats = np.zeros((114_000, 12), np.float32)
        g = ctypes.c_int(0)
        for i in range(114_000):
            shim.shim_song_features(h, i, feats[i].ctypes.data, ctypes.byref(g))
        for q in (0, 56_789, 113_999):
            want = oracle.scores(feats, feats[q])
            idx, sc = _rec(shim.shim_recommend_by_index, h, q, 10)
            assert_topn_matches(idx, sc, want, q, 10, ref_idx=oracle.topn_heap(want, q, 10))
        idx, _ = _rec(shim.shim_recommend, h, b"t056789", 10)
        assert idx.tolist() == oracle.topn_canonical(oracle.scores(feats, feats[56_789]), 56_789, 10)[0].tolist()
    finally:
        shim.shim_free(h)
    # ... and through the drop-in CLI (csrc/main.cpp), one thread and all of them: the same lines
    from spotify_recommender_amd import build
    runs = []
    for threads in ("1", None):
        env = dict(os.environ)
        if threads:
            env["OMP_NUM_THREADS"] = threads
        p = subprocess.run([str(build.BIN_CLI), "--id", "t056789", "-n", "10"], capture_output=True, text=True, cwd=tmp_path, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        assert "Operating in CPU fallback mode" in p.stdout and "Top 10" in p.stdout
        runs.append([l for l in p.stdout.splitlines() if "Track " in l])
    assert runs[0] == runs[1] and len(runs[0]) >= 11      # the query's block + ten recommendations


def test_out_of_memory_inside_a_query_is_an_error_code_not_a_terminate(tmp_path):
    """ADVICE r4 / VERDICT r4 item 4: everything a query allocates is allocated BEFORE its OpenMP region (an exception that
    leaves such a region is std::terminate behind the C-ABI).  A child process creates the CPU-backed node handle on 114 000
    rows, then lowers its address-space limit to what it already uses and asks for as many results as there are rows
    (a few MB of keys per thread): the call must come back with MI355REC_ERR_OUT_OF_MEMORY, the process must live, and
    with the limit lifted the same handle must answer again."""
    child = r'''
import resource, sys
import numpy as np
sys.path.insert(0, %r)
from spotify_recommender_amd import capi
from spotify_recommender_amd.engine import NodeEngine
rng = np.random.default_rng(3)
f = rng.random((114_000, 12), dtype=np.float32)
node = NodeEngine(f, n_devices=0, placement=capi.PLACEMENT_AUTO)
assert node.placement() == capi.PLACEMENT_CPU
idx, sc = node.query_row_topn(5, 10)
assert len(idx) == 10
out_idx = np.empty(114_000, np.int64); out_sc = np.empty(114_000, np.float32)   # the caller's buffers exist before the squeeze
import ctypes
vm = int([l for l in open("/proc/self/status") if l.startswith("VmSize")][0].split()[1]) * 1024
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
resource.setrlimit(resource.RLIMIT_AS, (vm + (1 << 20), hard))
cnt = ctypes.c_int(0)
rc = node._lib.mi355rec_sharded_query_row_topn(node._h, 5, 114_000, ctypes.c_void_p(out_idx.ctypes.data), ctypes.c_void_p(out_sc.ctypes.data), ctypes.byref(cnt))
resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
assert rc == capi.ERR_OUT_OF_MEMORY, rc
rc = node._lib.mi355rec_sharded_query_row_topn(node._h, 5, 114_000, ctypes.c_void_p(out_idx.ctypes.data), ctypes.c_void_p(out_sc.ctypes.data), ctypes.byref(cnt))
assert rc == capi.OK and cnt.value == 113_999, (rc, cnt.value)
assert out_idx[:10].tolist() == idx.tolist()
print("SURVIVED")
''' % str(ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    p = subprocess.run([os.sys.executable, "-c", child], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "SURVIVED" in p.stdout, (p.returncode, p.stdout[-400:], p.stderr[-800:])


def test_a_query_forks_no_more_threads_than_have_rows():
    """the team of a call is the threads that have work (num_threads(parts)), not the process's whole
    default team — the rest would spin at the join barrier and, under a cgroup CPU quota, burn what the workers need.
    Counted from outside: the threads of a child process after its first queries on 10 000 rows (two parts: a thread per
    8192 rows) with OMP_NUM_THREADS = 8."""
    child = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
os.environ["OMP_NUM_THREADS"] = "8"
from spotify_recommender_amd import capi
from spotify_recommender_amd.engine import NodeEngine
f = np.random.default_rng(1).random((10_000, 12), dtype=np.float32)
node = NodeEngine(f, n_devices=0, placement=capi.PLACEMENT_AUTO)
before = len(os.listdir("/proc/self/task"))
for r in range(20):
    node.query_row_topn(r, 10)
after = len(os.listdir("/proc/self/task"))
print("THREADS", before, after, node.note())
''' % str(ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", OMP_NUM_THREADS="8")
    p = subprocess.run([os.sys.executable, "-c", child], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-800:]
    before, after = [int(x) for x in p.stdout.split("THREADS")[1].split()[:2]]
    # 10 000 rows = two parts: the queries may add ONE worker thread to the caller's, not seven
    assert after - before <= 1, p.stdout
