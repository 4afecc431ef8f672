"""The C++ drop-in (Recommender class + CLI) on the GPU, checked against the
oracle exactly as the reference's main.cpp would drive it."""
import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def shim():
    import torch  # one HIP runtime per process: torch's first (see capi.lib)
    assert torch.cuda.is_available()
    from spotify_recommender_amd import build, capi
    capi.lib()
    build.build_shim()
    L = ctypes.CDLL(str(build.LIB_SHIM))
    L.shim_from_matrix.argtypes = [ctypes.c_void_p, ctypes.c_int64]
    L.shim_from_matrix.restype = ctypes.c_void_p
    L.shim_load.argtypes = [ctypes.c_char_p]
    L.shim_load.restype = ctypes.c_void_p
    for name in ("shim_free", "shim_initialize", "shim_is_initialized", "shim_is_gpu_enabled", "shim_get_song_count"):
        getattr(L, name).argtypes = [ctypes.c_void_p]
    L.shim_recommend_by_index.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    L.shim_recommend_by_index.restype = ctypes.c_int64
    for name in ("shim_recommend", "shim_recommend_by_name"):
        getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
        getattr(L, name).restype = ctypes.c_int64
    L.shim_similarities.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    return L


def rec(L, h, fn, key, topn):
    out = np.full(max(topn, 1), -1, np.int32)
    sc = np.zeros(max(topn, 1), np.float32)
    n = fn(h, key, topn, out.ctypes.data, sc.ctypes.data, max(topn, 1))
    return out[:max(n, 0)], sc[:max(n, 0)]


def test_recommender_class_matches_oracle(shim, golden_dir):
    g = np.load(golden_dir / "catalogue4096.npz")
    f = np.ascontiguousarray(g["feats"])
    h = shim.shim_from_matrix(f.ctypes.data, f.shape[0])
    try:
        assert shim.shim_is_initialized(h) == 0
        idx, _ = rec(shim, h, shim.shim_recommend_by_index, 5, 10)
        assert len(idx) == 0                                # "Recommender not initialized"
        assert shim.shim_initialize(h) == 1
        assert shim.shim_is_initialized(h) == 1 and shim.shim_is_gpu_enabled(h) == 1
        assert shim.shim_get_song_count(h) == 4096
        for i, q in enumerate(g["queries"]):
            for topn in (1, 10, 100):
                idx, sc = rec(shim, h, shim.shim_recommend_by_index, int(q), topn)
                assert_topn_matches(idx, sc, g["scores"][i], int(q), topn, ref_idx=g[f"heap_top{topn}"][i])
            sims = np.zeros(4096, np.float32)
            assert shim.shim_similarities(h, int(q), sims.ctypes.data) == 1
            assert np.array_equal(sims.view(np.uint32), g["scores"][i].view(np.uint32))
        # lookups: exact id; name exact (case-insensitive) before substring; first match wins
        idx, _ = rec(shim, h, shim.shim_recommend, b"id1234", 10)
        assert idx.tolist() == g["canon_top10"][list(g["queries"]).index(1234)].tolist()
        idx_a, _ = rec(shim, h, shim.shim_recommend_by_name, b"song 7", 10)     # exact "Song 7", not "Song 70"
        assert idx_a.tolist() == g["canon_top10"][list(g["queries"]).index(7)].tolist()
        idx_b, _ = rec(shim, h, shim.shim_recommend_by_name, b"ONG 409", 5)     # substring -> first = "Song 409"
        want = oracle.topn_canonical(oracle.scores(f, f[409]), 409, 5)[0]
        assert idx_b.tolist() == want.tolist()
        for bad in (b"nope", b""):
            pass
        assert len(rec(shim, h, shim.shim_recommend, b"missing-id", 5)[0]) == 0
        assert len(rec(shim, h, shim.shim_recommend_by_name, b"zzzz", 5)[0]) == 0
        assert len(rec(shim, h, shim.shim_recommend_by_index, -1, 5)[0]) == 0
        assert len(rec(shim, h, shim.shim_recommend_by_index, 4096, 5)[0]) == 0
        assert len(rec(shim, h, shim.shim_recommend_by_index, 5, 0)[0]) == 0
    finally:
        shim.shim_free(h)


def test_cli_end_to_end(tmp_path, golden_dir):
    """recommender --preprocess / --id / --song, as quickstart.sh drives the reference."""
    from spotify_recommender_amd import build
    build.build_shim()
    exe = str(build.BIN_CLI)
    env = dict(os.environ)
    run = lambda *a: subprocess.run([exe, *a], cwd=tmp_path, capture_output=True, text=True, env=env)
    p = run("--preprocess", str(golden_dir / "sample_songs.csv"))
    assert p.returncode == 0 and "Valid songs: 4 out of 6" in p.stdout and "Unique genres: 2" in p.stdout
    assert (tmp_path / "songs_data.bin").read_bytes() == (golden_dir / "sample_songs_data.bin").read_bytes()
    p = run("--id", "dupA", "-n", "10")
    assert p.returncode == 0, p.stderr
    assert "Top 3 Recommendations" in p.stdout          # topN > N-1 -> N-1
    assert "Successfully initialized with 4 songs on GPU" in p.stdout
    want = np.load(golden_dir / "sample_loaded.npz")
    f = want["feats"]
    order = oracle.topn_canonical(oracle.scores(f, f[3]), 3, 3)[0]
    names = [want["strings"][i][1] for i in order]
    pos = [p.stdout.index(f'"{n}"') for n in names]
    assert pos == sorted(pos)                           # printed best first
    p = run("--song", "lala")
    assert p.returncode == 0 and "Title:   Lalala" in p.stdout
    p = run("--song", "no such song")
    assert p.returncode == 1 and "not found" in p.stderr
    p = run("--id", "dupA", "-n", "0")
    assert p.returncode == 1 and "must be positive" in p.stderr
    # an absurd topN is clamped to N-1 like the reference's heap (Recommender.cu:300): no
    # buffer is sized by it, no scan round is run for it
    p = run("--id", "dupA", "-n", "2000000000")
    assert p.returncode == 0 and "Top 3 Recommendations" in p.stdout, p.stderr
    assert run().returncode == 1 and run("--bogus").returncode == 1


def test_config1_114k_csv_through_the_cli_classes(shim, tmp_path):
    """BASELINE configs[0] plumbing: a 114 000-row Spotify-shaped CSV (114 genres
    x 1000 tracks, grouped by genre like the Kaggle file) -> preprocess ->
    songs_data.bin -> Recommender, top-10, against the oracle on the loaded
    features.  (The reference runs this config on its CPU path; here it is the
    same HIP path.)"""
    from tests.test_datamanager import make_csv
    csv = tmp_path / "dataset.csv"
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
    csv.write_text("\n".join(lines) + "\n")
    shim.shim_preprocess.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    out = tmp_path / "songs_data.bin"
    assert shim.shim_preprocess(str(csv).encode(), str(out).encode()) == 1
    h = shim.shim_load(str(out).encode())
    assert h
    try:
        assert shim.shim_initialize(h) == 1 and shim.shim_get_song_count(h) == 114_000
        shim.shim_song_features = shim.shim_song_features
        feats = np.zeros((114_000, 12), np.float32)
        shim.shim_song_features.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        g = ctypes.c_int(0)
        for i in range(114_000):
            shim.shim_song_features(h, i, feats[i].ctypes.data, ctypes.byref(g))
        assert feats[:, :11].min() == 0.0 and feats[:, :11].max() == 1.0      # min-max normalised
        assert feats[113_999, 11] == 1.0 and feats[0, 11] == 0.0              # genre_id / (G-1)
        for q in (0, 56_789, 113_999):
            want = oracle.scores(feats, feats[q])
            idx, sc = rec(shim, h, shim.shim_recommend_by_index, q, 10)
            assert_topn_matches(idx, sc, want, q, 10, ref_idx=oracle.topn_heap(want, q, 10))
        idx, _ = rec(shim, h, shim.shim_recommend, b"t056789", 10)
        assert idx.tolist() == oracle.topn_canonical(oracle.scores(feats, feats[56_789]), 56_789, 10)[0].tolist()
    finally:
        shim.shim_free(h)


def test_fast_loader_path_on_a_million_songs(shim, tmp_path):
    """the CLI's query modes load songs_data.bin through
    DataManager::loadCatalogue (one pass over a mapping -> matrix + ids + names + record
    offsets) and Recommender::initialize(matrix, ids, names); nothing builds vector<Song>.
    Same recommendations as the loadData + initialize(vector<Song>) path on a 1 M-song
    file, checked against the oracle; load + init wall times of both are reported."""
    import time
    L = shim
    for name, res, args in (("shim_fast_load", ctypes.c_void_p, [ctypes.c_char_p]),
                            ("shim_fast_initialize", ctypes.c_int, [ctypes.c_void_p]),
                            ("shim_fast_free", None, [ctypes.c_void_p]),
                            ("shim_fast_song_count", ctypes.c_int64, [ctypes.c_void_p]),
                            ("shim_fast_recommend", ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]),
                            ("shim_fast_recommend_by_name", ctypes.c_int64, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]),
                            ("shim_fast_song_string", ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64]),
                            ("shim_write_synthetic_bin", ctypes.c_int, [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int])):
        getattr(L, name).restype = res
        getattr(L, name).argtypes = args
    n = 1_000_000
    rng = np.random.default_rng(2026)
    f = rng.random((n, 12), dtype=np.float32)
    path = tmp_path / "songs_data.bin"
    assert L.shim_write_synthetic_bin(str(path).encode(), f.ctypes.data, n, 114) == 1
    t0 = time.perf_counter()
    slow = L.shim_load(str(path).encode())
    assert slow and L.shim_initialize(slow) == 1
    t_slow = time.perf_counter() - t0
    t0 = time.perf_counter()
    fast = L.shim_fast_load(str(path).encode())
    assert fast and L.shim_fast_initialize(fast) == 1
    t_fast = time.perf_counter() - t0
    print(f"\n1M songs ({path.stat().st_size / 1e6:.0f} MB): loadData + initialize {t_slow:.2f} s, "
          f"loadCatalogue + initialize(matrix) {t_fast:.2f} s")
    try:
        assert L.shim_fast_song_count(fast) == n
        for q in (0, 123_456, n - 1):
            a, _ = rec(L, slow, L.shim_recommend, f"id{q}".encode(), 10)
            out = np.full(10, -1, np.int32)
            cnt = L.shim_fast_recommend(fast, f"id{q}".encode(), 10, out.ctypes.data, 10)
            assert cnt == 10 and out.tolist() == a.tolist()
            want = oracle.scores(f, f[q], threads=0)
            ci, _ = oracle.topn_canonical(want, q, 10)
            assert out.tolist() == ci.tolist()
        out = np.full(5, -1, np.int32)
        assert L.shim_fast_recommend_by_name(fast, b"song 4242", 5, out.ctypes.data, 5) == 5   # exact, case-insensitive
        want = oracle.scores(f, f[4242], threads=0)
        assert out.tolist() == oracle.topn_canonical(want, 4242, 5)[0].tolist()
        buf = ctypes.create_string_buffer(64)
        for which, text in ((0, b"id777777"), (1, b"Song 777777"), (2, b"Artist %d" % (777777 % 977)), (3, b"genre-%d" % (777777 % 114))):
            k = L.shim_fast_song_string(fast, 777_777, which, buf, 64)
            assert buf.raw[:k] == text
        print(f"load + init: fast loader {t_fast:.3f} s, loadData + vector<Song> {t_slow:.3f} s")   # reported, not asserted
    finally:
        L.shim_free(slow)
        L.shim_fast_free(fast)
