"""ctypes wrapper of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; never from the product package (spotify_recommender_amd/).
"""
from __future__ import annotations

import ctypes
import subprocess
from ctypes import c_float, c_int, c_int32, c_int64, c_uint32, c_void_p
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "liboracle.so"
_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        src = HERE / "cosine_oracle.c"
        if not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime:
            subprocess.run(["make", "-C", str(HERE), "liboracle.so"], check=True, capture_output=True)
        L = ctypes.CDLL(str(LIB_PATH))
        L.oracle_scores.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p]
        L.oracle_scores.restype = None
        L.oracle_scores_omp.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int]
        L.oracle_scores_omp.restype = None
        L.oracle_topn_heap.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_void_p]
        L.oracle_topn_heap.restype = c_int64
        L.oracle_topn_canonical.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]
        L.oracle_topn_canonical.restype = c_int64
        L.oracle_recommend_by_index.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p]
        L.oracle_recommend_by_index.restype = c_int64
        L.oracle_recommend_omp.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int]
        L.oracle_recommend_omp.restype = c_int64
        L.oracle_mt19937_uniform.argtypes = [c_uint32, c_int64, c_void_p]
        L.oracle_mt19937_uniform.restype = None
        L.oracle_max_threads.argtypes = []
        L.oracle_max_threads.restype = c_int
        _lib = L
    return _lib


_native = None


def native_recommend_omp():
    """The OpenMP baseline compiled for THIS host (-O3 -march=native, still -ffp-contract=off): bench.py's
    `cpu_baseline.march_native` variant ("state -march").  Built into a temporary directory at
    run time — a -march=native object must not travel to another machine — and never used as a checker."""
    global _native
    if _native is None:
        import tempfile
        d = Path(tempfile.mkdtemp(prefix="oracle_native_"))
        out = d / "liboracle_native.so"
        flags = ["-std=c11", "-O3", "-march=native", "-fPIC", "-fopenmp", "-ffp-contract=off"]
        subprocess.run(["gcc", *flags, "-shared", "-o", str(out), str(HERE / "cosine_oracle.c"), "-lm"], check=True,
                       capture_output=True)
        L = ctypes.CDLL(str(out))
        L.oracle_recommend_omp.argtypes = [c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int]
        L.oracle_recommend_omp.restype = c_int64
        _native = (L, "gcc " + " ".join(flags))
    L, flags = _native

    def run(feats, song_index: int, topn: int, threads: int = 0):
        a = _feats(feats)
        idx = np.empty(max(topn, 1), dtype=np.int32)
        sc = np.empty(max(topn, 1), dtype=np.float32)
        c = L.oracle_recommend_omp(a.ctypes.data, a.strides[0] // 4, a.shape[0], song_index, topn,
                                   idx.ctypes.data, sc.ctypes.data, threads)
        return idx[:max(c, 0)].copy(), sc[:max(c, 0)].copy()

    return run, flags


def _feats(feats) -> np.ndarray:
    a = np.asarray(feats)
    assert a.dtype == np.float32 and a.ndim == 2 and a.shape[1] >= 12
    assert a.strides[1] == 4 and a.strides[0] % 4 == 0
    return a


def scores(feats, query, threads: int = 1) -> np.ndarray:
    """Recommender.cu:256-273 over rows of `feats` (any row stride)."""
    a = _feats(feats)
    q = np.ascontiguousarray(np.asarray(query, dtype=np.float32).reshape(12))
    out = np.empty(a.shape[0], dtype=np.float32)
    if threads == 1:
        lib().oracle_scores(a.ctypes.data, a.strides[0] // 4, a.shape[0], q.ctypes.data, out.ctypes.data)
    else:
        lib().oracle_scores_omp(a.ctypes.data, a.strides[0] // 4, a.shape[0], q.ctypes.data,
                                out.ctypes.data, threads)
    return out


def topn_heap(score_vec, exclude: int, topn: int) -> np.ndarray:
    """Recommender.cu:293-315: the reference's exact output order."""
    s = np.ascontiguousarray(np.asarray(score_vec, dtype=np.float32))
    out = np.empty(max(topn, 1), dtype=np.int32)
    c = lib().oracle_topn_heap(s.ctypes.data, s.shape[0], exclude, topn, out.ctypes.data)
    return out[:c].copy()


def topn_canonical(score_vec, exclude: int, topn: int):
    s = np.ascontiguousarray(np.asarray(score_vec, dtype=np.float32))
    idx = np.empty(max(topn, 1), dtype=np.int32)
    sc = np.empty(max(topn, 1), dtype=np.float32)
    c = lib().oracle_topn_canonical(s.ctypes.data, s.shape[0], exclude, topn, idx.ctypes.data, sc.ctypes.data)
    return idx[:c].copy(), sc[:c].copy()


def recommend_by_index(feats, song_index: int, topn: int) -> np.ndarray:
    """Recommender.cu:275-318 (serial, reference order)."""
    a = _feats(feats)
    out = np.empty(max(topn, 1), dtype=np.int32)
    c = lib().oracle_recommend_by_index(a.ctypes.data, a.strides[0] // 4, a.shape[0], song_index,
                                        topn, out.ctypes.data, None)
    if c < 0:
        return np.empty(0, dtype=np.int32)
    return out[:c].copy()


def recommend_omp(feats, song_index: int, topn: int, threads: int = 0):
    """OpenMP rows + per-thread top-N + merge (canonical order)."""
    a = _feats(feats)
    idx = np.empty(max(topn, 1), dtype=np.int32)
    sc = np.empty(max(topn, 1), dtype=np.float32)
    c = lib().oracle_recommend_omp(a.ctypes.data, a.strides[0] // 4, a.shape[0], song_index, topn,
                                   idx.ctypes.data, sc.ctypes.data, threads)
    if c < 0:
        return np.empty(0, dtype=np.int32), np.empty(0, dtype=np.float32)
    return idx[:c].copy(), sc[:c].copy()


def mt19937_uniform(seed: int, rows: int, cols: int = 12) -> np.ndarray:
    """std::mt19937(seed) + uniform_real_distribution<float>(0,1), row-major."""
    out = np.empty((rows, cols), dtype=np.float32)
    lib().oracle_mt19937_uniform(seed, rows * cols, out.ctypes.data)
    return out


def max_threads() -> int:
    return int(lib().oracle_max_threads())
