"""ctypes binding of the C-ABI in include/mi355rec.h (core) and include/mi355rec_diag.h (statistics, controls, hooks).

This is plumbing only: every compute call goes to the HIP library.  If the
library is missing the import of :func:`lib` raises — there is no Python or CPU
fallback for the hot path.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p
from pathlib import Path

PKG = Path(__file__).resolve().parent
# MI355REC_LIB: another build of the same C-ABI (tests/test_gpu_experiments.py runs the A/B routes of the MI355REC_EXPERIMENTS
# build this way, in a child process; the product never sets it)
LIB_PATH = Path(os.environ["MI355REC_LIB"]) if os.environ.get("MI355REC_LIB") else PKG / "libmi355rec.so"

DIM = 12
MAX_TOPN_FAST = 1024
BATCH_AUTO, BATCH_MULTI, BATCH_MFMA, BATCH_HALF, BATCH_Q8, BATCH_MFMA_NOSKIP = 0, 1, 2, 3, 4, 5
REPLICA_AUTO, REPLICA_OFF, REPLICA_ON, REPLICA_FP16 = 0, 1, 2, 3
TRANSPORT_PEER, TRANSPORT_RCCL = 1, 2
PLACEMENT_AUTO, PLACEMENT_SHARDED, PLACEMENT_REPLICATED, PLACEMENT_CPU = 0, 1, 2, 3
CREATE_NO_REPLICA = 1
DEBUG_HANDOFF_POISON, DEBUG_HANDOFF_DROP_STORES, DEBUG_HANDOFF_NO_LAST_RIDER = 1, 2, 4
BUILD_EXPERIMENTS, BUILD_PHASE_CLOCK, BUILD_TEST_HOOKS = 1, 2, 4
PROBE_FP32_ROWS, PROBE_FP16_REPLICA, PROBE_Q8_REPLICA = 0, 1, 2

OK = 0
ERR_INVALID_ARG = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_OUT_OF_MEMORY = -4


class Stats(ctypes.Structure):
    _fields_ = [
        ("rows", c_int64),
        ("row_base", c_int64),
        ("device", c_int32),
        ("compute_units", c_int32),
        ("grid_blocks", c_int32),
        ("block_threads", c_int32),
        ("bytes_per_query", c_int64),
        ("last_scan_ms", c_float),
        ("last_merge_ms", c_float),
        ("last_pass_ms", c_float),
        ("batched_grid_blocks", c_int32),
        ("batched_margin", c_float),
        ("replica_bytes_per_query", c_int64),
        ("replica_active", c_int32),
        ("replica_grid_blocks", c_int32),
        ("replica_build_ms", c_float),
        ("replica_margin_single", c_float),
        ("replica_margin_multi", c_float),
        ("replica_single_bytes_per_query", c_int64),
        ("replica_single_row_bytes", c_int32),
        ("lone_fused_queries", c_int32),
        ("route_fp32", c_int64),
        ("route_fp16", c_int64),
        ("route_q8", c_int64),
        ("route_q8_lone", c_int64),
        ("route_multi_fp32", c_int64),
        ("route_multi_fp16", c_int64),
        ("route_multi_q8", c_int64),
        ("route_mfma_two_pass", c_int64),
        ("route_exact_queue", c_int64),
        ("device_bytes_per_row", c_int32),
    ]


# name -> (restype, argtypes); mirrors include/mi355rec.h one to one
SIGNATURES = {
    "mi355rec_build_flags": (c_int, []),
    "mi355rec_device_count": (c_int, []),
    "mi355rec_last_global_error": (c_char_p, []),
    "mi355rec_create": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, POINTER(c_void_p)]),
    "mi355rec_create_device": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, POINTER(c_void_p)]),
    "mi355rec_create_ex": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, c_int, POINTER(c_void_p)]),
    "mi355rec_create_device_ex": (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, c_int, POINTER(c_void_p)]),
    "mi355rec_destroy": (None, [c_void_p]),
    "mi355rec_last_error": (c_char_p, [c_void_p]),
    "mi355rec_create_lane": (c_int, [c_void_p, POINTER(c_void_p)]),
    "mi355rec_own_stream": (c_void_p, [c_void_p]),
    "mi355rec_lane_status": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "mi355rec_stats": (c_int, [c_void_p, POINTER(Stats)]),
    "mi355rec_stats_sized": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, POINTER(ctypes.c_size_t)]),
    "mi355rec_scores_row": (c_int, [c_void_p, c_int64, c_void_p]),
    "mi355rec_scores": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mi355rec_query_row_topn": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, POINTER(c_int)]),
    "mi355rec_query_topn": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, POINTER(c_int)]),
    "mi355rec_query_batch_topn": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mi355rec_enqueue_row_keys": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mi355rec_enqueue_query_keys": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mi355rec_enqueue_row_keys_streamed": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mi355rec_enqueue_query_keys_streamed": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mi355rec_enqueue_flush": (c_int, [c_void_p, c_void_p]),
    "mi355rec_enqueue_batch_keys": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mi355rec_enqueue_batch_keys_streamed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mi355rec_batch_pointers_ok": (c_int, [c_void_p, c_int]),
    "mi355rec_enqueue_batch_mixed_keys": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mi355rec_enqueue_batch_mixed_keys_streamed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mi355rec_enqueue_batch_keys_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "mi355rec_set_batch_path": (c_int, [c_void_p, c_int]),
    "mi355rec_set_replica": (c_int, [c_void_p, c_int]),
    "mi355rec_rebuild_replica": (c_int, [c_void_p]),
    "mi355rec_replica_counters": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    "mi355rec_batched_last_counters": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int64),
                                               POINTER(c_int32)]),
    "mi355rec_batched_pass2_pairs": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    "mi355rec_enqueue_merge_keys": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mi355rec_enqueue_merge_keys_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_int,
                                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    "mi355rec_enqueue_scores": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "mi355rec_enqueue_stream_probe": (c_int, [c_void_p, c_void_p, c_void_p]),
    "mi355rec_enqueue_stream_probe_of": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "mi355rec_set_timing": (c_int, [c_void_p, c_int]),
    "mi355rec_fetch_row": (c_int, [c_void_p, c_int64, c_void_p]),
    "mi355rec_create_sharded": (c_int, [c_void_p, c_int64, c_int, c_int, POINTER(c_void_p)]),
    "mi355rec_create_sharded_on": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, POINTER(c_void_p)]),
    "mi355rec_create_placed": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "mi355rec_auto_shards": (c_int, [c_int64, c_int]),
    "mi355rec_sharded_placement": (c_int, [c_void_p]),
    "mi355rec_sharded_destroy": (None, [c_void_p]),
    "mi355rec_sharded_last_error": (c_char_p, [c_void_p]),
    "mi355rec_sharded_set_transport": (c_int, [c_void_p, c_int]),
    "mi355rec_sharded_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int64), c_void_p, c_void_p]),
    "mi355rec_sharded_query_row_topn": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, POINTER(c_int)]),
    "mi355rec_sharded_query_topn": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, POINTER(c_int)]),
    "mi355rec_sharded_query_batch_topn": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "mi355rec_sharded_scores_row": (c_int, [c_void_p, c_int64, c_void_p]),
    "mi355rec_row_ptr": (c_int, [c_void_p, c_int64, POINTER(c_void_p)]),
    "mi355rec_enqueue_ptr_keys": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mi355rec_enqueue_ptr_keys_streamed": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "mi355rec_sharded_set_timing": (c_int, [c_void_p, c_int]),
    "mi355rec_sharded_shard_stats": (c_int, [c_void_p, c_int, POINTER(Stats)]),
    "mi355rec_sharded_set_replica": (c_int, [c_void_p, c_int]),
    "mi355rec_sharded_rows_by_pointer": (c_int, [c_void_p]),
    "mi355rec_sharded_note": (c_char_p, [c_void_p]),
    "mi355rec_sharded_set_window": (c_int, [c_void_p, c_int]),
    "mi355rec_sharded_set_window_mode": (c_int, [c_void_p, c_int]),
    "mi355rec_sharded_enqueue_row": (c_int, [c_void_p, c_int64, c_int, POINTER(c_int64)]),
    "mi355rec_sharded_enqueue_query": (c_int, [c_void_p, c_void_p, c_int64, c_int, POINTER(c_int64)]),
    "mi355rec_sharded_enqueue_flush": (c_int, [c_void_p]),
    "mi355rec_sharded_wait": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, POINTER(c_int)]),
    "mi355rec_sharded_stream_stats": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "mi355rec_debug_handoff": (c_int, [c_void_p, c_int]),
    "mi355rec_sharded_rccl_ranks": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "mi355rec_pack_key": (c_uint64, [c_float, c_int64]),
    "mi355rec_key_score": (c_float, [c_uint64]),
    "mi355rec_key_row": (c_int64, [c_uint64]),
}

# declared under #ifdef MI355REC_TEST_HOOKS in include/mi355rec_diag.h: absent from the product library
TEST_HOOKS = frozenset({"mi355rec_debug_handoff"})

_lib = None


def lib() -> ctypes.CDLL:
    """Load libmi355rec.so (built by ``spotify_recommender_amd.build``)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m spotify_recommender_amd.build` "
                "(hipcc --offload-arch=gfx950). The cosine top-N path has no fallback."
            )
        # One HIP runtime per process: torch bundles its own libamdhip64 and
        # whichever runtime touches the GPU first owns it.  Importing torch
        # first maps its runtime, and our library's libamdhip64.so dependency
        # then resolves to that same copy (same SONAME).  Without torch the
        # library uses /opt/rocm's runtime (the C++ CLI path).
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover
            pass
        handle = ctypes.CDLL(str(LIB_PATH))
        lenient = os.environ.get("MI355REC_CAPI_LENIENT") == "1"   # tools only: A/B against a library of an EARLIER round (--lib)
        for name, (restype, argtypes) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            except AttributeError:
                if lenient or name in TEST_HOOKS:   # (test hooks: only in -DMI355REC_TEST_HOOKS builds, see has_test_hooks)
                    continue
                raise
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def has_test_hooks() -> bool:
    """True when the loaded library was built with -DMI355REC_TEST_HOOKS (libmi355rec_testhooks.so, the experiments build):
    mi355rec_debug_handoff exists.  The product library returns False."""
    L = lib()
    return hasattr(L, "mi355rec_build_flags") and bool(L.mi355rec_build_flags() & BUILD_TEST_HOOKS)


def has_experiments() -> bool:
    """True when the loaded library is an MI355REC_EXPERIMENTS build (tools/: A/B routes and environment knobs)."""
    L = lib()
    if not hasattr(L, "mi355rec_build_flags"):   # (a library of an earlier round, loaded leniently by a tool)
        return False
    return bool(L.mi355rec_build_flags() & BUILD_EXPERIMENTS)


class Mi355Error(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mi355rec error {code}: {message}")
        self.code = code


def check(rc: int, handle=None) -> None:
    if rc == OK:
        return
    L = lib()
    msg = L.mi355rec_last_error(handle) if handle else L.mi355rec_last_global_error()
    raise Mi355Error(rc, (msg or b"").decode("utf-8", "replace"))
