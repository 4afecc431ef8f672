"""Host-side plumbing over the C-ABI: device memory via torch, streams, RCCL.

`CosineEngine` owns one catalogue shard on one GPU.  `ShardedEngine` is the
multi-GPU path required by BASELINE.json's north_star: one process per GPU,
rows sharded contiguously, ONE all-gather of `topn` packed keys per rank per
query over RCCL, then the same device merge kernel that merges the
per-workgroup lists.  No arithmetic happens in Python.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import numpy as np

from . import capi


def _np_f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class CosineEngine:
    """One row shard of the N x 12 fp32 catalogue resident on one MI355X.

    Replaces the reference's device state (d_features / d_queryFeature /
    d_similarities, Recommender.h:91-94) and its per-query pipeline
    (Recommender.cu:184-254 + :293-315).
    """

    def __init__(self, feats, device: int = 0, row_base: int = 0, flags: int = 0):
        """flags: capi.CREATE_NO_REPLICA = keep the fp32 rows only (48 B per row resident instead of 84)."""
        self._lib = capi.lib()
        self._h = ctypes.c_void_p()
        self._keepalive = None
        self.device = int(device)
        try:
            import torch
        except Exception:  # pragma: no cover - torch is part of the image
            torch = None
        if torch is not None and isinstance(feats, torch.Tensor):
            if not feats.is_cuda:
                feats = feats.numpy()
            else:
                if feats.dtype != torch.float32 or feats.dim() != 2 or feats.shape[1] != capi.DIM:
                    raise ValueError("catalogue must be float32 [n, 12]")
                if not feats.is_contiguous():
                    raise ValueError("catalogue tensor must be contiguous (row-major)")
                self._keepalive = feats
                self.device = feats.device.index if feats.device.index is not None else 0
                rc = self._lib.mi355rec_create_device_ex(
                    ctypes.c_void_p(feats.data_ptr()), feats.shape[0], feats.shape[1],
                    self.device, int(row_base), int(flags), ctypes.byref(self._h))
                capi.check(rc)
                self.rows = int(feats.shape[0])
                self.row_base = int(row_base)
                return
        arr = _np_f32(feats)
        if arr.ndim != 2 or arr.shape[1] != capi.DIM:
            raise ValueError("catalogue must be float32 [n, 12]")
        rc = self._lib.mi355rec_create_ex(
            arr.ctypes.data_as(ctypes.c_void_p), arr.shape[0], arr.shape[1], self.device,
            int(row_base), int(flags), ctypes.byref(self._h))
        capi.check(rc)
        self.rows = int(arr.shape[0])
        self.row_base = int(row_base)

    def lane(self) -> "CosineEngine":
        """Another handle over the same rows and replicas with its own stream state (mi355rec_create_lane): queries dealt
        alternately over two lanes on two streams overlap where one handle's launches cannot."""
        other = CosineEngine.__new__(CosineEngine)
        other._lib = self._lib
        other._h = ctypes.c_void_p()
        other._keepalive = self._keepalive
        other.device = self.device
        other.rows = self.rows
        other.row_base = self.row_base
        capi.check(self._lib.mi355rec_create_lane(self._h, ctypes.byref(other._h)), self._h)
        return other

    def lane_status(self) -> dict:
        """What mi355rec_create_lane found for this lane's stream: {"stream_attempts", "overlaps_parent" (1 / 0 / -1)}."""
        a, o = ctypes.c_int(0), ctypes.c_int(-1)
        capi.check(self._lib.mi355rec_lane_status(self._h, ctypes.byref(a), ctypes.byref(o)), self._h)
        return {"stream_attempts": a.value, "overlaps_parent": o.value}

    def own_stream(self):
        """The HIP stream the library created with this handle, as a torch stream (mi355rec_own_stream): the stream to run a
        lane on — streams taken from torch's pool later may share a hardware queue, and then lanes do not overlap."""
        import torch
        return torch.cuda.ExternalStream(int(self._lib.mi355rec_own_stream(self._h)), device=self.device)

    # -- lifetime ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mi355rec_destroy(self._h)
            self._h = ctypes.c_void_p()
        self._keepalive = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- synchronous host API ------------------------------------------------
    def scores_row(self, local_row: int) -> np.ndarray:
        out = np.empty(self.rows, dtype=np.float32)
        capi.check(self._lib.mi355rec_scores_row(self._h, int(local_row), out.ctypes.data_as(ctypes.c_void_p)), self._h)
        return out

    def scores(self, query) -> np.ndarray:
        q = _np_f32(query).reshape(capi.DIM)
        out = np.empty(self.rows, dtype=np.float32)
        capi.check(self._lib.mi355rec_scores(self._h, q.ctypes.data_as(ctypes.c_void_p),
                                             out.ctypes.data_as(ctypes.c_void_p)), self._h)
        return out

    def query_row_topn(self, local_row: int, topn: int) -> Tuple[np.ndarray, np.ndarray]:
        n_out = max(int(topn), 1)
        idx = np.empty(n_out, dtype=np.int64)
        score = np.empty(n_out, dtype=np.float32)
        count = ctypes.c_int(0)
        capi.check(self._lib.mi355rec_query_row_topn(
            self._h, int(local_row), int(topn), idx.ctypes.data_as(ctypes.c_void_p),
            score.ctypes.data_as(ctypes.c_void_p), ctypes.byref(count)), self._h)
        return idx[:count.value].copy(), score[:count.value].copy()

    def bound_query_row_topn(self, topn: int):
        """`mi355rec_query_row_topn` with CALLER-OWNED result buffers bound once, as a C or C++ host would call it: returns
        call(row) -> (idx, score, count), the same two arrays every time (valid until the next call).  What a latency
        loop should use: query_row_topn() allocates and converts per call (~3 us of Python around a 46 us query)."""
        topn = int(topn)
        n_out = max(topn, 1)
        idx = np.empty(n_out, dtype=np.int64)
        score = np.empty(n_out, dtype=np.float32)
        count = ctypes.c_int(0)
        p_idx, p_score, p_count = idx.ctypes.data_as(ctypes.c_void_p), score.ctypes.data_as(ctypes.c_void_p), ctypes.byref(count)
        fn, h = self._lib.mi355rec_query_row_topn, self._h

        def call(local_row: int):
            rc = fn(h, local_row, topn, p_idx, p_score, p_count)
            if rc:
                capi.check(rc, h)
            return idx, score, count.value
        return call

    def query_topn(self, query, exclude_global: int, topn: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _np_f32(query).reshape(capi.DIM)
        n_out = max(int(topn), 1)
        idx = np.empty(n_out, dtype=np.int64)
        score = np.empty(n_out, dtype=np.float32)
        count = ctypes.c_int(0)
        capi.check(self._lib.mi355rec_query_topn(
            self._h, q.ctypes.data_as(ctypes.c_void_p), int(exclude_global), int(topn),
            idx.ctypes.data_as(ctypes.c_void_p), score.ctypes.data_as(ctypes.c_void_p),
            ctypes.byref(count)), self._h)
        return idx[:count.value].copy(), score[:count.value].copy()

    def query_batch_topn(self, queries, exclude_global, topn: int):
        q = _np_f32(queries).reshape(-1, capi.DIM)
        b = q.shape[0]
        excl = None
        if exclude_global is not None:
            excl = np.ascontiguousarray(np.asarray(exclude_global, dtype=np.int64).reshape(b))
        n_out = max(int(topn), 1)
        idx = np.empty((b, n_out), dtype=np.int64)
        score = np.empty((b, n_out), dtype=np.float32)
        counts = np.zeros(b, dtype=np.int32)
        capi.check(self._lib.mi355rec_query_batch_topn(
            self._h, q.ctypes.data_as(ctypes.c_void_p), b,
            excl.ctypes.data_as(ctypes.c_void_p) if excl is not None else None, int(topn),
            idx.ctypes.data_as(ctypes.c_void_p), score.ctypes.data_as(ctypes.c_void_p),
            counts.ctypes.data_as(ctypes.c_void_p)), self._h)
        return idx, score, counts

    # -- asynchronous device API -------------------------------------------
    @staticmethod
    def _stream_ptr(stream) -> ctypes.c_void_p:
        if stream is None:
            import torch
            stream = torch.cuda.current_stream()
        if hasattr(stream, "cuda_stream"):
            return ctypes.c_void_p(stream.cuda_stream)
        return ctypes.c_void_p(int(stream))

    def enqueue_row_keys(self, local_row: int, topn: int, out_keys, stream=None) -> None:
        capi.check(self._lib.mi355rec_enqueue_row_keys(
            self._h, int(local_row), int(topn), ctypes.c_void_p(out_keys.data_ptr()),
            self._stream_ptr(stream)), self._h)

    def enqueue_query_keys(self, query, exclude_global: int, topn: int, out_keys, stream=None) -> None:
        q = _np_f32(query).reshape(capi.DIM)
        capi.check(self._lib.mi355rec_enqueue_query_keys(
            self._h, q.ctypes.data_as(ctypes.c_void_p), int(exclude_global), int(topn),
            ctypes.c_void_p(out_keys.data_ptr()), self._stream_ptr(stream)), self._h)

    def enqueue_row_keys_streamed(self, local_row: int, topn: int, out_keys, stream=None) -> None:
        """Deferred merge: complete after the NEXT streamed call's work, or after enqueue_flush."""
        capi.check(self._lib.mi355rec_enqueue_row_keys_streamed(
            self._h, int(local_row), int(topn), ctypes.c_void_p(out_keys.data_ptr()),
            self._stream_ptr(stream)), self._h)

    def bound_enqueue_row_keys_streamed(self, topn: int, stream=None):
        """`mi355rec_enqueue_row_keys_streamed` with everything but the row and the output pointer bound once, as a C or C++
        host would call it: returns call(row, out_ptr) with out_ptr = ctypes.c_void_p(tensor.data_ptr()).  What a
        throughput loop should use: the general wrapper spends ~2 us per call on attribute look-ups around an 8 us launch,
        and a two-lane stream at 17 us per query notices (tools/lanes.cpp: 16.7 us from C++)."""
        fn, h, sp, topn = self._lib.mi355rec_enqueue_row_keys_streamed, self._h, self._stream_ptr(stream), int(topn)

        def call(row: int, out_ptr) -> None:
            rc = fn(h, row, topn, out_ptr, sp)
            if rc:
                capi.check(rc, h)
        return call

    def enqueue_query_keys_streamed(self, query, exclude_global: int, topn: int, out_keys, stream=None) -> None:
        q = _np_f32(query).reshape(capi.DIM)
        capi.check(self._lib.mi355rec_enqueue_query_keys_streamed(
            self._h, q.ctypes.data_as(ctypes.c_void_p), int(exclude_global), int(topn),
            ctypes.c_void_p(out_keys.data_ptr()), self._stream_ptr(stream)), self._h)

    def enqueue_flush(self, stream=None) -> None:
        capi.check(self._lib.mi355rec_enqueue_flush(self._h, self._stream_ptr(stream)), self._h)

    def enqueue_batch_keys(self, queries, exclude_global, topn: int, out_keys, stream=None) -> None:
        """Multi-query passes: 12 queries share one scan of the shard (topn <= 128)."""
        q = _np_f32(queries).reshape(-1, capi.DIM)
        excl = None
        if exclude_global is not None:
            excl = np.ascontiguousarray(np.asarray(exclude_global, dtype=np.int64).reshape(q.shape[0]))
        capi.check(self._lib.mi355rec_enqueue_batch_keys(
            self._h, q.ctypes.data_as(ctypes.c_void_p),
            excl.ctypes.data_as(ctypes.c_void_p) if excl is not None else None, q.shape[0], int(topn),
            ctypes.c_void_p(out_keys.data_ptr()), self._stream_ptr(stream)), self._h)

    def enqueue_batch_keys_streamed(self, queries, exclude_global, topn: int, out_keys, stream=None) -> None:
        """A stream of batches: complete after the second streamed batch call behind it, or after enqueue_flush."""
        q = _np_f32(queries).reshape(-1, capi.DIM)
        excl = None
        if exclude_global is not None:
            excl = np.ascontiguousarray(np.asarray(exclude_global, dtype=np.int64).reshape(q.shape[0]))
        capi.check(self._lib.mi355rec_enqueue_batch_keys_streamed(
            self._h, q.ctypes.data_as(ctypes.c_void_p),
            excl.ctypes.data_as(ctypes.c_void_p) if excl is not None else None, q.shape[0], int(topn),
            ctypes.c_void_p(out_keys.data_ptr()), self._stream_ptr(stream)), self._h)

    def enqueue_batch_keys_dev(self, queries_dev, exclude_dev, topn: int, out_keys, stream=None) -> None:
        """Batched matrix-core path over queries already in device memory
        (float32 [batch, 12] tensor; exclude_dev int64 [batch] tensor or None)."""
        capi.check(self._lib.mi355rec_enqueue_batch_keys_dev(
            self._h, ctypes.c_void_p(queries_dev.data_ptr()),
            ctypes.c_void_p(exclude_dev.data_ptr()) if exclude_dev is not None else None,
            int(queries_dev.shape[0]), int(topn), ctypes.c_void_p(out_keys.data_ptr()),
            self._stream_ptr(stream)), self._h)

    def batched_last_counters(self) -> dict:
        """Diagnostics of the last chunk served by the batched path (synchronises)."""
        sp, qd, mx = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
        tot = ctypes.c_int64(0)
        capi.check(self._lib.mi355rec_batched_last_counters(
            self._h, ctypes.byref(sp), ctypes.byref(qd), ctypes.byref(tot), ctypes.byref(mx)), self._h)
        return {"special_rows": sp.value, "queued_queries": qd.value, "candidates_total": tot.value,
                "candidates_max": mx.value}

    def batched_pass2_pairs(self) -> dict:
        """(tile, query block) pairs pass 2 of the last batched chunk ran its MFMAs for, out of all (synchronises)."""
        done, total = ctypes.c_int64(0), ctypes.c_int64(0)
        capi.check(self._lib.mi355rec_batched_pass2_pairs(self._h, ctypes.byref(done), ctypes.byref(total)), self._h)
        return {"pairs_done": int(done.value), "pairs_total": int(total.value)}

    def set_replica(self, mode: int) -> None:
        """capi.REPLICA_AUTO / REPLICA_OFF (fp32 rows only) / REPLICA_ON: which copy single queries scan."""
        capi.check(self._lib.mi355rec_set_replica(self._h, int(mode)), self._h)

    def replica_counters(self) -> dict:
        """Cumulative: scans over the replica and rows they sent to the exact fp32 chain (synchronises)."""
        scans, rows = ctypes.c_int64(0), ctypes.c_int64(0)
        capi.check(self._lib.mi355rec_replica_counters(self._h, ctypes.byref(scans), ctypes.byref(rows)), self._h)
        return {"scans": int(scans.value), "rescored_rows": int(rows.value)}

    def debug_handoff(self, flags: int) -> None:
        """Test hook (capi.DEBUG_HANDOFF_*): poison / drop the cross-workgroup hand-offs of the streamed scans."""
        capi.check(self._lib.mi355rec_debug_handoff(self._h, int(flags)), self._h)

    def rebuild_replica(self) -> None:
        """After overwriting a borrowed catalogue in place (synchronous)."""
        capi.check(self._lib.mi355rec_rebuild_replica(self._h), self._h)

    def set_batch_path(self, path: int) -> None:
        """capi.BATCH_AUTO / BATCH_MULTI / BATCH_MFMA (tests, A/B measurements)."""
        capi.check(self._lib.mi355rec_set_batch_path(self._h, int(path)), self._h)

    def enqueue_merge_keys(self, lists, n_lists: int, list_len: int, topn: int, out_keys,
                           out_idx=None, out_score=None, stream=None) -> None:
        capi.check(self._lib.mi355rec_enqueue_merge_keys(
            self._h, ctypes.c_void_p(lists.data_ptr()), int(n_lists), int(list_len), int(topn),
            ctypes.c_void_p(out_keys.data_ptr()),
            ctypes.c_void_p(out_idx.data_ptr()) if out_idx is not None else None,
            ctypes.c_void_p(out_score.data_ptr()) if out_score is not None else None,
            self._stream_ptr(stream)), self._h)

    def enqueue_merge_keys_batch(self, lists, n_lists: int, list_len: int, list_stride: int, query_stride: int,
                                 batch: int, topn: int, out_keys, out_idx=None, out_score=None, stream=None) -> None:
        capi.check(self._lib.mi355rec_enqueue_merge_keys_batch(
            self._h, ctypes.c_void_p(lists.data_ptr()), int(n_lists), int(list_len), int(list_stride),
            int(query_stride), int(batch), int(topn), ctypes.c_void_p(out_keys.data_ptr()),
            ctypes.c_void_p(out_idx.data_ptr()) if out_idx is not None else None,
            ctypes.c_void_p(out_score.data_ptr()) if out_score is not None else None,
            self._stream_ptr(stream)), self._h)

    def enqueue_scores(self, local_row: int, query, out_scores, stream=None) -> None:
        q = None if query is None else _np_f32(query).reshape(capi.DIM)
        capi.check(self._lib.mi355rec_enqueue_scores(
            self._h, int(local_row), q.ctypes.data_as(ctypes.c_void_p) if q is not None else None,
            ctypes.c_void_p(out_scores.data_ptr()), self._stream_ptr(stream)), self._h)

    def enqueue_stream_probe(self, sink, stream=None, which: int = capi.PROBE_FP32_ROWS) -> None:
        """The plain read-only stream over one of the handle's buffers (fp32 rows, fp16 replica, 8-bit replica)."""
        capi.check(self._lib.mi355rec_enqueue_stream_probe_of(
            self._h, int(which), ctypes.c_void_p(sink.data_ptr()), self._stream_ptr(stream)), self._h)

    def set_timing(self, enabled) -> None:
        """0/False off, 1/True every launch, k > 1 every k-th launch."""
        capi.check(self._lib.mi355rec_set_timing(self._h, int(enabled)), self._h)

    def stats(self) -> capi.Stats:
        st = capi.Stats()
        capi.check(self._lib.mi355rec_stats(self._h, ctypes.byref(st)), self._h)
        return st


class NodeEngine:
    """The catalogue on the GPUs of one node driven by ONE process (mi355rec_create_placed): what the C++
    Recommender shim uses.  `placement`: capi.PLACEMENT_SHARDED (rows split over the devices; the default),
    PLACEMENT_REPLICATED (every device holds all rows and serves whole windows of the stream).
    `devices=None` -> devices 0 .. n_devices-1, n_devices = 0 letting the library choose; a list may repeat a
    device (virtual shards / replicas on a one-GPU box)."""

    def __init__(self, feats, devices=None, n_devices: int = 0, placement: int = capi.PLACEMENT_SHARDED):
        self._lib = capi.lib()
        self._h = ctypes.c_void_p()
        arr = _np_f32(feats)
        if arr.ndim != 2 or arr.shape[1] != capi.DIM:
            raise ValueError("catalogue must be float32 [n, 12]")
        devs = None if devices is None else np.ascontiguousarray(np.asarray(devices, dtype=np.int32))
        rc = self._lib.mi355rec_create_placed(
            arr.ctypes.data_as(ctypes.c_void_p), arr.shape[0], arr.shape[1],
            devs.ctypes.data_as(ctypes.c_void_p) if devs is not None else None,
            len(devs) if devs is not None else int(n_devices), int(placement), ctypes.byref(self._h))
        if rc != capi.OK:
            raise capi.Mi355Error(rc, (self._lib.mi355rec_sharded_last_error(None) or b"").decode("utf-8", "replace"))
        self.rows = int(arr.shape[0])

    def placement(self) -> int:
        return int(self._lib.mi355rec_sharded_placement(self._h))

    def _check(self, rc: int) -> None:
        if rc != capi.OK:
            raise capi.Mi355Error(rc, (self._lib.mi355rec_sharded_last_error(self._h) or b"").decode("utf-8", "replace"))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mi355rec_sharded_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def info(self) -> dict:
        n, t = ctypes.c_int(0), ctypes.c_int(0)
        rows = ctypes.c_int64(0)
        devs = np.zeros(64, dtype=np.int32)
        per = np.zeros(64, dtype=np.int64)
        self._check(self._lib.mi355rec_sharded_info(self._h, ctypes.byref(n), ctypes.byref(t), ctypes.byref(rows),
                                                    devs.ctypes.data_as(ctypes.c_void_p), per.ctypes.data_as(ctypes.c_void_p)))
        return {"n_shards": n.value, "transport": t.value, "rows": rows.value,
                "devices": devs[: n.value].tolist(), "shard_rows": per[: n.value].tolist()}

    def set_transport(self, transport: int) -> None:
        self._check(self._lib.mi355rec_sharded_set_transport(self._h, int(transport)))

    def query_row_topn(self, global_row: int, topn: int) -> Tuple[np.ndarray, np.ndarray]:
        n_out = max(int(topn), 1)
        idx = np.empty(n_out, dtype=np.int64)
        score = np.empty(n_out, dtype=np.float32)
        count = ctypes.c_int(0)
        self._check(self._lib.mi355rec_sharded_query_row_topn(
            self._h, int(global_row), int(topn), idx.ctypes.data_as(ctypes.c_void_p),
            score.ctypes.data_as(ctypes.c_void_p), ctypes.byref(count)))
        return idx[:count.value].copy(), score[:count.value].copy()

    def query_topn(self, query, exclude_global: int, topn: int) -> Tuple[np.ndarray, np.ndarray]:
        q = _np_f32(query).reshape(capi.DIM)
        n_out = max(int(topn), 1)
        idx = np.empty(n_out, dtype=np.int64)
        score = np.empty(n_out, dtype=np.float32)
        count = ctypes.c_int(0)
        self._check(self._lib.mi355rec_sharded_query_topn(
            self._h, q.ctypes.data_as(ctypes.c_void_p), int(exclude_global), int(topn),
            idx.ctypes.data_as(ctypes.c_void_p), score.ctypes.data_as(ctypes.c_void_p), ctypes.byref(count)))
        return idx[:count.value].copy(), score[:count.value].copy()

    def query_batch_topn(self, queries, exclude_global, topn: int):
        q = _np_f32(queries).reshape(-1, capi.DIM)
        b = q.shape[0]
        excl = None
        if exclude_global is not None:
            excl = np.ascontiguousarray(np.asarray(exclude_global, dtype=np.int64).reshape(b))
        n_out = max(int(topn), 1)
        idx = np.empty((b, n_out), dtype=np.int64)
        score = np.empty((b, n_out), dtype=np.float32)
        counts = np.zeros(b, dtype=np.int32)
        self._check(self._lib.mi355rec_sharded_query_batch_topn(
            self._h, q.ctypes.data_as(ctypes.c_void_p), b,
            excl.ctypes.data_as(ctypes.c_void_p) if excl is not None else None, int(topn),
            idx.ctypes.data_as(ctypes.c_void_p), score.ctypes.data_as(ctypes.c_void_p),
            counts.ctypes.data_as(ctypes.c_void_p)))
        return idx, score, counts

    def scores_row(self, global_row: int) -> np.ndarray:
        out = np.empty(self.rows, dtype=np.float32)
        self._check(self._lib.mi355rec_sharded_scores_row(self._h, int(global_row), out.ctypes.data_as(ctypes.c_void_p)))
        return out

    # -- the stream of single queries (asynchronous; tickets) ---------------------------
    def set_window(self, window: int) -> None:
        self._check(self._lib.mi355rec_sharded_set_window(self._h, int(window)))

    def set_window_mode(self, batched: bool) -> None:
        self._check(self._lib.mi355rec_sharded_set_window_mode(self._h, 1 if batched else 0))

    def enqueue_row(self, global_row: int, topn: int) -> int:
        t = ctypes.c_int64(-1)
        self._check(self._lib.mi355rec_sharded_enqueue_row(self._h, int(global_row), int(topn), ctypes.byref(t)))
        return int(t.value)

    def enqueue_query(self, query, exclude_global: int, topn: int) -> int:
        q = _np_f32(query).reshape(capi.DIM)
        t = ctypes.c_int64(-1)
        self._check(self._lib.mi355rec_sharded_enqueue_query(self._h, q.ctypes.data_as(ctypes.c_void_p), int(exclude_global),
                                                            int(topn), ctypes.byref(t)))
        return int(t.value)

    def enqueue_flush(self) -> None:
        self._check(self._lib.mi355rec_sharded_enqueue_flush(self._h))

    def wait(self, ticket: int, topn: int) -> Tuple[np.ndarray, np.ndarray]:
        idx = np.empty(int(topn), dtype=np.int64)
        score = np.empty(int(topn), dtype=np.float32)
        count = ctypes.c_int(0)
        self._check(self._lib.mi355rec_sharded_wait(self._h, int(ticket), idx.ctypes.data_as(ctypes.c_void_p),
                                                   score.ctypes.data_as(ctypes.c_void_p), ctypes.byref(count)))
        return idx[:count.value].copy(), score[:count.value].copy()

    def stream_stats(self) -> dict:
        q, e, ns = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)
        self._check(self._lib.mi355rec_sharded_stream_stats(self._h, ctypes.byref(q), ctypes.byref(e), ctypes.byref(ns)))
        return {"queries": int(q.value), "exchanges": int(e.value), "host_ns": int(ns.value)}

    def set_timing(self, enabled) -> None:
        self._check(self._lib.mi355rec_sharded_set_timing(self._h, int(enabled)))

    def shard_stats(self, shard: int) -> capi.Stats:
        st = capi.Stats()
        self._check(self._lib.mi355rec_sharded_shard_stats(self._h, int(shard), ctypes.byref(st)))
        return st

    def set_replica(self, mode: int) -> None:
        self._check(self._lib.mi355rec_sharded_set_replica(self._h, int(mode)))

    def rccl_ranks(self) -> dict:
        """What RCCL reports about the RCCL transport's communicators (mi355rec_sharded_rccl_ranks): how many the handle holds,
        ncclCommCount of the first, and whether all of them agree (0 / 0 / False until that transport has been used)."""
        comms, ranks, agree = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        self._check(self._lib.mi355rec_sharded_rccl_ranks(self._h, ctypes.byref(comms), ctypes.byref(ranks), ctypes.byref(agree)))
        return {"communicators": int(comms.value), "ranks": int(ranks.value), "ranks_agree": bool(agree.value)}

    def rows_by_pointer(self) -> bool:
        return bool(self._lib.mi355rec_sharded_rows_by_pointer(self._h))

    def note(self) -> str:
        return (self._lib.mi355rec_sharded_note(self._h) or b"").decode("utf-8", "replace")


# ---- packed keys on the host (pure bit manipulation, mirrors kernels.hip.h) ----

def unpack_keys(keys) -> Tuple[np.ndarray, np.ndarray]:
    """(row, score) arrays from packed uint64/int64 keys; empty keys dropped."""
    k = np.asarray(keys).astype(np.uint64, copy=False).reshape(-1)
    k = k[k != 0]
    rows = (~k.astype(np.uint32)).astype(np.int64)  # low 32 bits = ~row
    hi = (k >> np.uint64(32)).astype(np.uint32)
    neg = (hi & np.uint32(0x80000000)) == 0
    bits = np.where(neg, ~hi, hi & np.uint32(0x7FFFFFFF)).astype(np.uint32)
    return rows, bits.view(np.float32)


# ---- row sharding -----------------------------------------------------------

def shard_bounds(n_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous row block of `rank`: [lo, hi).

    Balanced: the first n_rows % world_size ranks hold one row more, so a rank is
    empty only when n_rows < world_size (a ceil-division split leaves trailing
    ranks empty much earlier, e.g. 10 rows over 8 ranks).  An empty rank takes
    part in the collective with an all-zero key list (`ShardedEngine(local=None)`).
    """
    n, w, r = int(n_rows), int(world_size), int(rank)
    per, rem = divmod(n, w)
    lo = r * per + min(r, rem)
    hi = lo + per + (1 if r < rem else 0)
    return lo, hi


class ShardedEngine:
    """Row-sharded catalogue, one process per GPU, one all-gather per query.

    `local` is this rank's engine over rows [lo, hi) created with
    row_base=lo, so the keys it emits already carry GLOBAL row ids and the
    merged result does not depend on the number of ranks.  Everything is
    enqueued on the current stream; nothing synchronises the host.
    """

    def __init__(self, local, max_topn: int, group=None, device=None, always_gather: bool = False, lanes: int = 1):
        """lanes > 1: the windowed stream of single queries is dealt over that many lanes of `local` (CosineEngine.lane(): the
        same rows and replicas, own stream state), each on its own stream — a rank's launches then overlap as on one GPU."""
        import torch
        import torch.distributed as dist

        self._torch = torch
        self._dist = dist
        self.local = local
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.max_topn = int(max_topn)
        if not 1 <= self.max_topn <= capi.MAX_TOPN_FAST:
            # the merge of the gathered lists is a single launch (mi355rec_enqueue_merge_keys)
            raise ValueError(f"max_topn must be in [1, {capi.MAX_TOPN_FAST}], got {max_topn}")
        self.always_gather = bool(always_gather)  # run the collective even at world size 1
        # RCCL gathers device tensors directly.  Under gloo (CPU rehearsals, or several
        # ranks sharing one GPU in a test) the keys are staged through host memory.
        self._stage_host = str(dist.get_backend(group)).lower() == "gloo" and torch.device(
            device if device is not None else "cuda").type == "cuda"
        dev = device if device is not None else torch.device("cuda", local.device)
        self.device = dev
        self._lanes = [local]
        self._lane_streams = [None]
        if int(lanes) > 1 and hasattr(local, "lane"):
            self._lanes = [local] + [local.lane() for _ in range(int(lanes) - 1)]
            self._lane_streams = [ln.own_stream() for ln in self._lanes]
            self._lane_ev = [torch.cuda.Event() for _ in self._lanes]      # "this lane's launches so far" -> the collective's stream
            self._buf_ev = [None, None]                                    # "window buffer b has been gathered" -> the lanes
        self._w_calls = 0
        self.local_keys = torch.zeros(self.max_topn, dtype=torch.int64, device=dev)
        self.gathered = torch.zeros(self.world * self.max_topn, dtype=torch.int64, device=dev)
        self.out_keys = torch.zeros(self.max_topn, dtype=torch.int64, device=dev)
        self.out_idx = torch.full((self.max_topn,), -1, dtype=torch.int64, device=dev)
        self.out_score = torch.zeros(self.max_topn, dtype=torch.float32, device=dev)

    def _all_gather(self, gathered, local) -> None:
        if not self._stage_host:
            self._dist.all_gather_into_tensor(gathered, local, group=self.group)
            return
        host_out = self._torch.empty(gathered.shape, dtype=gathered.dtype)
        self._dist.all_gather_into_tensor(host_out, local.cpu(), group=self.group)
        gathered.copy_(host_out)

    def enqueue_query(self, query, exclude_global: int, topn: int) -> None:
        """Scan the local shard, all-gather the candidates, merge on device."""
        if topn > self.max_topn:
            raise ValueError(f"topn {topn} > max_topn {self.max_topn}")
        k = int(topn)
        local = self.local_keys[:k]
        self.local.enqueue_query_keys(query, exclude_global, k, local)
        if self.world == 1 and not self.always_gather:
            gathered = local
        else:
            gathered = self.gathered[: self.world * k]
            self._all_gather(gathered, local)
        self.local.enqueue_merge_keys(gathered, self.world, k, k, self.out_keys[:k],
                                      self.out_idx[:k], self.out_score[:k])

    # -- windowed single queries: one all-gather per `window` queries ---------------
    def enqueue_query_windowed(self, query, exclude_global: int, topn: int, window: int = 16) -> int:
        """A stream of single queries on a sharded catalogue: every query is still one
        full pass over every shard, but the exchange is amortised — the local merge of
        query k rides in the scan launch of query k + 1 (streamed C-ABI calls) and the
        per-rank key lists of `window` queries cross xGMI in ONE all-gather followed by ONE
        batched merge launch.

        The local stream is NOT drained when a window fills: a streamed query's keys are complete, in stream
        order, behind the second streamed call after it (the stream runs one call behind and its merge rides in
        the launch after that; csrc/sharded.hip keeps the same lag, kWindowLag), so a full window's all-gather
        is enqueued two queries into the NEXT window.  Returns the number of queries whose results became
        available during this call (0 or `window`): they are in `self.window_keys / window_idx / window_score`
        ([count, topn]; also appended to `self.merged_windows`, which every call resets).  `flush_window`
        closes what is left."""
        torch = self._torch
        k, w = int(topn), int(window)
        if k > self.max_topn:
            raise ValueError(f"topn {topn} > max_topn {self.max_topn}")
        self.merged_windows = []
        if getattr(self, "_w_shape", None) != (w, k):
            if getattr(self, "_w_count", 0) or getattr(self, "_w_pending", None) is not None:
                self.flush_window()
                self.merged_windows = []
            dev = self.device
            self._w_local = [torch.zeros(w * k, dtype=torch.int64, device=dev) for _ in range(2)]
            self._w_gather = [torch.zeros(self.world * w * k, dtype=torch.int64, device=dev) for _ in range(2)]
            self._w_keys = [torch.zeros(w * k, dtype=torch.int64, device=dev) for _ in range(2)]
            self._w_idx = [torch.full((w * k,), -1, dtype=torch.int64, device=dev) for _ in range(2)]
            self._w_score = [torch.zeros(w * k, dtype=torch.float32, device=dev) for _ in range(2)]
            self._w_shape = (w, k)
            if len(self._lanes) > 1:   # (the lanes' streams do not wait for the stream that just zeroed these buffers)
                torch.cuda.current_stream().synchronize()
                self._buf_ev = [None, None]
            # a real engine underneath (not a test double): the call bound once — output pointers per slot instead of a tensor
            # slice per query, the handle and the lane's stream resolved here (this loop is what every rank runs per query:
            # 15 us of Python per query made eight ranks no faster than one)
            self._w_fast = None
            if all(hasattr(ln, "_h") and hasattr(ln, "_lib") for ln in self._lanes):
                self._w_ptrs = [[ctypes.c_void_p(t.data_ptr() + s * k * 8) for s in range(w)] for t in self._w_local]
                self._w_fast = [(ln._lib.mi355rec_enqueue_query_keys_streamed, ln._h,
                                 ln._stream_ptr(ls) if ls is not None else None) for ln, ls in zip(self._lanes, self._lane_streams)]
            self._w_count = 0
            self._w_cur = 0
            self._w_pending = None
        slot, buf = self._w_count, self._w_cur
        nl = len(self._lanes)
        lane = self._w_calls % nl
        self._w_calls += 1
        if nl > 1 and slot == 0 and self._buf_ev[buf] is not None:
            for ls in self._lane_streams:   # this buffer's previous window has been gathered before a lane writes to it again
                ls.wait_event(self._buf_ev[buf])
        if self._w_fast is not None:
            fn, h, sp = self._w_fast[lane]
            q = query if (isinstance(query, np.ndarray) and query.dtype == np.float32 and query.size == capi.DIM
                          and query.flags.c_contiguous) else _np_f32(query).reshape(capi.DIM)
            rc = fn(h, ctypes.c_void_p(q.ctypes.data), int(exclude_global), k, self._w_ptrs[buf][slot],
                    sp if sp is not None else CosineEngine._stream_ptr(None))
            if rc:
                capi.check(rc, h)
        elif nl > 1:
            self._lanes[lane].enqueue_query_keys_streamed(query, exclude_global, k, self._w_local[buf][slot * k:(slot + 1) * k],
                                                          stream=self._lane_streams[lane])
        else:
            self.local.enqueue_query_keys_streamed(query, exclude_global, k, self._w_local[buf][slot * k:(slot + 1) * k])
        self._w_count += 1
        done = 0
        lag = 2 * nl   # a lane runs one call behind and its merge rides in the call after that: two calls PER LANE
        if self._w_pending is not None and self._w_count >= lag:   # the window before is complete behind this call
            done = self._merge_window(*self._w_pending)
            self._w_pending = None
        if self._w_count == w:
            if w >= lag + 1:   # (a smaller window would fill again before its predecessor is `lag` calls old)
                self._w_pending = (buf, w)
                self._w_cur = 1 - buf
                self._w_count = 0
            else:
                done = self.flush_window()
        return done

    def _merge_window(self, buf: int, cnt: int) -> int:
        """ONE all-gather + ONE batched merge for the `cnt` queries of window buffer `buf` (its local key lists are
        complete in stream order)."""
        w, k = self._w_shape
        need = cnt * k
        local = self._w_local[buf][:need]
        if len(self._lanes) > 1:   # the collective's stream waits for what every lane has been given so far
            cur = self._torch.cuda.current_stream()
            for ls, ev in zip(self._lane_streams, self._lane_ev):
                ev.record(ls)
                cur.wait_event(ev)
        if self.world == 1 and not self.always_gather:
            gathered = local
        else:
            gathered = self._w_gather[buf][: self.world * need]
            self._all_gather(gathered, local)
        self.local.enqueue_merge_keys_batch(gathered, self.world, k, need, k, cnt, k, self._w_keys[buf][:need],
                                            self._w_idx[buf][:need], self._w_score[buf][:need])
        if len(self._lanes) > 1:
            ev = self._torch.cuda.Event()
            ev.record(self._torch.cuda.current_stream())
            self._buf_ev[buf] = ev
        self.window_keys = self._w_keys[buf][:need].view(cnt, k)
        self.window_idx = self._w_idx[buf][:need].view(cnt, k)
        self.window_score = self._w_score[buf][:need].view(cnt, k)
        if not hasattr(self, "merged_windows"):
            self.merged_windows = []
        self.merged_windows.append((self.window_keys, self.window_idx, self.window_score))
        return cnt

    def flush_window(self) -> int:
        """Closes the stream of windows: drains the local pipeline, then merges the window that was still waiting
        for its lag (if any) and the open, possibly partial one — each ONE all-gather + ONE batched merge.  Returns
        the number of queries of the LAST window merged (0: nothing was outstanding); `self.merged_windows` lists
        the results of every window this call merged, oldest first."""
        self.merged_windows = []
        cnt = getattr(self, "_w_count", 0)
        pending = getattr(self, "_w_pending", None)
        if cnt == 0 and pending is None:
            return 0
        for ln, ls in zip(self._lanes, self._lane_streams):
            ln.enqueue_flush(stream=ls) if ls is not None else ln.enqueue_flush()
        done = 0
        if pending is not None:
            done = self._merge_window(*pending)
            self._w_pending = None
        if cnt:
            done = self._merge_window(self._w_cur, cnt)
            self._w_count = 0
        return done

    def enqueue_batch(self, queries, exclude_global, topn: int):
        """`batch` queries: local multi-query passes, ONE all-gather of batch*topn
        keys per rank, one merge launch (a workgroup per query).  Results in
        self.batch_keys / batch_idx / batch_score ([batch, topn])."""
        torch = self._torch
        q = np.ascontiguousarray(np.asarray(queries, dtype=np.float32).reshape(-1, 12))
        b, k = q.shape[0], int(topn)
        need = b * k
        if getattr(self, "_batch_cap", 0) < need:
            dev = self.device
            self._b_local = torch.zeros(need, dtype=torch.int64, device=dev)
            self._b_gather = torch.zeros(self.world * need, dtype=torch.int64, device=dev)
            self._b_keys = torch.zeros(need, dtype=torch.int64, device=dev)
            self._b_idx = torch.full((need,), -1, dtype=torch.int64, device=dev)
            self._b_score = torch.zeros(need, dtype=torch.float32, device=dev)
            self._batch_cap = need
        local = self._b_local[:need]
        self.local.enqueue_batch_keys(q, exclude_global, k, local)
        if self.world == 1 and not self.always_gather:
            gathered = local
        else:
            gathered = self._b_gather[: self.world * need]
            self._all_gather(gathered, local)
        self.local.enqueue_merge_keys_batch(gathered, self.world, k, need, k, b, k, self._b_keys[:need],
                                            self._b_idx[:need], self._b_score[:need])
        self.batch_keys = self._b_keys[:need].view(b, k)
        self.batch_idx = self._b_idx[:need].view(b, k)
        self.batch_score = self._b_score[:need].view(b, k)

    def query(self, query, exclude_global: int, topn: int):
        """Synchronous convenience wrapper: (rows, scores) as numpy arrays."""
        self.enqueue_query(query, exclude_global, topn)
        idx = self.out_idx[:topn].cpu().numpy()
        score = self.out_score[:topn].cpu().numpy()
        keep = idx >= 0
        return idx[keep], score[keep]
