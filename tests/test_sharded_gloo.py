"""The N>1 host path (row sharding + ONE all-gather + merge) on CPU with gloo,
world_size 2 and 3.  The local scan and the merge are the HIP kernels in the
product; here a test double built on the oracle stands in for them so that the
orchestration in ShardedEngine (shard bounds, global ids, the collective's
shapes, rank-independence of the result) is exercised without a GPU."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def np_pack_keys(scores, rows):
    s = (np.asarray(scores, np.float32) + np.float32(0)).view(np.uint32).astype(np.uint64)
    neg = (s & np.uint64(0x80000000)) != 0
    ordered = np.where(neg, (~s) & np.uint64(0xFFFFFFFF), s | np.uint64(0x80000000))
    low = (~np.asarray(rows, np.uint64)) & np.uint64(0xFFFFFFFF)
    return (ordered << np.uint64(32)) | low


class OracleShard:
    """Test double with CosineEngine's enqueue_* surface, CPU tensors."""

    def __init__(self, feats, row_base):
        from oracle import oracle
        self.oracle = oracle
        self.f = feats
        self.row_base = row_base
        self.device = 0

    def enqueue_query_keys(self, query, exclude_global, topn, out_keys, stream=None):
        s = self.oracle.scores(self.f, query)
        idx, sc = self.oracle.topn_canonical(s, exclude_global - self.row_base, topn)
        keys = np_pack_keys(sc, idx.astype(np.int64) + self.row_base)
        out = np.zeros(topn, np.uint64)
        out[: len(keys)] = keys
        out_keys.copy_(torch.from_numpy(out.view(np.int64)))

    # the streamed calls only differ in WHEN the device merge runs: same results
    def enqueue_query_keys_streamed(self, query, exclude_global, topn, out_keys, stream=None):
        self.enqueue_query_keys(query, exclude_global, topn, out_keys)

    def enqueue_flush(self, stream=None):
        pass

    def enqueue_batch_keys(self, queries, exclude_global, topn, out_keys, stream=None):
        q = np.asarray(queries, np.float32).reshape(-1, 12)
        for b in range(q.shape[0]):
            self.enqueue_query_keys(q[b], int(exclude_global[b]), topn, out_keys[b * topn:(b + 1) * topn])

    def enqueue_merge_keys_batch(self, lists, n_lists, list_len, list_stride, query_stride, batch, topn,
                                 out_keys, out_idx=None, out_score=None, stream=None):
        flat = lists.numpy().view(np.uint64)
        for b in range(batch):
            rows = np.concatenate([flat[b * query_stride + l * list_stride: b * query_stride + l * list_stride + list_len]
                                   for l in range(n_lists)])
            view = torch.from_numpy(rows.view(np.int64).copy())
            self.enqueue_merge_keys(view, n_lists, list_len, topn, out_keys[b * topn:(b + 1) * topn],
                                    None if out_idx is None else out_idx[b * topn:(b + 1) * topn],
                                    None if out_score is None else out_score[b * topn:(b + 1) * topn])

    def enqueue_merge_keys(self, lists, n_lists, list_len, topn, out_keys, out_idx=None,
                           out_score=None, stream=None):
        k = np.sort(lists.numpy().view(np.uint64)[: n_lists * list_len])[::-1][:topn].copy()
        out_keys.copy_(torch.from_numpy(k.view(np.int64)))
        from spotify_recommender_amd.engine import unpack_keys
        rows, scores = unpack_keys(k)
        if out_idx is not None:
            out_idx.fill_(-1)
            out_idx[: len(rows)] = torch.from_numpy(rows)
        if out_score is not None:
            out_score.zero_()
            out_score[: len(rows)] = torch.from_numpy(scores.copy())


def _worker(rank, world, port, n_rows, result_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from spotify_recommender_amd.engine import ShardedEngine, shard_bounds
        rng = np.random.default_rng(1234)
        f = rng.random((n_rows, 12), dtype=np.float32)
        f[n_rows - 1] = f[3]                     # duplicate of a query in the last shard
        lo, hi = shard_bounds(n_rows, world, rank)
        shard = OracleShard(f[lo:hi], lo)
        eng = ShardedEngine(shard, max_topn=64, device=torch.device("cpu"))
        out = {}
        for q in (3, n_rows // 2, n_rows - 1):
            for topn in (1, 10, 64):
                idx, sc = eng.query(f[q], q, topn)
                out[f"{q}_{topn}"] = idx
                want = oracle.scores(f, f[q])
                ci, cs = oracle.topn_canonical(want, q, topn)
                assert idx.tolist() == ci.tolist(), (rank, q, topn)
                assert np.array_equal(sc, cs + np.float32(0))
        # batched path: one all-gather for the whole batch
        qrows = [3, n_rows // 2, n_rows - 1, 17, 4242]
        eng.enqueue_batch(f[qrows], np.array(qrows), 10)
        for b, q in enumerate(qrows):
            want = oracle.scores(f, f[q])
            ci, _ = oracle.topn_canonical(want, q, 10)
            assert eng.batch_idx[b].numpy().tolist() == ci.tolist(), (rank, q)
            out[f"batch_{q}"] = eng.batch_idx[b].numpy()
        # windowed single queries: one all-gather per window of 4, a partial last window
        qrows = [3, n_rows // 2, n_rows - 1, 17, 4242, 99]
        got = []
        # (a full window is merged two queries into the next one: the local stream is not drained in between)
        done = []
        for q in qrows:
            n_done = eng.enqueue_query_windowed(f[q], q, 10, window=4)
            done.append(n_done)
            if n_done:
                got.extend(eng.window_idx.numpy().copy())
        assert done == [0, 0, 0, 0, 0, 4], done
        assert eng.flush_window() == 2
        got.extend(eng.window_idx.numpy().copy())
        assert eng.flush_window() == 0
        # a flush with a full window still waiting for its lag merges both, oldest first
        for q in qrows[:5]:
            assert eng.enqueue_query_windowed(f[q], q, 10, window=4) == 0
        assert eng.flush_window() == 1 and [int(x[1].shape[0]) for x in eng.merged_windows] == [4, 1]
        both = np.concatenate([x[1].numpy() for x in eng.merged_windows])
        assert np.array_equal(both, np.array(got[:5]))
        for q, idx in zip(qrows, got):
            ci, _ = oracle.topn_canonical(oracle.scores(f, f[q]), q, 10)
            assert idx.tolist() == ci.tolist(), (rank, q)
            out[f"window_{q}"] = idx
        # what an N > 1 bench line says about its process group (bench.py's `group` object): every rank
        # gets the same report, the ranks are COUNTED by an all-reduce, and every rank's kernel figures are in it
        from spotify_recommender_amd.benchlegs import rank_group_report
        rep = rank_group_report(dist, torch, rank, world, torch.device("cpu"), f"cpu-rank{rank}", f"uuid-{rank}",
                                kernel_ms=0.01 * (rank + 1), alg_bytes=(hi - lo) * 48)
        assert rep["backend"] == "gloo" and rep["ranks"] == world and rep["ranks_counted"] == world
        assert rep["launcher_world_size"] == world and rep["distinct_devices"] == world and len(rep["devices"]) == world
        assert [g["rank"] for g in rep["per_gpu"]] == list(range(world))
        for g in rep["per_gpu"]:
            assert g["avg_kernel_ms"] == round(0.01 * (g["rank"] + 1), 5) and g["frac"] > 0 and g["algorithmic_bytes_per_launch"] > 0
        assert sum(g["algorithmic_bytes_per_launch"] for g in rep["per_gpu"]) == n_rows * 48     # the shards cover the catalogue
        out["group_ranks_counted"] = np.array([rep["ranks_counted"]])
        np.savez(Path(result_dir) / f"rank{rank}.npz", **out)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_query_matches_oracle_on_every_rank(tmp_path, world):
    n_rows = 10_007  # ragged: the last shard is shorter
    mp.spawn(_worker, args=(world, _free_port(), n_rows, str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for key in ranks[0].files:
        for r in ranks[1:]:
            assert ranks[0][key].tolist() == r[key].tolist()   # every rank holds the merged result
