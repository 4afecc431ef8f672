"""The multi-GPU code path with REAL kernels and REAL processes: 2 and 3 ranks,
one process each, all on the one GPU of the test box (RCCL refuses several ranks
per device, so the collective runs over gloo with host staging — everything
else is the production path: per-rank engines over row shards with global row
ids, ShardedEngine, the device merge).  Results must equal the oracle on the
whole catalogue on every rank."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, n_rows, result_dir):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        from spotify_recommender_amd import CosineEngine, ShardedEngine, shard_bounds
        torch.cuda.set_device(0)
        rng = np.random.default_rng(777)
        f = rng.random((n_rows, 12), dtype=np.float32)
        f[n_rows - 5:] = f[11]                      # duplicates of a query in the last shard
        lo, hi = shard_bounds(n_rows, world, rank)
        shard = torch.from_numpy(f[lo:hi]).to("cuda:0")
        eng = CosineEngine(shard, row_base=lo)
        sh = ShardedEngine(eng, max_topn=100)
        out = {}
        for q in (11, n_rows // 2, n_rows - 1):
            for topn in (1, 10, 100):
                idx, sc = sh.query(f[q], q, topn)
                want = oracle.scores(f, f[q])
                ci, cs = oracle.topn_canonical(want, q, topn)
                assert idx.tolist() == ci.tolist(), (rank, q, topn)
                assert np.array_equal(sc, cs + np.float32(0))
                out[f"{q}_{topn}"] = idx
        qrows = np.array([11, 5, n_rows - 1, n_rows // 3, 99, 100, 101, 7, 8, 9, 10, 12, 13], dtype=np.int64)
        sh.enqueue_batch(f[qrows], qrows, 50)
        got = sh.batch_idx.cpu().numpy()
        for b, q in enumerate(qrows):
            want = oracle.scores(f, f[q])
            ci, _ = oracle.topn_canonical(want, int(q), 50)
            assert got[b].tolist() == ci.tolist(), (rank, "batch", q)
        # the windowed stream of single queries over two lanes per rank (ShardedEngine(lanes=2)): every rank makes the same calls
        sh2 = ShardedEngine(eng, max_topn=100, lanes=2)
        stream_rows = [11, n_rows // 2, n_rows - 1, 5, 99, n_rows // 3, 7, 8, 9, 10, 12, 13, 100]
        got = []
        for q in stream_rows:
            if sh2.enqueue_query_windowed(f[q], q, 20, window=5):
                torch.cuda.synchronize()
                got.extend(sh2.window_idx.cpu().numpy())
        sh2.flush_window()
        torch.cuda.synchronize()
        for _keys, widx, _sc in sh2.merged_windows:
            got.extend(widx.cpu().numpy())
        assert len(got) == len(stream_rows), (rank, len(got))
        for q, idx in zip(stream_rows, got):
            ci, _ = oracle.topn_canonical(oracle.scores(f, f[q]), int(q), 20)
            assert idx.tolist() == ci.tolist(), (rank, "lanes", q)
        np.savez(Path(result_dir) / f"rank{rank}.npz", **out)
        eng.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_as_processes_on_one_gpu(tmp_path, world):
    n_rows = 600_011
    mp.spawn(_worker, args=(world, _free_port(), n_rows, str(tmp_path)), nprocs=world, join=True)
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    for key in ranks[0].files:
        for r in ranks[1:]:
            assert ranks[0][key].tolist() == r[key].tolist()
