"""The row-sharded engine behind the C-ABI (mi355rec_create_sharded*, csrc/sharded.hip):
one process, one stream per shard, peer stores or one RCCL all-gather, merge on the
first device.  On a one-GPU box the orchestration is exercised with VIRTUAL shards
(the same device listed several times); the RCCL transport runs a real
ncclAllGather at one rank.  Everything is compared with the oracle."""
import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Node():
    import torch
    assert torch.cuda.is_available()
    from spotify_recommender_amd.engine import NodeEngine
    return NodeEngine


def check(node, f, q, excl, topn):
    idx, sc = node.query_topn(q, excl, topn)
    want = oracle.scores(f, q)
    assert_topn_matches(idx, sc, want, excl, topn, ref_idx=oracle.topn_heap(want, excl, topn))
    return idx, sc


def test_one_device_equals_the_single_engine(Node):
    from spotify_recommender_amd import capi
    from spotify_recommender_amd.engine import CosineEngine
    rng = np.random.default_rng(1)
    f = rng.random((300_000, 12), dtype=np.float32)
    with Node(f, n_devices=0) as node, CosineEngine(f) as eng:
        info = node.info()
        assert info["n_shards"] >= 1 and info["rows"] == 300_000 and sum(info["shard_rows"]) == 300_000
        for row in (0, 12345, 299_999):
            a_idx, a_sc = node.query_row_topn(row, 100)
            b_idx, b_sc = eng.query_row_topn(row, 100)
            assert a_idx.tolist() == b_idx.tolist() and np.array_equal(a_sc.view(np.uint32), b_sc.view(np.uint32))
            want = oracle.scores(f, f[row])
            assert_topn_matches(a_idx, a_sc, want, row, 100, ref_idx=oracle.topn_heap(want, row, 100))
        assert np.array_equal(node.scores_row(777).view(np.uint32), oracle.scores(f, f[777]).view(np.uint32))


@pytest.mark.parametrize("shards", [2, 3, 8])
def test_virtual_shards_match_the_oracle(Node, shards):
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(shards)
    n = 400_003
    f = rng.random((n, 12), dtype=np.float32)
    f[n - 5] = f[7]                            # a duplicate of a query row in the last shard
    with Node(f, devices=[0] * shards) as node:
        info = node.info()
        assert info["n_shards"] == shards and info["transport"] == capi.TRANSPORT_PEER
        assert max(info["shard_rows"]) - min(info["shard_rows"]) <= 1
        for row in (7, n // 2, n - 1):
            idx, sc = node.query_row_topn(row, 100)
            want = oracle.scores(f, f[row])
            assert_topn_matches(idx, sc, want, row, 100, ref_idx=oracle.topn_heap(want, row, 100))
        check(node, f, rng.random(12, dtype=np.float32), -1, 1)
        # a batch: per-shard batched passes, ONE exchange, one batched merge launch
        qrows = rng.integers(0, n, size=40)
        idx, sc, counts = node.query_batch_topn(f[qrows], qrows, 50)
        for b, row in enumerate(qrows):
            want = oracle.scores(f, f[row])
            assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, int(row), 50)
        # the full score vector is assembled shard by shard
        assert np.array_equal(node.scores_row(n - 5).view(np.uint32), oracle.scores(f, f[n - 5]).view(np.uint32))
        # topn above the single-launch merge limit: per-shard rounds + host key merge
        idx, sc = node.query_row_topn(11, 3000)
        want = oracle.scores(f, f[11])
        assert_topn_matches(idx, sc, want, 11, 3000)
        # the RCCL transport refuses several shards on one device, loudly
        with pytest.raises(capi.Mi355Error):
            node.set_transport(capi.TRANSPORT_RCCL)
            node.query_row_topn(7, 10)


def test_more_shards_than_rows_and_bad_arguments(Node):
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(5)
    f = rng.random((5, 12), dtype=np.float32)
    with Node(f, devices=[0] * 8) as node:
        assert node.info()["shard_rows"] == [1, 1, 1, 1, 1, 0, 0, 0]
        idx, sc = node.query_row_topn(2, 10)       # topn > rows - 1
        want = oracle.scores(f, f[2])
        assert_topn_matches(idx, sc, want, 2, 10)
        with pytest.raises(capi.Mi355Error):
            node.query_row_topn(5, 3)
        with pytest.raises(capi.Mi355Error):
            node.query_row_topn(0, 0)
    with pytest.raises(capi.Mi355Error):
        Node(f, devices=[0, 99])
    with pytest.raises(capi.Mi355Error):
        Node(f, n_devices=4096)


def test_rccl_transport_issues_a_real_all_gather(Node):
    """One shard per device is what RCCL wants; with one visible device that is one
    rank: ncclCommInitAll + a grouped ncclAllGather really run on the box."""
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(9)
    f = rng.random((250_000, 12), dtype=np.float32)
    with Node(f, n_devices=1) as node:
        node.set_transport(capi.TRANSPORT_RCCL)
        assert node.info()["transport"] == capi.TRANSPORT_RCCL
        for row in (3, 200_000):
            idx, sc = node.query_row_topn(row, 100)
            want = oracle.scores(f, f[row])
            assert_topn_matches(idx, sc, want, row, 100, ref_idx=oracle.topn_heap(want, row, 100))
        qrows = rng.integers(0, 250_000, size=20)
        idx, sc, counts = node.query_batch_topn(f[qrows], qrows, 10)
        for b, row in enumerate(qrows):
            want = oracle.scores(f, f[row])
            assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, int(row), 10)


# ---- the asynchronous stream on the sharded handle (tickets, windows) ------------------------

@pytest.mark.parametrize("shards,window,n", [(1, 16, 300_007), (3, 4, 300_007), (8, 16, 400_003), (2, 1, 100_001),
                                             (2, 5, 2_400_001)])   # the last: shards that scan their fp16 replica
def test_streamed_queries_on_the_sharded_handle(Node, shards, window, n):
    """enqueue_row / enqueue_query only enqueue (one streamed launch per shard), the key lists of a
    WINDOW of queries share one exchange + one batched merge; results by ticket, compared with the
    oracle.  Partial windows, waits that have to close the window themselves, vectors and rows mixed."""
    rng = np.random.default_rng(100 + shards)
    f = rng.random((n, 12), dtype=np.float32)
    f[n - 3] = f[11]                                  # a duplicate of a query row in the last shard
    topn = 50
    with Node(f, devices=[0] * shards) as node:
        assert node.rows_by_pointer() and node.note() == ""
        node.set_window(window)
        qrows = [11, n - 1, n // 2] + rng.integers(0, n, size=2 * window + 3).tolist()
        for chunk in range(0, len(qrows), 2 * window):        # at most 2 windows in flight: all results still kept
            part = qrows[chunk:chunk + 2 * window]
            tickets = [node.enqueue_row(r, topn) for r in part]
            vec = rng.random(12, dtype=np.float32)
            tv = node.enqueue_query(vec, 5, topn)
            assert tickets == sorted(tickets) and tv > tickets[-1]
            for t, r in zip(tickets, part):                    # the first wait closes whatever is open
                idx, sc = node.wait(t, topn)
                want = oracle.scores(f, f[r])
                assert_topn_matches(idx, sc, want, r, topn, ref_idx=oracle.topn_heap(want, r, topn))
            idx, sc = node.wait(tv, topn)
            want = oracle.scores(f, vec)
            assert_topn_matches(idx, sc, want, 5, topn)
        st = node.stream_stats()
        assert st["queries"] == len(qrows) + (len(qrows) + 2 * window - 1) // (2 * window) and st["exchanges"] >= 1
        if window >= 2 and n // shards >= 65536:
            # a window went to every shard as ONE batched call: at most one pass over the replica per 32 queries
            scans = node.shard_stats(0).replica_bytes_per_query
            assert scans > 0
        # the synchronous calls still work in between, and give the same answer as the stream
        a_idx, a_sc = node.query_row_topn(11, topn)
        t = node.enqueue_row(11, topn)
        b_idx, b_sc = node.wait(t, topn)
        assert a_idx.tolist() == b_idx.tolist() and np.array_equal(a_sc.view(np.uint32), b_sc.view(np.uint32))
        # both window modes (a window as ONE batched call per shard / one streamed launch per query): same results
        for batched in (False, True):
            node.set_window_mode(batched)
            part = qrows[:window + 2]
            tickets = [node.enqueue_row(r, topn) for r in part]
            for t, r in zip(tickets, part):
                idx, sc = node.wait(t, topn)
                want = oracle.scores(f, f[r])
                assert_topn_matches(idx, sc, want, r, topn, ref_idx=oracle.topn_heap(want, r, topn))


def test_stream_ring_and_bad_tickets(Node):
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(21)
    n = 120_000
    f = rng.random((n, 12), dtype=np.float32)
    with Node(f, devices=[0, 0]) as node:
        node.set_window(2)
        with pytest.raises(capi.Mi355Error):
            node.wait(0, 10)                                    # nothing was ever enqueued
        tickets = [node.enqueue_row(int(r), 10) for r in range(0, 24)]     # 12 windows of 2: a ring of 4 is kept
        node.enqueue_flush()
        with pytest.raises(capi.Mi355Error, match="no longer kept"):
            node.wait(tickets[0], 10)
        for r in (20, 21, 22, 23):
            idx, sc = node.wait(tickets[r], 10)
            want = oracle.scores(f, f[r])
            assert_topn_matches(idx, sc, want, r, 10, ref_idx=oracle.topn_heap(want, r, 10))
        with pytest.raises(capi.Mi355Error):
            node.wait(10_000, 10)
        with pytest.raises(capi.Mi355Error):
            node.enqueue_row(n, 10)
        with pytest.raises(capi.Mi355Error):
            node.enqueue_row(0, 2000)                           # streamed queries: topn <= 1024
        with pytest.raises(capi.Mi355Error):
            node.set_window(0)
        # a change of topn re-makes the buffers; tickets keep growing
        t = node.enqueue_row(7, 33)
        assert t > tickets[-1]
        idx, sc = node.wait(t, 33)
        want = oracle.scores(f, f[7])
        assert_topn_matches(idx, sc, want, 7, 33, ref_idx=oracle.topn_heap(want, 7, 33))


def test_stream_with_more_shards_than_rows(Node):
    rng = np.random.default_rng(22)
    f = rng.random((5, 12), dtype=np.float32)
    with Node(f, devices=[0] * 8) as node:                      # three shards are empty: they answer with empty lists
        node.set_window(3)
        tickets = [node.enqueue_row(r, 10) for r in range(5)]
        for r, t in enumerate(tickets):
            idx, sc = node.wait(t, 10)
            want = oracle.scores(f, f[r])
            assert_topn_matches(idx, sc, want, r, 10)


def test_stream_over_the_rccl_transport(Node):
    """One shard per device (one rank on this box): the window's keys go through a real grouped
    ncclAllGather on the shard's stream, then the batched merge."""
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(23)
    n = 200_000
    f = rng.random((n, 12), dtype=np.float32)
    with Node(f, n_devices=1) as node:
        node.set_transport(capi.TRANSPORT_RCCL)
        node.set_window(4)
        rows = rng.integers(0, n, size=10).tolist()
        tickets = [node.enqueue_row(r, 20) for r in rows]
        node.enqueue_flush()
        for r, t in list(zip(rows, tickets))[-8:]:
            idx, sc = node.wait(t, 20)
            want = oracle.scores(f, f[r])
            assert_topn_matches(idx, sc, want, r, 20, ref_idx=oracle.topn_heap(want, r, 20))
        # what RCCL itself says about the communicators the exchange just ran on (mi355rec_sharded_rccl_ranks: ncclCommCount /
        # ncclCommUserRank) — the figure bench.py --gpus N puts in its line as `transport.rccl.ranks`
        assert node.rccl_ranks() == {"communicators": 1, "ranks": 1, "ranks_agree": True}
    with Node(f, n_devices=1) as node:        # ... and before that transport has been used: nothing to report
        assert node.rccl_ranks() == {"communicators": 0, "ranks": 0, "ranks_agree": False}


# ---- placement: the size-aware default and the replicated mode ----------------------------

def test_the_default_number_of_shards_follows_the_size_of_the_catalogue():
    """n_devices = 0 no longer means "every visible GPU whatever N is": a shard keeps at least 4 M rows."""
    from spotify_recommender_amd import capi
    L = capi.lib()
    assert L.mi355rec_auto_shards(114_000, 8) == 1            # BASELINE configs[0]: one device, no exchange
    assert L.mi355rec_auto_shards(1_000_000, 8) == 1
    assert L.mi355rec_auto_shards(7_999_999, 8) == 1
    assert L.mi355rec_auto_shards(10_000_000, 8) == 2
    assert L.mi355rec_auto_shards(32_000_000, 8) == 8
    assert L.mi355rec_auto_shards(100_000_000, 8) == 8        # configs[4]
    assert L.mi355rec_auto_shards(100_000_000, 4) == 4 and L.mi355rec_auto_shards(100_000_000, 1) == 1
    assert L.mi355rec_auto_shards(5, 8) == 1 and L.mi355rec_auto_shards(10_000_000, 0) == 0


@pytest.mark.parametrize("replicas,window,n", [(1, 16, 300_007), (3, 4, 300_007), (8, 16, 400_003), (2, 1, 100_001),
                                               (2, 5, 2_400_001)])
def test_replicated_placement_matches_the_oracle(Node, replicas, window, n):
    """MI355REC_PLACEMENT_REPLICATED with VIRTUAL replicas (the same device listed several times): every replica
    holds all rows, whole windows of the stream go to the replicas in turn and nothing is exchanged.  Same
    protocol as the sharded stream's test: tickets, partial windows, waits that close the window, vectors and rows
    mixed, both window modes, synchronous calls in between (they take the replicas in turn as well)."""
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(300 + replicas)
    f = rng.random((n, 12), dtype=np.float32)
    f[n - 3] = f[11]
    topn = 50
    with Node(f, devices=[0] * replicas, placement=capi.PLACEMENT_REPLICATED) as node:
        info = node.info()
        assert node.placement() == capi.PLACEMENT_REPLICATED
        assert info["n_shards"] == replicas and info["shard_rows"] == [n] * replicas and info["rows"] == n
        node.set_window(window)
        qrows = [11, n - 1, n // 2] + rng.integers(0, n, size=3 * window + 3).tolist()
        for chunk in range(0, len(qrows), 2 * window):        # at most 2 windows in flight: all results still kept
            part = qrows[chunk:chunk + 2 * window]
            tickets = [node.enqueue_row(r, topn) for r in part]
            vec = rng.random(12, dtype=np.float32)
            tv = node.enqueue_query(vec, 5, topn)
            assert tickets == sorted(tickets) and tv > tickets[-1]
            for t, r in zip(tickets, part):
                idx, sc = node.wait(t, topn)
                want = oracle.scores(f, f[r])
                assert_topn_matches(idx, sc, want, r, topn, ref_idx=oracle.topn_heap(want, r, topn))
            idx, sc = node.wait(tv, topn)
            want = oracle.scores(f, vec)
            assert_topn_matches(idx, sc, want, 5, topn)
        # synchronous calls: each goes to the next replica; all give the stream's answer
        t = node.enqueue_row(11, topn)
        b_idx, b_sc = node.wait(t, topn)
        for _ in range(replicas + 1):
            a_idx, a_sc = node.query_row_topn(11, topn)
            assert a_idx.tolist() == b_idx.tolist() and np.array_equal(a_sc.view(np.uint32), b_sc.view(np.uint32))
        qb = rng.integers(0, n, size=20)
        idx, sc, counts = node.query_batch_topn(f[qb], qb, 30)
        for b, row in enumerate(qb):
            want = oracle.scores(f, f[row])
            assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, int(row), 30)
        assert np.array_equal(node.scores_row(n - 3).view(np.uint32), oracle.scores(f, f[n - 3]).view(np.uint32))
        idx, sc = node.query_row_topn(11, 3000)                # above the single-launch merge limit
        assert_topn_matches(idx, sc, oracle.scores(f, f[11]), 11, 3000)
        for batched in (False, True):
            node.set_window_mode(batched)
            part = qrows[:2 * window + 1]
            tickets = [node.enqueue_row(r, topn) for r in part]
            for t, r in zip(tickets, part):
                idx, sc = node.wait(t, topn)
                want = oracle.scores(f, f[r])
                assert_topn_matches(idx, sc, want, r, topn, ref_idx=oracle.topn_heap(want, r, topn))
        node.set_transport(capi.TRANSPORT_RCCL)                # accepted and ignored: replicas exchange nothing
        t = node.enqueue_row(7, topn)
        idx, sc = node.wait(t, topn)
        assert_topn_matches(idx, sc, oracle.scores(f, f[7]), 7, topn)


def test_replicated_stream_ring_and_bad_tickets(Node):
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(31)
    n = 120_000
    f = rng.random((n, 12), dtype=np.float32)
    with Node(f, devices=[0, 0, 0], placement=capi.PLACEMENT_REPLICATED) as node:
        node.set_window(2)
        tickets = [node.enqueue_row(int(r), 10) for r in range(0, 24)]     # 12 windows of 2 over 3 replicas: a ring of 4 is kept
        node.enqueue_flush()
        with pytest.raises(capi.Mi355Error, match="no longer kept"):
            node.wait(tickets[0], 10)
        for r in (20, 21, 22, 23):
            idx, sc = node.wait(tickets[r], 10)
            want = oracle.scores(f, f[r])
            assert_topn_matches(idx, sc, want, r, 10, ref_idx=oracle.topn_heap(want, r, 10))
        with pytest.raises(capi.Mi355Error):
            node.enqueue_row(n, 10)
        t = node.enqueue_row(7, 33)                            # a change of topn re-makes the buffers
        idx, sc = node.wait(t, 33)
        assert_topn_matches(idx, sc, oracle.scores(f, f[7]), 7, 33)
    with pytest.raises(capi.Mi355Error):
        Node(f, devices=[0], placement=7)
