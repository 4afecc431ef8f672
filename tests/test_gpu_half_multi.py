"""Multi-query passes over the replicas (csrc/replica_multi.hip.h, mi355::scan_half_multi_kernel):
2 ... 32 queries share ONE pass — 12 B/row through the integer matrix core with an fp16 re-check of the
candidates, or 24 B/row through the fp16 matrix core — candidates resolved in the same launch; every key
is still the exact fp32 chain on the fp32 row.
Through the C-ABI (mi355rec_query_batch_topn / _enqueue_batch_keys with MI355REC_BATCH_HALF forced, and
under AUTO where up to 32 queries on a shard with a replica take this path), against the oracle:
scores bit-exact, ids tie-aware, keys identical to the single-query path.

The hostile cases are the replica scan's own (tests/test_gpu_replica.py): value ranges of the error
bound's model, special rows and queries, ordered catalogues, duplicates and mass ties (candidate lists
that fill at every tile boundary and are resolved on the spot), tiny shards.
"""
import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches
from tests.test_batched_margin import catalogues

pytestmark = pytest.mark.gpu

HALF, AUTO, ON = 3, 0, 2


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


from tests.conftest import experiment_build


@pytest.fixture(params=["q8", "fp16"] if experiment_build() else ["fp16"])
def Engine(torch_cuda, request):
    """Every test runs once per front end of the multi-query pass: rows streamed from the fp16 replica
    (MI355REC_BATCH_HALF: the product's) or from the 8-bit replica through the integer matrix core
    (MI355REC_BATCH_Q8: an A/B route of MI355REC_EXPERIMENTS builds since round 5 — skipped on the product library)."""
    from spotify_recommender_amd.engine import CosineEngine

    class FrontEnd(CosineEngine):
        """HALF forced by a test means "the multi-query pass whatever the count": with this fixture's front end."""
        front = 4 if request.param == "q8" else 3

        def set_batch_path(self, path):
            super().set_batch_path(self.front if path == HALF else path)

    return FrontEnd


def passes_of(batch):
    """Passes over the replica one call makes: chains of 36 queries, 32 queries per pass."""
    total = 0
    while batch > 0:
        chain = min(batch, 36)
        total += (chain + 31) // 32
        batch -= chain
    return total


def check_batch(eng, f, queries, excl, topn, label, path=HALF, torch=None):
    eng.set_batch_path(path)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    idx, sc, counts = eng.query_batch_topn(queries, excl, topn)
    for b in range(len(queries)):
        want = oracle.scores(f, queries[b], threads=0)
        ex = int(excl[b]) if excl is not None else -1
        try:
            assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, ex, topn, ref_idx=oracle.topn_heap(want, ex, topn))
        except AssertionError as e:
            raise AssertionError(f"{label}: query {b} of {len(queries)}, topn {topn}: {e}") from e
    return idx, sc, counts


@pytest.mark.parametrize("n", [70_001, 300_001, 2_500_003])
def test_batches_match_the_oracle_and_the_single_query_path(Engine, torch_cuda, n):
    torch = torch_cuda
    from spotify_recommender_amd.engine import unpack_keys
    rng = np.random.default_rng(n)
    f = rng.random((n, 12), dtype=np.float32)
    f[50:60] = f[3]                                        # ties with a query
    f[n - 2] = f[3]
    with Engine(f) as eng:
        assert eng.stats().replica_bytes_per_query > 0
        before = eng.replica_counters()
        passes = 0
        for batch, topn in ((2, 100), (5, 10), (12, 100), (32, 128), (36, 1), (70, 16), (7, 100)):
            qrows = rng.integers(0, n, size=batch)
            qrows[0] = 3
            queries = f[qrows].copy()
            excl = qrows.astype(np.int64)
            if batch > 2:
                queries[2] = rng.random(12, dtype=np.float32)     # an external query, nothing excluded
                excl[2] = -1
            idx, sc, counts = check_batch(eng, f, queries, excl, topn, f"n={n}")
            passes += passes_of(batch)
            # the asynchronous entry: packed keys, identical to one single-query scan each
            keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys(queries, excl, topn, keys)
            passes += passes_of(batch)
            single = torch.zeros(topn, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            keys = keys.cpu().numpy().reshape(batch, topn)
            for b in (0, batch // 2, batch - 1):
                k_rows, k_sc = unpack_keys(keys[b])
                assert k_rows.tolist() == idx[b][:counts[b]].tolist()
                eng.enqueue_query_keys(queries[b], int(excl[b]), topn, single)
                torch.cuda.synchronize()
                assert np.array_equal(single.cpu().numpy(), keys[b]), (batch, topn, b)
        after = eng.replica_counters()
        # every pass went over the replica, and sent a few thousand rows per query to the exact chain — not all, not none
        extra_single = 3 * 7 if eng.stats().replica_active else 0   # the single-query checks scan the replica too (AUTO: from 1 M rows up)
        assert after["scans"] - before["scans"] == passes + extra_single
        per_pass = (after["rescored_rows"] - before["rescored_rows"]) / (after["scans"] - before["scans"])
        assert 1 <= per_pass < 0.05 * n * 32, per_pass


def test_up_to_32_queries_take_this_path_under_auto(Engine):
    rng = np.random.default_rng(5)
    n = 400_000
    f = rng.random((n, 12), dtype=np.float32)
    rows = rng.integers(0, n, size=32)
    with Engine(f) as eng:
        for batch in (2, 12, 16, 32):
            before = eng.replica_counters()["scans"]
            check_batch(eng, f, f[rows[:batch]], rows[:batch].astype(np.int64), 100, "auto", path=AUTO)
            assert eng.replica_counters()["scans"] == before + 1      # ONE pass over the replica
        # 33 and more queries: the two-pass matrix-core path (no replica SCAN is counted)
        before = eng.replica_counters()["scans"]
        rows13 = rng.integers(0, n, size=33)
        check_batch(eng, f, f[rows13], rows13.astype(np.int64), 100, "auto 33", path=AUTO)
        assert eng.replica_counters()["scans"] == before
        assert eng.batched_last_counters()["queued_queries"] == 0


@pytest.mark.parametrize("which", range(7))
def test_hostile_value_ranges(Engine, which):
    rng = np.random.default_rng(200 + which)
    n = 400_003
    name, f = list(catalogues(rng, n))[which]
    f = np.ascontiguousarray(f, dtype=np.float32)
    queries = np.stack([f[rng.integers(0, n)], f[rng.integers(0, n)] * np.float32(3), rng.random(12, dtype=np.float32),
                        rng.normal(0, 1, 12).astype(np.float32), np.eye(12, dtype=np.float32)[3],
                        f[rng.integers(0, n)], -f[rng.integers(0, n)]])
    with Engine(f) as eng:
        for topn in (1, 100, 128):
            check_batch(eng, f, queries, None, topn, name)


def test_special_rows_and_queries(Engine):
    """NaN / inf / denormal / zero / overflowing rows inside and outside the sampled regions, and queries the
    bound cannot be claimed for (zero, tiny, huge, NaN, inf) in the SAME pass as ordinary ones."""
    rng = np.random.default_rng(9)
    n = 600_011
    f = rng.random((n, 12), dtype=np.float32)
    regions = min(256, n // 1024)
    stride = (n // regions) & ~1
    sampled = np.array([b * stride + o for b in range(0, regions, 5) for o in (0, 1, 127, 128, 600, 1023)])
    unsampled = np.array([b * stride + o for b in range(2, regions, 7) for o in (1024, 1500, stride - 1)])
    vals = [np.nan, np.inf, -np.inf, 1e-42, 3e19, -3e19, 0.0]
    for i, r in enumerate(np.concatenate([sampled, unsampled])):
        f[r, rng.integers(0, 12)] = vals[i % len(vals)]
    f[sampled[::4] + 2] = 0.0
    f[unsampled[::3] + 1] = np.float32(1e-42)
    f[sampled[::6] + 3] = np.float32(6e-5) * rng.random((len(sampled[::6]), 12), dtype=np.float32)
    big = np.zeros(12, dtype=np.float32)
    big[:3] = (3e19, 3e19, -3e19)
    f[sampled[::9] + 4] = big
    qrows = rng.choice(n, size=4, replace=False)
    extra = np.zeros((8, 12), dtype=np.float32)
    extra[0, :3] = 1e19
    extra[1] = f[sampled[1]]
    extra[2] = f[sampled[0]]
    extra[3] = 0.0
    extra[4] = rng.random(12, dtype=np.float32) * np.float32(1e-5)
    extra[5] = rng.random(12, dtype=np.float32) * np.float32(1e-4)
    extra[6] = -rng.random(12, dtype=np.float32)
    extra[7] = rng.normal(0, 1, 12).astype(np.float32) * np.float32(1e17)
    queries = np.concatenate([f[qrows], extra])
    excl = np.concatenate([qrows.astype(np.int64), np.full(8, -1)])
    with Engine(f) as eng:
        for topn in (1, 100):
            check_batch(eng, f, queries, excl, topn, "special rows / queries")
        z = int(sampled[0] + 2)                                 # a zero row as a query: every score 0, order by row index
        check_batch(eng, f, np.stack([f[z], f[7]]), np.array([z, 7]), 50, "zero-row query")


@pytest.mark.parametrize("descending", [False, True])
def test_catalogue_ordered_by_similarity(Engine, descending):
    rng = np.random.default_rng(10)
    n = 700_003
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)
    if descending:
        t = t[::-1]
    f = np.ones((n, 12), dtype=np.float32)
    f[:, :6] = (0.1 + 0.9 * t)[:, None]
    f[:, 6:] += rng.random((n, 6), dtype=np.float32) * np.float32(1e-3)
    q = np.ones((8, 12), dtype=np.float32)
    q[:, 6:] += rng.random((8, 6), dtype=np.float32) * np.float32(0.02)
    with Engine(f) as eng:
        for topn in (1, 100):
            check_batch(eng, f, q, None, topn, f"ordered desc={descending}")


def test_duplicates_mass_ties_and_identical_catalogues(Engine):
    rng = np.random.default_rng(11)
    n = 500_009
    f = rng.random((n, 12), dtype=np.float32)
    regions = min(256, n // 1024)
    stride = (n // regions) & ~1
    qrow = int(rng.integers(0, n))
    dup = np.concatenate([np.arange(0, regions, 3) * stride + 5, rng.choice(n, 150, replace=False)])
    f[dup] = f[qrow]                                          # > topn exact duplicates of a query, many of them sampled
    v = rng.random(12, dtype=np.float32) + np.float32(0.5)
    tie_rows = rng.choice(np.setdiff1d(np.arange(n), dup), size=5000, replace=False)
    f[tie_rows] = v[None, :] * (np.float32(2.0) ** rng.integers(-3, 4, size=5000)).astype(np.float32)[:, None]
    better = rng.choice(np.setdiff1d(np.arange(n), np.concatenate([tie_rows, dup])), size=60, replace=False)
    tie_query = v + rng.random(12, dtype=np.float32) * np.float32(0.05)
    f[better] = tie_query[None, :] * (1 + rng.random((60, 12), dtype=np.float32) * np.float32(1e-3))
    queries = np.stack([tie_query, tie_query * np.float32(2), f[qrow], f[17], v])
    excl = np.array([-1, -1, qrow, 17, -1])
    with Engine(f) as eng:
        for topn in (100, 128, 7):
            check_batch(eng, f, queries, excl, topn, "mass ties / duplicates")
    # every row identical: every row ties with every other for every query, every list fills at every tile
    same = np.tile(np.linspace(0.05, 0.95, 12, dtype=np.float32), (90_000, 1))
    with Engine(same) as eng:
        eng.set_batch_path(HALF)
        idx, sc, counts = eng.query_batch_topn(same[:8], np.arange(8), 64)
        for b in range(8):
            assert idx[b].tolist() == [i for i in range(65) if i != b][:64]
    # ascending scores for every query: the local thresholds keep rising, compaction after compaction
    n = 120_000
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None]
    f = np.ones((n, 12), dtype=np.float32)
    f[:, :6] = 1.0 - 0.9 * (1.0 - t)
    queries = np.ones((32, 12), dtype=np.float32)
    queries[:, 6:] += np.linspace(0, 0.01, 32, dtype=np.float32)[:, None]
    with Engine(f) as eng:
        check_batch(eng, f, queries, None, 100, "ascending")


def test_small_shards_row_base_and_fallbacks(Engine):
    rng = np.random.default_rng(12)
    # below 65536 rows the path is not taken even when forced (no seedable sample worth a launch): same results
    for n in (3, 777, 5000, 66_000):
        f = rng.random((n, 12), dtype=np.float32)
        with Engine(f) as eng:
            rows = np.unique(np.array([0, n - 1, n // 2]))
            check_batch(eng, f, f[rows], rows.astype(np.int64), 10, f"n={n}")
    n = 300_000
    f = rng.random((n, 12), dtype=np.float32)
    lo = 123_457
    with Engine(f[lo:], row_base=lo) as eng:                  # a shard: keys carry global row ids
        eng.set_batch_path(HALF)
        q = rng.random((4, 12), dtype=np.float32)
        idx, sc, counts = eng.query_batch_topn(q, np.array([lo + 5, -1, lo, n - 1]), 100)
        for b, ex in enumerate((lo + 5, -1, lo, n - 1)):
            want = oracle.scores(f, q[b], threads=0)
            want[:lo] = -2.0
            assert_topn_matches(idx[b], sc[b], want, ex, 100)
    # replica switched off: the exact multi-query pass serves the batch
    f = rng.random((200_000, 12), dtype=np.float32)
    with Engine(f) as eng:
        eng.set_replica(1)
        before = eng.replica_counters()["scans"]
        rows = rng.integers(0, 200_000, size=6)
        check_batch(eng, f, f[rows], rows.astype(np.int64), 20, "replica off")
        assert eng.replica_counters()["scans"] == before


def test_a_stream_of_batches(Engine, torch_cuda):
    """mi355rec_enqueue_batch_keys_streamed: call k + 1 launches batch k together with the mergers of batch
    k - 1 and the seed riders of batch k + 1; the flush closes the stream.  Batches of every size (more than
    32 queries = several groups), different topn in one stream, single streamed queries in between (the two
    kinds of stream close each other), a rebuild of the replica in the middle: every key list vs the oracle."""
    torch = torch_cuda
    from spotify_recommender_amd.engine import unpack_keys
    rng = np.random.default_rng(77)
    n = 1_200_001
    f = rng.random((n, 12), dtype=np.float32)
    f[70:75] = f[9]
    plan = [(12, 100), (32, 100), (1, 10), (40, 16), (7, 128), (2, 100), (12, 100), (12, 100)]
    outs, meta = [], []
    with Engine(f) as eng:
        eng.set_batch_path(HALF)       # this fixture's front end for every batch of the stream
        for step, (batch, topn) in enumerate(plan):
            qrows = rng.integers(0, n, size=batch)
            qrows[0] = 9
            queries = f[qrows].copy()
            excl = qrows.astype(np.int64)
            if batch > 2:
                queries[1] = rng.random(12, dtype=np.float32)
                excl[1] = -1
            keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys_streamed(queries, excl, topn, keys)
            outs.append(keys)
            meta.append((queries, excl, topn))
            if step == 3:      # a streamed SINGLE query in the middle: closes the batch stream, and is closed by the next batch
                single = torch.zeros(50, dtype=torch.int64, device="cuda")
                eng.enqueue_row_keys_streamed(123, 50, single)
            if step == 5:
                eng.rebuild_replica()          # completes everything that is in flight first
        eng.enqueue_flush()
        torch.cuda.synchronize()
        for keys, (queries, excl, topn) in zip(outs, meta):
            got = keys.cpu().numpy().reshape(len(queries), topn)
            for b in range(len(queries)):
                want = oracle.scores(f, queries[b], threads=0)
                idx, sc = unpack_keys(got[b])
                assert_topn_matches(idx, sc, want, int(excl[b]), topn, ref_idx=oracle.topn_heap(want, int(excl[b]), topn))
        want = oracle.scores(f, f[123], threads=0)
        idx, sc = unpack_keys(single.cpu().numpy())
        assert_topn_matches(idx, sc, want, 123, 50, ref_idx=oracle.topn_heap(want, 123, 50))
        # a flush with nothing pending is a no-op; a stream of one batch works
        eng.enqueue_flush()
        keys = torch.zeros(3 * 20, dtype=torch.int64, device="cuda")
        eng.enqueue_batch_keys_streamed(f[[5, 6, 7]], np.array([5, 6, 7]), 20, keys)
        eng.enqueue_flush()
        torch.cuda.synchronize()
        got = keys.cpu().numpy().reshape(3, 20)
        for b, r in enumerate((5, 6, 7)):
            want = oracle.scores(f, f[r], threads=0)
            idx, sc = unpack_keys(got[b])
            assert_topn_matches(idx, sc, want, r, 20, ref_idx=oracle.topn_heap(want, r, 20))
    # a shard without a replica: the streamed entry serves the batch at once
    small = rng.random((30_000, 12), dtype=np.float32)
    with Engine(small) as eng:
        keys = torch.zeros(4 * 10, dtype=torch.int64, device="cuda")
        eng.enqueue_batch_keys_streamed(small[[1, 2, 3, 4]], np.array([1, 2, 3, 4]), 10, keys)
        eng.enqueue_flush()
        torch.cuda.synchronize()
        got = keys.cpu().numpy().reshape(4, 10)
        for b, r in enumerate((1, 2, 3, 4)):
            want = oracle.scores(small, small[r], threads=0)
            idx, sc = unpack_keys(got[b])
            assert_topn_matches(idx, sc, want, r, 10, ref_idx=oracle.topn_heap(want, r, 10))


def test_stream_of_batches_fails_safe_under_stale_hand_offs(Engine, torch_cuda):
    """The multi-query stream's hand-offs (seed riders -> last rider -> the next launch's cutoffs) under
    mi355rec_debug_handoff: stale samples of earlier batches (a perfect score under the last epochs), cutoffs of
    +1.0, dropped stores, launches without a last rider.  Wrong-epoch values count as absent, so every key list must
    still be the one the plain (not streamed) path returns.  (tests/test_gpu_replica.py has the single-query twin.)"""
    from spotify_recommender_amd import capi as _capi
    if not _capi.has_test_hooks():
        pytest.skip("mi355rec_debug_handoff is not in the product library: this test runs against libmi355rec_testhooks.so "
                    "(tests/test_gpu_testhooks.py, a child process)")
    from spotify_recommender_amd import capi
    torch = torch_cuda
    rng = np.random.default_rng(99)
    n = 1_500_000
    f = rng.random((n, 12), dtype=np.float32)
    P, D, L = capi.DEBUG_HANDOFF_POISON, capi.DEBUG_HANDOFF_DROP_STORES, capi.DEBUG_HANDOFF_NO_LAST_RIDER
    hooks = {2: P, 4: D, 6: L, 8: P | D, 10: P | L, 12: P | D | L, 13: P, 14: P | D | L}
    topn = 50
    with Engine(f) as eng:
        eng.set_batch_path(HALF)
        batches = [rng.integers(0, n, size=int(b)) for b in rng.choice([2, 12, 12, 32], size=18)]
        want = []
        for qrows in batches:       # the reference: each batch alone, not streamed
            keys = torch.zeros(len(qrows) * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys(f[qrows], qrows.astype(np.int64), topn, keys)
            want.append(keys)
        torch.cuda.synchronize()
        got = []
        for step, qrows in enumerate(batches):
            if step in hooks:
                eng.debug_handoff(hooks[step])
            keys = torch.zeros(len(qrows) * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys_streamed(f[qrows], qrows.astype(np.int64), topn, keys)
            got.append(keys)
            if step == 9:
                eng.enqueue_flush()
        eng.debug_handoff(P)
        eng.enqueue_flush()
        torch.cuda.synchronize()
        bad = [i for i in range(len(batches)) if not torch.equal(want[i], got[i])]
        assert not bad, bad
        # a batch on its own hands its cutoffs from the sample launch's last workgroups to the pass: the same hooks
        for step, hook in enumerate((P, D, L, P | D | L, P | L)):
            eng.debug_handoff(hook)
            keys = torch.zeros(len(batches[step]) * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys(f[batches[step]], batches[step].astype(np.int64), topn, keys)
            torch.cuda.synchronize()
            assert torch.equal(want[step], keys), (step, hook)
        # and against the oracle for one batch that ran right behind a poisoned, last-rider-less launch
        check = got[13].cpu().numpy().reshape(len(batches[13]), topn)
        from spotify_recommender_amd.engine import unpack_keys
        for b, r in enumerate(batches[13][:4]):
            s = oracle.scores(f, f[r], threads=0)
            idx, sc = unpack_keys(check[b])
            assert_topn_matches(idx, sc, s, int(r), topn, ref_idx=oracle.topn_heap(s, int(r), topn))
