"""The single-query scans over the replicas — 8-bit (csrc/replica_q8.hip.h, the default) and fp16
(csrc/replica.hip.h, MI355REC_REPLICA_FP16) — against the oracle and against the fp32 scan, through
the C-ABI.  Every test runs once per replica.

The replica only rules rows OUT; every key that leaves the kernel is computed from the
fp32 row by the exact chain, so the bar is the usual one: scores bit-exact, ids identical
(tie-aware).  The diagnostics (rows sent to the exact chain) are asserted as well, so a
build that silently re-scored everything — or nothing — would fail.

The launch-wide cutoff comes from a sample: up to 256 regions of 1024 rows spaced
(n / regions) & ~1 rows apart (fp16), of 2048 rows spaced (n / regions) & ~3 apart (8-bit);
"in the sample" below means rows placed there.
"""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches
from tests.test_batched_margin import catalogues

pytestmark = pytest.mark.gpu

ON, OFF = 2, 1     # ON is rebound per test by replica_kind: 2 = ON (8-bit replica), 3 = FP16


from tests.conftest import experiment_build

# (single queries over the fp16 replica are an A/B route of MI355REC_EXPERIMENTS builds since round 5: the product library
# runs every test over the 8-bit replica only)
KINDS = [2, 3] if experiment_build() else [2]


@pytest.fixture(autouse=True, params=KINDS, ids=["q8", "fp16"][:len(KINDS)])
def replica_kind(request):
    global ON
    ON = request.param
    yield request.param
    ON = 2


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def Engine(torch_cuda):
    from spotify_recommender_amd.engine import CosineEngine
    return CosineEngine


def sample_rows(n):
    if ON == 3:
        regions = min(256, n // 1024)
        return regions, (n // regions) & ~1
    regions = min(256, n // 2048)
    return regions, (n // regions) & ~3


def check_queries(eng, f, queries, excl, topns, label, rows=None):
    """queries[i] (vector) or rows[i] (catalogue row as the query) through the replica scan vs the oracle."""
    for i in range(len(excl)):
        q = f[rows[i]] if rows is not None else queries[i]
        want = oracle.scores(f, np.ascontiguousarray(q, dtype=np.float32), threads=0)
        ex = int(excl[i])
        for topn in topns:
            if rows is not None:
                idx, sc = eng.query_row_topn(int(rows[i]), topn)
            else:
                idx, sc = eng.query_topn(q, ex, topn)
            try:
                assert_topn_matches(idx, sc, want, ex, topn, ref_idx=oracle.topn_heap(want, ex, topn))
            except AssertionError as e:
                raise AssertionError(f"{label}: query {i} topn {topn}: {e}") from e


def test_uniform_catalogue_and_the_rescored_share(Engine):
    rng = np.random.default_rng(7)
    n = 2_500_003                      # AUTO takes the replica from 1 M rows; odd: the last pair is half empty
    f = rng.random((n, 12), dtype=np.float32)
    rows = rng.integers(0, n, size=12)
    rows[0], rows[1] = n - 1, 0
    with Engine(f) as eng:
        if ON == 3:
            eng.set_replica(ON)
        st = eng.stats()
        assert st.replica_active == 1 and st.replica_bytes_per_query == (n + 1) // 2 * 48
        assert st.replica_single_row_bytes == (24 if ON == 3 else 12)
        assert st.replica_single_bytes_per_query == ((n + 1) // 2 * 48 if ON == 3 else (n + 3) // 4 * 48)
        before = eng.replica_counters()
        check_queries(eng, f, None, rows, (1, 10, 100, 1000), "uniform 2.5 M", rows=rows)
        after = eng.replica_counters()
        scans = after["scans"] - before["scans"]
        assert scans == 12 * 4
        per_scan = (after["rescored_rows"] - before["rescored_rows"]) / scans
        # the fast path really is one: a few thousand of 2.5 M rows go to the exact chain, not all and not none
        assert 1 <= per_scan < 0.02 * n, per_scan
        # the same queries over the fp32 rows: identical keys
        import torch
        k_on = torch.zeros((len(rows), 100), dtype=torch.int64, device="cuda")
        k_off = torch.zeros_like(k_on)
        for i, r in enumerate(rows):
            eng.enqueue_row_keys(int(r), 100, k_on[i])
        eng.set_replica(OFF)
        assert eng.stats().replica_active == 0
        for i, r in enumerate(rows):
            eng.enqueue_row_keys(int(r), 100, k_off[i])
        torch.cuda.synchronize()
        assert torch.equal(k_on, k_off)
        assert eng.replica_counters()["scans"] == after["scans"] + len(rows)


def test_streamed_queries_over_the_replica(Engine, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(8)
    n = 900_001
    f = rng.random((n, 12), dtype=np.float32)
    rows = rng.integers(0, n, size=9)
    vecs = rng.random((4, 12), dtype=np.float32)
    with Engine(f) as eng:
        eng.set_replica(ON)
        keys = torch.zeros((13, 100), dtype=torch.int64, device="cuda")
        for i, r in enumerate(rows):
            eng.enqueue_row_keys_streamed(int(r), 100, keys[i])
        for j, v in enumerate(vecs):
            eng.enqueue_query_keys_streamed(v, -1, 100, keys[9 + j])
        eng.enqueue_flush()
        torch.cuda.synchronize()
        got = keys.cpu().numpy().view(np.uint64)
        for i in range(13):
            q = f[rows[i]] if i < 9 else vecs[i - 9]
            ex = int(rows[i]) if i < 9 else -1
            want = oracle.scores(f, q, threads=0)
            idx = (~got[i] & np.uint64(0xffffffff)).astype(np.int64)
            assert_topn_matches(idx, None, want, ex, 100, ref_idx=oracle.topn_heap(want, ex, 100))
        # alternating with fp32-row queries in one stream: the riding merge takes either kind of list
        eng.set_replica(OFF)
        eng.enqueue_row_keys_streamed(int(rows[0]), 100, keys[0])
        eng.set_replica(ON)
        eng.enqueue_row_keys_streamed(int(rows[1]), 100, keys[1])
        eng.set_replica(OFF)
        eng.enqueue_row_keys_streamed(int(rows[2]), 100, keys[2])
        eng.enqueue_flush()
        torch.cuda.synchronize()
        again = keys.cpu().numpy().view(np.uint64)
        assert np.array_equal(again[:3], got[:3])


@pytest.mark.parametrize("which", range(7))
def test_hostile_value_ranges(Engine, which):
    """The catalogues the error bound is checked on in tests/test_batched_margin.py (rounding
    midpoints, subnormal tails after normalisation, rows at the validity edge, huge rows ...)."""
    rng = np.random.default_rng(100 + which)
    n = 400_003
    name, f = list(catalogues(rng, n))[which]
    f = np.ascontiguousarray(f, dtype=np.float32)
    queries = np.stack([f[rng.integers(0, n)], f[rng.integers(0, n)] * np.float32(3), rng.random(12, dtype=np.float32),
                        rng.normal(0, 1, 12).astype(np.float32), np.eye(12, dtype=np.float32)[3]])
    with Engine(f) as eng:
        eng.set_replica(ON)
        check_queries(eng, f, queries, np.full(len(queries), -1), (1, 100, 1000), name)


def test_special_rows_and_queries(Engine):
    """NaN / inf / denormal / zero / overflowing rows inside and outside the sampled regions, and
    queries the bound cannot be claimed for (zero, tiny, huge, NaN, inf): exact chain only."""
    rng = np.random.default_rng(9)
    n = 600_011
    f = rng.random((n, 12), dtype=np.float32)
    regions, stride = sample_rows(n)
    sampled = np.array([b * stride + o for b in range(0, regions, 5) for o in (0, 1, 127, 128, 600, 1023)])
    R = 1024 if ON == 3 else 2048
    unsampled = np.array([b * stride + o for b in range(2, regions, 7) for o in (R, R + 476, stride - 1)])
    vals = [np.nan, np.inf, -np.inf, 1e-42, 3e19, -3e19, 0.0]
    for i, r in enumerate(np.concatenate([sampled, unsampled])):
        f[r, rng.integers(0, 12)] = vals[i % len(vals)]
    f[sampled[::4] + 2] = 0.0                        # exactly-zero rows
    f[unsampled[::3] + 1] = np.float32(1e-42)        # rows whose squares underflow
    f[sampled[::6] + 3] = np.float32(6e-5) * rng.random((len(sampled[::6]), 12), dtype=np.float32)   # around the validity edge
    big = np.zeros(12, dtype=np.float32)
    big[:3] = (3e19, 3e19, -3e19)
    f[sampled[::9] + 4] = big
    qrows = rng.choice(n, size=6, replace=False)
    queries = f[qrows].copy()
    excl = qrows.astype(np.int64)
    extra = np.zeros((8, 12), dtype=np.float32)
    extra[0, :3] = 1e19                              # exact = inf / inf = NaN -> clamped to 1.0 for the overflow rows
    extra[1] = f[sampled[1]]                          # a component is inf
    extra[2] = f[sampled[0]]                          # a component is NaN
    extra[3] = 0.0
    extra[4] = rng.random(12, dtype=np.float32) * np.float32(1e-5)    # below the validity edge
    extra[5] = rng.random(12, dtype=np.float32) * np.float32(1e-4)    # straddles it
    extra[6] = -rng.random(12, dtype=np.float32)      # every cosine negative
    extra[7] = rng.normal(0, 1, 12).astype(np.float32) * np.float32(1e17)
    queries = np.concatenate([queries, extra])
    excl = np.concatenate([excl, np.full(8, -1)])
    with Engine(f) as eng:
        eng.set_replica(ON)
        check_queries(eng, f, queries, excl, (1, 100), "special rows / queries")
        # a zero row as the query (row index API): every score is 0, the order is by row index
        z = int(sampled[0] + 2)
        idx, sc = eng.query_row_topn(z, 50)
        want = oracle.scores(f, f[z], threads=0)
        assert_topn_matches(idx, sc, want, z, 50, ref_idx=oracle.topn_heap(want, z, 50))


@pytest.mark.parametrize("descending", [False, True])
def test_catalogue_ordered_by_similarity(Engine, descending):
    """The sample sees the far end of an ordered catalogue only through its spread regions."""
    rng = np.random.default_rng(10)
    n = 700_003
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)
    if descending:
        t = t[::-1]
    f = np.ones((n, 12), dtype=np.float32)
    f[:, :6] = (0.1 + 0.9 * t)[:, None]
    f[:, 6:] += rng.random((n, 6), dtype=np.float32) * np.float32(1e-3)
    q = np.ones((4, 12), dtype=np.float32)
    q[:, 6:] += rng.random((4, 6), dtype=np.float32) * np.float32(0.02)
    with Engine(f) as eng:
        eng.set_replica(ON)
        check_queries(eng, f, q, np.full(4, -1), (1, 100, 1000), f"ordered desc={descending}")


def test_duplicates_and_mass_ties_at_the_nth_score(Engine):
    rng = np.random.default_rng(11)
    n = 500_009
    f = rng.random((n, 12), dtype=np.float32)
    regions, stride = sample_rows(n)
    qrow = int(rng.integers(0, n))
    dup = np.concatenate([np.arange(0, regions, 3) * stride + 5, rng.choice(n, 150, replace=False)])
    f[dup] = f[qrow]                                  # > topn exact duplicates of the query, many of them sampled
    v = rng.random(12, dtype=np.float32) + np.float32(0.5)
    tie_rows = rng.choice(np.setdiff1d(np.arange(n), dup), size=5000, replace=False)
    f[tie_rows] = v[None, :] * (np.float32(2.0) ** rng.integers(-3, 4, size=5000)).astype(np.float32)[:, None]
    better = rng.choice(np.setdiff1d(np.arange(n), np.concatenate([tie_rows, dup])), size=60, replace=False)
    tie_query = v + rng.random(12, dtype=np.float32) * np.float32(0.05)
    f[better] = tie_query[None, :] * (1 + rng.random((60, 12), dtype=np.float32) * np.float32(1e-3))
    s = oracle.scores(f, tie_query, threads=0)
    top = np.sort(s)[::-1]
    assert top[99] == top[100], "construction: the 100th score must sit inside the tie run"
    with Engine(f) as eng:
        eng.set_replica(ON)
        check_queries(eng, f, np.stack([tie_query, tie_query * np.float32(2)]), np.array([-1, -1]), (100, 1000), "mass ties")
        check_queries(eng, f, None, np.array([qrow]), (1, 100, 300), "duplicates of the query", rows=np.array([qrow]))


@pytest.mark.parametrize("clusters", [3, 40])
def test_clustered_catalogue_where_a_replica_rules_little_out(Engine, torch_cuda, clusters):
    """Rows shaped like min-max normalised audio features in a few tight clusters (discrete key / mode / genre columns,
    spread 0.01, 2 % exact duplicates): every row of the query's cluster lies within the pre-filters' margins of the
    cutoff, so a third (3 clusters) or a fortieth of the catalogue goes to the exact chain — candidate buffers fill,
    compactions run under pressure, a tile holds many candidates instead of one.  Same keys as the oracle, lone and
    streamed; and the diagnostics show the pressure was there."""
    torch = torch_cuda
    rng = np.random.default_rng(100 + clusters)
    n = 1_300_003
    centres = rng.random((clusters, 12), dtype=np.float32)
    centres[:, 2] = rng.integers(0, 12, clusters).astype(np.float32) / np.float32(11)
    centres[:, 4] = rng.integers(0, 2, clusters).astype(np.float32)
    centres[:, 11] = rng.integers(0, 114, clusters).astype(np.float32) / np.float32(113)
    which = rng.integers(0, clusters, n)
    noise = (rng.standard_normal((n, 12)) * 0.01).astype(np.float32)
    noise[:, [2, 4, 11]] = 0
    f = np.clip(centres[which] + noise, 0, 1).astype(np.float32)
    dup = rng.integers(0, n, n // 50)
    f[dup] = f[rng.integers(0, n, n // 50)]
    rows = rng.integers(0, n, size=5)
    with Engine(f) as eng:
        eng.set_replica(ON)
        before = eng.replica_counters()
        check_queries(eng, f, None, rows, (10, 100, 1000), f"{clusters} clusters", rows=rows)
        after = eng.replica_counters()
        per_scan = (after["rescored_rows"] - before["rescored_rows"]) / (after["scans"] - before["scans"])
        assert per_scan > 0.5 * n / clusters, per_scan     # the whole cluster, or most of it, went to the exact chain
        ring = torch.zeros((len(rows), 100), dtype=torch.int64, device="cuda")
        for i, r in enumerate(rows):
            eng.enqueue_row_keys_streamed(int(r), 100, ring[i])
        eng.enqueue_flush()
        torch.cuda.synchronize()
        from spotify_recommender_amd.engine import unpack_keys
        for i, r in enumerate(rows):
            want = oracle.scores(f, f[int(r)], threads=0)
            idx, sc = unpack_keys(ring[i].cpu().numpy())
            assert_topn_matches(idx, sc, want, int(r), 100, ref_idx=oracle.topn_heap(want, int(r), 100))


def test_small_and_ragged_shards(Engine):
    """Fewer rows than one sampled region, fewer than topn, one row, row_base shards."""
    rng = np.random.default_rng(12)
    for n in (1, 2, 3, 777, 1023, 1024, 1025, 2049, 5000):
        f = rng.random((n, 12), dtype=np.float32)
        with Engine(f) as eng:
            eng.set_replica(ON)
            rows = np.unique(np.array([0, n - 1, n // 2]))
            check_queries(eng, f, None, rows, (1, 10, 100), f"n={n}", rows=rows)
    n = 300_000
    f = rng.random((n, 12), dtype=np.float32)
    lo = 123_457
    with Engine(f[lo:], row_base=lo) as eng:          # a shard: keys carry global row ids
        eng.set_replica(ON)
        q = rng.random(12, dtype=np.float32)
        idx, sc = eng.query_topn(q, lo + 5, 100)
        want = oracle.scores(f, q, threads=0)
        want[:lo] = -2.0                               # rows of other shards cannot appear
        assert_topn_matches(idx, sc, want, lo + 5, 100)


def test_rebuild_after_overwriting_a_borrowed_matrix(Engine, torch_cuda):
    torch = torch_cuda
    rng = np.random.default_rng(13)
    n = 300_001
    f0 = rng.random((n, 12), dtype=np.float32)
    f1 = rng.random((n, 12), dtype=np.float32)
    t = torch.from_numpy(f0).cuda()
    with Engine(t) as eng:
        eng.set_replica(ON)
        check_queries(eng, f0, None, np.array([17]), (100,), "before", rows=np.array([17]))
        t.copy_(torch.from_numpy(f1))
        torch.cuda.synchronize()
        eng.rebuild_replica()
        check_queries(eng, f1, None, np.array([17]), (100,), "after the rebuild", rows=np.array([17]))


def test_handle_without_a_replica(Engine):
    """MI355REC_CREATE_NO_REPLICA (a create flag; earlier rounds read an environment variable): 48 B per row resident
    instead of 84, every query over the fp32 rows."""
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(14)
    f = rng.random((1_200_000, 12), dtype=np.float32)      # a size at which a default handle WOULD build and scan replicas
    with Engine(f, flags=capi.CREATE_NO_REPLICA) as eng:
        st = eng.stats()
        assert st.replica_bytes_per_query == 0 and st.replica_active == 0 and st.device_bytes_per_row == 48
        with pytest.raises(RuntimeError, match="without a replica"):
            eng.set_replica(ON)
        idx, sc = eng.query_row_topn(3, 10)
        want = oracle.scores(f, f[3], threads=0)
        assert_topn_matches(idx, sc, want, 3, 10)
        assert eng.replica_counters() == {"scans": 0, "rescored_rows": 0}
        st = eng.stats()
        assert st.route_fp32 == 1 and st.route_q8 + st.route_q8_lone + st.route_fp16 == 0
    with Engine(f) as eng:
        assert eng.stats().device_bytes_per_row == 84
    with pytest.raises(capi.Mi355Error):
        Engine(f, flags=64)


def test_replica_queries_replay_from_a_hip_graph(Engine, torch_cuda):
    """Seed + scan + merge allocate nothing and never synchronise, so single queries over the
    replica can be captured into a hipGraph (torch's CUDAGraph on a side stream) and replayed —
    here with the same query rows over new catalogue bytes (borrowed matrix overwritten in place,
    replica rebuilt)."""
    torch = torch_cuda
    rng = np.random.default_rng(15)
    n, topn = 500_000, 50
    f = rng.random((n, 12), dtype=np.float32)
    t = torch.from_numpy(f).cuda()
    side = torch.cuda.Stream()
    rows = [5, 123_456, n - 1]
    keys = torch.zeros((len(rows), topn), dtype=torch.int64, device="cuda")
    with Engine(t) as eng:
        eng.set_replica(ON)
        with torch.cuda.stream(side):
            for i, r in enumerate(rows):
                eng.enqueue_row_keys(r, topn, keys[i], stream=side)     # warm-up on the capture stream
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for i, r in enumerate(rows):
                eng.enqueue_row_keys(r, topn, keys[i], stream=side)
        for trial in range(2):
            if trial:
                f = np.random.default_rng(16).random((n, 12), dtype=np.float32)
                t.copy_(torch.from_numpy(f))
                torch.cuda.synchronize()
                eng.rebuild_replica()
            keys.zero_()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            got = keys.cpu().numpy().view(np.uint64)
            for i, r in enumerate(rows):
                want = oracle.scores(f, f[r], threads=0)
                idx = (~got[i] & np.uint64(0xffffffff)).astype(np.int64)
                assert_topn_matches(idx, None, want, r, topn, ref_idx=oracle.topn_heap(want, r, topn))


def test_sharded_node_with_replica_sized_shards():
    """mi355rec_create_sharded over two VIRTUAL shards of > 2 M rows each: every shard's single
    queries take the replica scan under AUTO, the peer-store merge sees the same keys."""
    from spotify_recommender_amd.engine import NodeEngine
    rng = np.random.default_rng(17)
    n = 4_300_001
    f = rng.random((n, 12), dtype=np.float32)
    with NodeEngine(f, devices=[0, 0]) as node:
        assert min(node.info()["shard_rows"]) >= 2 * 1024 * 1024
        for row in (3, n // 2 + 1, n - 1):
            idx, sc = node.query_row_topn(row, 100)
            want = oracle.scores(f, f[row], threads=0)
            assert_topn_matches(idx, sc, want, row, 100, ref_idx=oracle.topn_heap(want, row, 100))
        q = rng.random(12, dtype=np.float32)
        idx, sc = node.query_topn(q, -1, 10)
        want = oracle.scores(f, q, threads=0)
        assert_topn_matches(idx, sc, want, -1, 10, ref_idx=oracle.topn_heap(want, -1, 10))


@pytest.mark.parametrize("n", [777, 5000, 40_000, 700_001, 4_000_001])
def test_streams_run_one_call_behind(Engine, torch_cuda, n):
    """Over the replica a streamed query is LAUNCHED by the next streamed call (whose sample rides in
    that launch) or by the flush: streams of one query, mixed row / vector queries with different
    topn, synchronous queries in between, a flush in the middle, shards too small to spare seed
    riders (5000 rows) or to have a sample at all (777 rows), and one large enough (4 M rows) for the
    8-bit scan's ticketed tiles and for the cutoff its last seed rider leaves for the next launch."""
    torch = torch_cuda
    rng = np.random.default_rng(n)
    f = rng.random((n, 12), dtype=np.float32)
    with Engine(f) as eng:
        eng.set_replica(ON)

        def verify(keys, q, ex, topn):
            got = keys.cpu().numpy().view(np.uint64)
            want = oracle.scores(f, q, threads=0)
            cnt = min(topn, n - (1 if ex >= 0 else 0))
            idx = (~got[:cnt] & np.uint64(0xffffffff)).astype(np.int64)
            assert_topn_matches(idx, None, want, ex, topn, ref_idx=oracle.topn_heap(want, ex, topn))
            assert not got[cnt:].any()

        # a stream of ONE query
        k1 = torch.zeros(100, dtype=torch.int64, device="cuda")
        eng.enqueue_row_keys_streamed(3, 100, k1)
        eng.enqueue_flush()
        torch.cuda.synchronize()
        verify(k1, f[3], 3, 100)
        eng.enqueue_flush()                                   # nothing pending: no-op
        # mixed stream, a synchronous query and a flush in the middle
        plan = [("row", int(rng.integers(0, n)), 10), ("vec", rng.random(12, dtype=np.float32), 100),
                ("row", n - 1, 1000), ("row", 0, 1), ("vec", rng.random(12, dtype=np.float32), 64),
                ("row", int(rng.integers(0, n)), 100), ("row", int(rng.integers(0, n)), 100)]
        outs = [torch.zeros(t, dtype=torch.int64, device="cuda") for _, _, t in plan]
        for i, (kind, q, topn) in enumerate(plan):
            if kind == "row":
                eng.enqueue_row_keys_streamed(q, topn, outs[i])
            else:
                eng.enqueue_query_keys_streamed(q, -1, topn, outs[i])
            if i == 2:
                idx, sc = eng.query_row_topn(5 % n, 20)      # synchronous, own scratch: the stream is untouched
                want = oracle.scores(f, f[5 % n], threads=0)
                assert_topn_matches(idx, sc, want, 5 % n, 20, ref_idx=oracle.topn_heap(want, 5 % n, 20))
            if i == 4:
                eng.enqueue_flush()
        eng.enqueue_flush()
        torch.cuda.synchronize()
        for i, (kind, q, topn) in enumerate(plan):
            if kind == "row":
                verify(outs[i], f[q], q, topn)
            else:
                verify(outs[i], q, -1, topn)


def test_replica_built_on_demand_under_a_running_stream(Engine, torch_cuda):
    """A shard below 65536 rows is created WITHOUT a replica (no AUTO path would read it).  Streamed
    queries over the fp32 rows, then set_replica(ON) — which builds the replica and everything keyed on
    it (the sample buffers of streamed queries included) — then more streamed queries, a rebuild in the
    middle of the stream (it completes the stashed and the pending query first), a flush: every key
    list against the oracle.  (the lazily allocated scratch must be all-or-nothing.)"""
    torch = torch_cuda
    from spotify_recommender_amd.engine import unpack_keys
    rng = np.random.default_rng(31)
    n, topn = 40_000, 20
    f = rng.random((n, 12), dtype=np.float32)
    rows = rng.integers(0, n, size=14).tolist()
    keys = torch.zeros((len(rows), topn), dtype=torch.int64, device="cuda")
    with Engine(f) as eng:
        st = eng.stats()
        assert st.replica_bytes_per_query == 0 and st.replica_active == 0
        for i in range(0, 4):
            eng.enqueue_row_keys_streamed(rows[i], topn, keys[i])           # fp32 rows, merge riding
        eng.set_replica(ON)                                                   # builds it now (flushes the stream first)
        st = eng.stats()
        assert st.replica_bytes_per_query == (n + 1) // 2 * 48 and st.replica_active == 1
        for i in range(4, 9):
            eng.enqueue_row_keys_streamed(rows[i], topn, keys[i])           # replica scan, one call behind
        eng.rebuild_replica()                                                 # completes the stashed + pending query first
        for i in range(9, 14):
            eng.enqueue_query_keys_streamed(f[rows[i]], rows[i], topn, keys[i])
        eng.enqueue_flush()
        torch.cuda.synchronize()
        got = keys.cpu().numpy()
        for i, r in enumerate(rows):
            want = oracle.scores(f, f[r], threads=0)
            idx, sc = unpack_keys(got[i])
            assert_topn_matches(idx, sc, want, r, topn, ref_idx=oracle.topn_heap(want, r, topn))
        assert eng.replica_counters()["scans"] == 10


@pytest.mark.parametrize("ordered", [False, True])
def test_long_stream_over_a_large_shard(Engine, torch_cuda, ordered):
    """60 streamed queries over 5 M rows (8-bit replica: tiles handed out by ticket, every workgroup a list,
    cutoff from the previous launch's last seed rider), uniform and sorted by similarity to the queries'
    neighbourhood (every best row in the last tiles): each key list against the oracle, and every row
    scanned exactly once (a tile taken twice would show as a duplicate key, one dropped as a missing one)."""
    torch = torch_cuda
    rng = np.random.default_rng(77 + ordered)
    n = 5_000_003
    f = rng.random((n, 12), dtype=np.float32)
    if ordered:
        t = np.sort(rng.random(n)).astype(np.float32)
        f[:, :6] = (0.1 + 0.9 * t)[:, None]
        f[:, 6:] = 1.0 + rng.random((n, 6), dtype=np.float32) * np.float32(1e-3)
    qrows = rng.integers(0, n, size=60)
    qrows[:4] = (0, n - 1, n - 2049, 2048)
    topns = [int(t) for t in rng.choice([1, 10, 100, 1000], size=60)]
    with Engine(f) as eng:
        eng.set_replica(ON)
        outs = [torch.zeros(t, dtype=torch.int64, device="cuda") for t in topns]
        for i, r in enumerate(qrows):
            eng.enqueue_row_keys_streamed(int(r), topns[i], outs[i])
            if i == 30:
                eng.enqueue_flush()
        eng.enqueue_flush()
        torch.cuda.synchronize()
        for i, r in enumerate(qrows):
            got = outs[i].cpu().numpy().view(np.uint64)
            assert len(np.unique(got)) == len(got), f"query {i}: duplicate keys"
            want = oracle.scores(f, f[r], threads=0)
            idx = (~got & np.uint64(0xffffffff)).astype(np.int64)
            assert_topn_matches(idx, None, want, int(r), topns[i], ref_idx=oracle.topn_heap(want, int(r), topns[i]))
        # lone synchronous queries: over the 8-bit replica of a shard this size the scan launch merges its own
        # lists and raises the completion word itself (one launch after the sample)
        before = eng.stats().lone_fused_queries
        for r, topn in ((int(qrows[0]), 1), (int(qrows[5]), 10), (int(qrows[6]), 100), (int(qrows[1]), 1000), (int(qrows[7]), 1024)):
            idx, sc = eng.query_row_topn(r, topn)
            want = oracle.scores(f, f[r], threads=0)
            assert_topn_matches(idx, sc, want, r, topn, ref_idx=oracle.topn_heap(want, r, topn))
        assert eng.stats().lone_fused_queries - before == (5 if ON == 2 else 0)


def test_two_thousand_streamed_queries_key_for_key(Engine, torch_cuda):
    """The hand-offs inside a launch (last seed rider -> the next launch's cutoff; sample values written through and
    read past the L2) under many back-to-back launches: 2000 streamed queries with mixed topn over 4 M rows, restarted
    every few hundred, every key list compared on the device with the same query over the fp32 rows (which use none
    of those hand-offs).  A lost or stale cutoff would show as a missing key.  (tools/soak.py is the long version.)"""
    torch = torch_cuda
    rng = np.random.default_rng(2024 + ON)
    n = 4_000_000
    f = torch.rand((n, 12), generator=torch.Generator(device="cuda").manual_seed(5), device="cuda", dtype=torch.float32)
    rows = rng.integers(0, n, size=2000)
    topns = rng.choice([1, 10, 100, 100, 500, 1000], size=2000)
    with Engine(f) as eng:
        runs = {}
        for mode in (OFF, ON):
            eng.set_replica(mode)
            outs = []
            for i in range(len(rows)):
                k = torch.zeros(int(topns[i]), dtype=torch.int64, device="cuda")
                eng.enqueue_row_keys_streamed(int(rows[i]), int(topns[i]), k)
                outs.append(k)
                if i % 331 == 330:
                    eng.enqueue_flush()
            eng.enqueue_flush()
            torch.cuda.synchronize()
            runs[mode] = outs
        bad = [i for i in range(len(rows)) if not torch.equal(runs[OFF][i], runs[ON][i])]
        assert not bad, (len(bad), bad[:5])


def test_hand_offs_fail_safe_under_stale_values(Engine, torch_cuda):
    """VERDICT r3 item 3(a) / ADVICE r3 (high).  Inside a streamed launch the seed riders hand their sample to the rider
    that finishes last, and that one hands the next launch its cutoff (8-bit replica) or its bound (fp32 rows, round 5) —
    outside the stream order; the sample launch of a query alone over the fp32 rows ends the same way.
    mi355rec_debug_handoff makes a reader find what it would find if those stores had NOT landed: the most hostile values
    an earlier query could have left (a perfect score, a cutoff of +1.0 that rules out every row) under the last queries'
    epochs — the neighbourhood's slot included — sample stores that never happen, a launch in which no rider is the last.
    A value under the wrong epoch must count as absent (the bound only gets LOWER): every key list must still equal the
    ORACLE's, over both kinds of rows, and over the replica the rows sent to the exact chain must go UP — proof that the
    stale values were met and refused."""
    from spotify_recommender_amd import capi as _capi
    if not _capi.has_test_hooks():
        pytest.skip("mi355rec_debug_handoff is not in the product library: this test runs against libmi355rec_testhooks.so "
                    "(tests/test_gpu_testhooks.py, a child process)")
    if ON == 3:
        pytest.skip("the fp16 single-query scan hands its sample from launch to launch in stream order only")
    from spotify_recommender_amd import capi
    from spotify_recommender_amd.engine import unpack_keys
    torch = torch_cuda
    rng = np.random.default_rng(4242)
    n = 4_000_000
    f = torch.rand((n, 12), generator=torch.Generator(device="cuda").manual_seed(9), device="cuda", dtype=torch.float32)
    host = f.cpu().numpy()
    rows = rng.integers(0, n, size=72)
    topns = rng.choice([10, 100, 100, 300], size=72)
    P, D, L = capi.DEBUG_HANDOFF_POISON, capi.DEBUG_HANDOFF_DROP_STORES, capi.DEBUG_HANDOFF_NO_LAST_RIDER
    hooks = {7: P, 13: D, 19: L, 25: P | D, 31: P | L, 37: D | L, 43: P | D | L, 44: P, 45: P | D | L, 60: P | D | L}
    want = []
    for i in range(len(rows)):
        ci, _ = oracle.topn_canonical(oracle.scores(host, host[rows[i]], threads=0), int(rows[i]), int(topns[i]))
        want.append(ci.tolist())

    def stream(eng, with_hooks):
        outs = []
        for i in range(len(rows)):
            if with_hooks and i in hooks:
                eng.debug_handoff(hooks[i])
            k = torch.zeros(int(topns[i]), dtype=torch.int64, device="cuda")
            eng.enqueue_row_keys_streamed(int(rows[i]), int(topns[i]), k)
            outs.append(k)
            if i == 50:
                eng.enqueue_flush()
        if with_hooks:
            eng.debug_handoff(P)       # ... and with the last query still stashed
        eng.enqueue_flush()
        torch.cuda.synchronize()
        return [unpack_keys(k.cpu().numpy())[0].tolist() for k in outs]

    with Engine(f) as eng:
        for mode in (OFF, ON):
            eng.set_replica(mode)
            c0 = eng.replica_counters()["rescored_rows"]
            clean = stream(eng, False)
            c1 = eng.replica_counters()["rescored_rows"]
            hostile = stream(eng, True)
            c2 = eng.replica_counters()["rescored_rows"]
            bad = [i for i in range(len(rows)) if clean[i] != want[i]]
            assert not bad, (mode, "clean", len(bad), bad[:8])
            bad = [i for i in range(len(rows)) if hostile[i] != want[i]]
            assert not bad, (mode, "hostile", len(bad), bad[:8])
            if mode == ON:   # refused values cost candidates: far more rows took the exact chain than in the clean stream
                assert c2 - c1 > 3 * (c1 - c0), (c1 - c0, c2 - c1)
            # queries alone (the sample launch's last workgroup selects; over the replica the scan's last workgroup merges:
            # arrival counters that are never reset) after all that, and right behind a poisoning
            for r, topn in ((int(rows[0]), 100), (int(rows[1]), 10), (int(rows[2]), 1000)):
                eng.debug_handoff(P | L if topn == 10 else P)
                idx, sc = eng.query_row_topn(r, topn)
                ci, _ = oracle.topn_canonical(oracle.scores(host, host[r], threads=0), r, topn)
                assert idx.tolist() == ci.tolist(), (mode, r, topn)
