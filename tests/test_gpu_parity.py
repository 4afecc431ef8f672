"""GPU parity: the HIP path (through the C-ABI) against the oracle.

Bar (BASELINE.json north_star): scores within 1e-5 — we hold them to BIT-EXACT —
and identical top-N track ids (tie-aware).
"""
import json

import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_canonical_order, assert_topn_matches

pytestmark = pytest.mark.gpu

SCORE_TOLERANCE = 1e-5  # north_star bound; the assertions below are stricter (bitwise)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    return torch


@pytest.fixture(scope="module")
def Engine(engine_lib, torch_cuda):
    from spotify_recommender_amd.engine import CosineEngine
    assert engine_lib.mi355rec_device_count() >= 1
    return CosineEngine


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


# ---- golden catalogue --------------------------------------------------------

def test_golden_scores_bit_exact(Engine, golden_dir):
    g = np.load(golden_dir / "catalogue4096.npz")
    with Engine(g["feats"]) as eng:
        for i, q in enumerate(g["queries"]):
            got = eng.scores_row(int(q))
            assert np.max(np.abs(got - g["scores"][i])) <= SCORE_TOLERANCE
            assert np.array_equal(bits(got), bits(g["scores"][i])), f"query row {q}"
            got2 = eng.scores(g["feats"][q])
            assert np.array_equal(bits(got2), bits(g["scores"][i]))


@pytest.mark.parametrize("topn", [1, 10, 100])
def test_golden_topn(Engine, golden_dir, topn):
    g = np.load(golden_dir / "catalogue4096.npz")
    with Engine(g["feats"]) as eng:
        for i, q in enumerate(g["queries"]):
            idx, sc = eng.query_row_topn(int(q), topn)
            assert_topn_matches(idx, sc, g["scores"][i], int(q), topn, ref_idx=g[f"heap_top{topn}"][i])
            assert idx.tolist() == g[f"canon_top{topn}"][i].tolist()
            assert_canonical_order(idx, g["scores"][i])


def test_survey_tie_fixture(Engine, golden_dir):
    """The reference's recorded tie case: same rows, order free inside ties."""
    pins = json.loads((golden_dir / "survey_pins.json").read_text())
    base = np.linspace(0.1, 1.0, 12, dtype=np.float32)
    f = np.tile(base, (9, 1))
    f[7] = base[::-1]              # clearly less similar
    f[8] = base + np.float32(0.3)  # between
    s = oracle.scores(f, f[0])
    assert s[8] > s[7] and np.all(s[1:7] == s[0])
    with Engine(f) as eng:
        for k in (1, 3, 6, 8):
            idx, sc = eng.query_row_topn(0, k)
            assert_topn_matches(idx, sc, s, 0, k, ref_idx=oracle.topn_heap(s, 0, k))
            assert sorted(idx.tolist()) == sorted(pins[f"tie_top{k}"]) or k < 6


# ---- seeded random catalogues, ragged sizes -----------------------------------

@pytest.mark.parametrize("rows", [1, 2, 3, 63, 64, 65, 511, 513, 2048, 4097, 33_333, 262_144, 1_000_003])
def test_random_sizes(Engine, rows):
    rng = np.random.default_rng(rows)
    f = rng.random((rows, 12), dtype=np.float32)
    with Engine(f) as eng:
        for q in sorted({0, rows // 2, rows - 1}):
            want = oracle.scores(f, f[q])
            got = eng.scores_row(q)
            assert np.array_equal(bits(got), bits(want))
            for topn in (1, 10, 100):
                idx, sc = eng.query_row_topn(q, topn)
                assert_topn_matches(idx, sc, want, q, topn, ref_idx=oracle.topn_heap(want, q, topn))


def test_signed_features_and_clamp(Engine):
    rng = np.random.default_rng(11)
    f = (rng.random((50_000, 12), dtype=np.float32) - np.float32(0.5)) * np.float32(4.0)
    f[100] = -f[0]            # cosine -1
    f[101] = f[0] * 3.0       # cosine +1 (may exceed 1 before the clamp)
    with Engine(f) as eng:
        want = oracle.scores(f, f[0])
        got = eng.scores_row(0)
        assert np.array_equal(bits(got), bits(want))
        assert want.min() >= -1.0 and want.max() <= 1.0
        idx, sc = eng.query_row_topn(0, 50)
        assert_topn_matches(idx, sc, want, 0, 50, ref_idx=oracle.topn_heap(want, 0, 50))


def test_special_values(Engine):
    rng = np.random.default_rng(5)
    f = rng.random((20_000, 12), dtype=np.float32)
    f[3] = 0.0
    f[4, 2] = np.nan
    f[5, 7] = np.inf
    f[6] = np.float32(1e-30)
    f[7] = np.float32(1e-42)   # denormal features
    f[8] = np.float32(3e18)    # norm^2 overflows -> inf
    f[9, 0] = -np.inf
    for q in (0, 3, 4, 5, 6, 7, 8, 9):
        with Engine(f) as eng:
            want = oracle.scores(f, f[q])
            got = eng.scores_row(q)
            assert np.array_equal(bits(got), bits(want)), f"query {q}"
            idx, sc = eng.query_row_topn(q, 20)
            assert_topn_matches(idx, sc, want, q, 20)


def test_sums_that_overflow_in_one_order_only(Engine):
    """fp32 sums that overflow in the reference's sequential order but not in a
    paired order (and vice versa): q = (1e19,1e19,1e19,0..) against
    (3e19,3e19,-3e19,0..) is inf/inf = NaN -> clamped to 1.0 in the reference.  The
    packed-FMA pre-filter must not be trusted there (kApproxMaxNorm2 /
    kApproxMaxQueryNorm); single-query and multi-query kernels, pre-filter active
    (200 k rows, thresholds > 0)."""
    rng = np.random.default_rng(77)
    f = rng.random((200_000, 12), dtype=np.float32)
    big = np.zeros(12, dtype=np.float32)
    big[:3] = (3e19, 3e19, -3e19)
    mid = np.zeros(12, dtype=np.float32)
    mid[:3] = (1e19, 1e19, -1e19)        # |row|^2 = 3e38: finite, but partial sums may overflow
    where = rng.choice(200_000, size=300, replace=False)
    f[where[:150]] = big
    f[where[150:]] = mid
    f[where[::7], 5] = 1e18
    queries = np.zeros((6, 12), dtype=np.float32)
    queries[0, :3] = 1e19
    queries[1, :3] = 2e19
    queries[2, :3] = 1e18                # |q| below the guard: pre-filter on, rows guarded one by one
    queries[3, :3] = (1e19, -1e19, 1e19)
    queries[4] = f[0]
    queries[5] = f[1] * np.float32(1e18)
    with Engine(f) as eng:
        wants = [oracle.scores(f, q) for q in queries]
        assert np.count_nonzero(wants[0] == 1.0) >= 150
        for q, want in zip(queries, wants):
            got = eng.scores(q)
            assert np.array_equal(bits(got), bits(want))
            for topn in (10, 100):
                idx, sc = eng.query_topn(q, -1, topn)
                assert_topn_matches(idx, sc, want, -1, topn, ref_idx=oracle.topn_heap(want, -1, topn))
        for path in (1, 0):   # the exact multi-query pass, then whatever AUTO picks
            eng.set_batch_path(path)
            idx, sc, counts = eng.query_batch_topn(queries, None, 100)
            for b, want in enumerate(wants):
                assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, -1, 100)


def test_all_rows_identical_exercises_merge_fallback(Engine):
    """Every score ties: each workgroup list is full of equal scores, the head
    threshold cannot prune, and the merge takes its exact radix-select path."""
    f = np.tile(np.linspace(0.05, 0.95, 12, dtype=np.float32), (300_000, 1))
    with Engine(f) as eng:
        want = oracle.scores(f, f[17])
        for topn in (1, 100, 1000):
            idx, sc = eng.query_row_topn(17, topn)
            expect = [i for i in range(topn + 1) if i != 17][:topn]
            assert idx.tolist() == expect           # canonical: lowest indices
            assert_topn_matches(idx, sc, want, 17, topn)


def test_ascending_scores_force_compaction(Engine):
    """Adversarial order: every later row beats the running threshold."""
    n = 200_000
    q = np.ones(12, dtype=np.float32)
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None]
    f = np.ones((n, 12), dtype=np.float32)
    f[:, :6] = 1.0 - 0.9 * (1.0 - t)      # rows approach the all-ones query
    want = oracle.scores(f, q)
    with Engine(f) as eng:
        for topn in (10, 100, 1024):
            idx, sc = eng.query_topn(q, -1, topn)
            assert_topn_matches(idx, sc, want, -1, topn, ref_idx=oracle.topn_heap(want, -1, topn))


def test_topn_larger_than_rows_and_bad_args(Engine):
    from spotify_recommender_amd import capi
    rng = np.random.default_rng(2)
    f = rng.random((4, 12), dtype=np.float32)
    with Engine(f) as eng:
        idx, sc = eng.query_row_topn(1, 10)          # reference prints "Top 3"
        assert len(idx) == 3 and 1 not in idx
        assert_topn_matches(idx, sc, oracle.scores(f, f[1]), 1, 10)
        for bad_topn in (0, -5):
            with pytest.raises(capi.Mi355Error) as e:
                eng.query_row_topn(0, bad_topn)
            assert e.value.code == capi.ERR_INVALID_ARG
        for bad_row in (-1, 4, 10**9):
            with pytest.raises(capi.Mi355Error) as e:
                eng.query_row_topn(bad_row, 3)
            assert e.value.code == capi.ERR_INVALID_ARG
            assert "Invalid song index" in str(e.value)   # Recommender.cu:282
    with Engine(f[:1]) as eng:                        # a single song has no neighbours
        idx, sc = eng.query_row_topn(0, 5)
        assert len(idx) == 0


def test_topn_above_the_fast_limit_runs_in_rounds(Engine):
    """topn > 1024: rounds of 1024 bounded by the previous round's last key."""
    rng = np.random.default_rng(77)
    f = rng.random((60_000, 12), dtype=np.float32)
    f[100:140] = f[7]                                  # exact ties straddling a round boundary
    with Engine(f) as eng:
        want = oracle.scores(f, f[7])
        for topn in (1025, 2048, 3000):
            idx, sc = eng.query_row_topn(7, topn)
            assert_topn_matches(idx, sc, want, 7, topn, ref_idx=oracle.topn_heap(want, 7, topn))
            assert_canonical_order(idx, want)
    small = rng.random((1500, 12), dtype=np.float32)
    with Engine(small) as eng:                         # catalogue exhausted inside round 2
        want = oracle.scores(small, small[0])
        idx, sc = eng.query_row_topn(0, 5000)
        assert len(idx) == 1499
        assert_topn_matches(idx, sc, want, 0, 5000)


def test_near_ties_inside_the_prefilter_margin(Engine):
    """Thousands of rows whose scores differ by a few ulps around the running
    threshold: the approximate pre-filter (margin 8e-6) must never drop a row
    the exact chain would keep."""
    rng = np.random.default_rng(31)
    n = 300_000
    f = rng.random((n, 12), dtype=np.float32)
    q = f[0].copy()
    cluster = rng.choice(np.arange(1, n), size=6000, replace=False)
    scale = rng.uniform(0.5, 2.0, size=(6000, 1)).astype(np.float32)
    noise = 1.0 + rng.uniform(-3e-7, 3e-7, size=(6000, 12))
    f[cluster] = (q[None, :] * scale * noise).astype(np.float32)   # cos within ~1e-7 of 1.0
    want = oracle.scores(f, q)
    assert np.unique(want[cluster]).size > 3            # several distinct top scores, many ties
    with Engine(f) as eng:
        got = eng.scores_row(0)
        assert np.array_equal(bits(got), bits(want))
        for topn in (10, 100, 1000):
            idx, sc = eng.query_row_topn(0, topn)
            assert_topn_matches(idx, sc, want, 0, topn, ref_idx=oracle.topn_heap(want, 0, topn))


def test_external_query_and_exclude(Engine):
    rng = np.random.default_rng(21)
    f = rng.random((100_000, 12), dtype=np.float32)
    q = rng.random(12, dtype=np.float32)
    want = oracle.scores(f, q)
    with Engine(f) as eng:
        idx, sc = eng.query_topn(q, -1, 100)
        assert_topn_matches(idx, sc, want, -1, 100, ref_idx=oracle.topn_heap(want, -1, 100))
        best = int(idx[0])
        idx2, sc2 = eng.query_topn(q, best, 100)
        assert best not in idx2
        assert_topn_matches(idx2, sc2, want, best, 100)


def test_batch_matches_single(Engine):
    rng = np.random.default_rng(8)
    f = rng.random((70_000, 12), dtype=np.float32)
    rows = [0, 999, 69_999, 12_345, 7]
    with Engine(f) as eng:
        for path in (1, 0):   # the exact multi-query pass, then whatever AUTO picks
            eng.set_batch_path(path)
            idx, sc, counts = eng.query_batch_topn(f[rows], rows, 25)
            assert counts.tolist() == [25] * len(rows)
            for b, r in enumerate(rows):
                want = oracle.scores(f, f[r])
                assert_topn_matches(idx[b], sc[b], want, r, 25, ref_idx=oracle.topn_heap(want, r, 25))
                i1, s1 = eng.query_row_topn(r, 25)
                assert i1.tolist() == idx[b].tolist()


@pytest.mark.parametrize("rows", [5, 700, 70_000, 1_000_003])
def test_multi_query_pass_matches_single_queries(Engine, torch_cuda, rows):
    """8 queries share one scan (scan_multi_kernel): same keys as 8 single scans,
    for full and ragged batches, row and external queries, ties and exclusions."""
    torch = torch_cuda
    rng = np.random.default_rng(rows)
    f = rng.random((rows, 12), dtype=np.float32)
    if rows > 100:
        f[50:60] = f[3]                                   # ties with a query
    from spotify_recommender_amd.engine import unpack_keys
    with Engine(f) as eng:
        # the exact multi-query passes (forced), then once more under AUTO (the batched
        # matrix-core path from 3 queries up on shards of >= 65536 rows)
        for path, (batch, topn) in [(p, bt) for p in (1, 0) for bt in ((1, 10), (3, 5), (8, 100), (19, 128), (9, 1), (40, 16), (5, 200))]:
            eng.set_batch_path(path)
            qrows = rng.integers(0, rows, size=batch)
            qrows[0] = min(3, rows - 1)
            queries = f[qrows].copy()
            excl = qrows.astype(np.int64)
            if batch > 2:
                queries[2] = rng.random(12, dtype=np.float32)   # an external query, nothing excluded
                excl[2] = -1
            idx, sc, counts = eng.query_batch_topn(queries, excl, topn)
            keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys(queries, excl, topn, keys)
            torch.cuda.synchronize()
            keys = keys.cpu().numpy().reshape(batch, topn)
            for b in range(batch):
                want = oracle.scores(f, queries[b])
                ex = int(excl[b])
                assert counts[b] == min(topn, rows - (1 if ex >= 0 else 0))
                assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, ex, topn,
                                    ref_idx=oracle.topn_heap(want, ex, topn))
                k_rows, _ = unpack_keys(keys[b])
                assert k_rows.tolist() == idx[b][:counts[b]].tolist()


def test_multi_query_pass_adversarial(Engine):
    """Ascending scores for every query of the batch (compaction at every tile,
    all 8 candidate buffers at once) and an all-identical catalogue."""
    n = 120_000
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None]
    f = np.ones((n, 12), dtype=np.float32)
    f[:, :6] = 1.0 - 0.9 * (1.0 - t)
    queries = np.ones((8, 12), dtype=np.float32)
    queries[:, 6:] += np.linspace(0, 0.01, 8, dtype=np.float32)[:, None]
    with Engine(f) as eng:
        for path in (1, 0):   # the exact multi-query pass, then whatever AUTO picks
            eng.set_batch_path(path)
            idx, sc, counts = eng.query_batch_topn(queries, None, 100)
            for b in range(8):
                want = oracle.scores(f, queries[b])
                assert_topn_matches(idx[b], sc[b], want, -1, 100, ref_idx=oracle.topn_heap(want, -1, 100))
    same = np.tile(np.linspace(0.05, 0.95, 12, dtype=np.float32), (90_000, 1))
    with Engine(same) as eng:
        for path in (1, 0):
            eng.set_batch_path(path)
            idx, sc, counts = eng.query_batch_topn(same[:8], np.arange(8), 64)
            for b in range(8):
                assert idx[b].tolist() == [i for i in range(65) if i != b][:64]


def test_borrowed_device_matrix_and_row_base(Engine, torch_cuda):
    """create_device over a torch tensor; row_base shifts ids (shard semantics)."""
    torch = torch_cuda
    rng = np.random.default_rng(13)
    f = rng.random((150_000, 12), dtype=np.float32)
    t = torch.from_numpy(f).cuda()
    lo, hi = 50_000, 125_000
    q = f[60_000]
    with Engine(t[lo:hi], row_base=lo) as eng:
        want = oracle.scores(f[lo:hi], q)
        idx, sc = eng.query_topn(q, 60_000, 100)
        assert idx.min() >= lo and idx.max() < hi and 60_000 not in idx
        assert_topn_matches(idx - lo, sc, want, 60_000 - lo, 100)


def test_two_shard_merge_equals_single(Engine, torch_cuda):
    """The multi-GPU data path on one device: two shards -> keys -> merge
    kernel == one engine over the whole catalogue (result independent of G)."""
    torch = torch_cuda
    rng = np.random.default_rng(99)
    f = rng.random((400_001, 12), dtype=np.float32)
    f[300_000] = f[5]                       # a cross-shard duplicate of the query
    t = torch.from_numpy(f).cuda()
    topn = 100
    from spotify_recommender_amd.engine import shard_bounds
    with Engine(t) as whole:
        for parts in (2, 3, 8):
            shards = [Engine(t[lo:hi], row_base=lo) for lo, hi in
                      (shard_bounds(f.shape[0], parts, r) for r in range(parts))]
            gathered = torch.zeros(parts * topn, dtype=torch.int64, device="cuda")
            out_keys = torch.zeros(topn, dtype=torch.int64, device="cuda")
            out_idx = torch.zeros(topn, dtype=torch.int64, device="cuda")
            out_score = torch.zeros(topn, dtype=torch.float32, device="cuda")
            for qrow in (5, 123_456, 400_000):
                for r, sh in enumerate(shards):
                    sh.enqueue_query_keys(f[qrow], qrow, topn, gathered[r * topn:(r + 1) * topn])
                shards[0].enqueue_merge_keys(gathered, parts, topn, topn, out_keys, out_idx, out_score)
                torch.cuda.synchronize()
                ref_idx, ref_sc = whole.query_row_topn(qrow, topn)
                assert out_idx.cpu().numpy().tolist() == ref_idx.tolist()
                assert np.array_equal(bits(out_score.cpu().numpy()), bits(ref_sc))
                # ... and both against the ORACLE, not only against each other
                want = oracle.scores(f, f[qrow])
                assert_topn_matches(out_idx.cpu().numpy(), out_score.cpu().numpy(), want, qrow, topn,
                                    ref_idx=oracle.topn_heap(want, qrow, topn))
            for sh in shards:
                sh.close()


def test_sharded_batch_merge_layout(Engine, torch_cuda):
    """Batch of queries over 3 shards: per-shard multi-query passes, the
    [rank][query][key] layout an all-gather produces, one batched merge launch ==
    the whole-catalogue engine, query by query."""
    torch = torch_cuda
    rng = np.random.default_rng(4242)
    f = rng.random((250_003, 12), dtype=np.float32)
    t = torch.from_numpy(f).cuda()
    from spotify_recommender_amd.engine import shard_bounds
    parts, batch, topn = 3, 11, 50
    qrows = rng.integers(0, f.shape[0], size=batch)
    with Engine(t) as whole:
        shards = [Engine(t[lo:hi], row_base=lo) for lo, hi in
                  (shard_bounds(f.shape[0], parts, r) for r in range(parts))]
        gathered = torch.zeros(parts * batch * topn, dtype=torch.int64, device="cuda")
        for r, sh in enumerate(shards):
            sh.enqueue_batch_keys(f[qrows], qrows, topn, gathered[r * batch * topn:(r + 1) * batch * topn])
        out_keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
        out_idx = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
        out_score = torch.zeros(batch * topn, dtype=torch.float32, device="cuda")
        shards[0].enqueue_merge_keys_batch(gathered, parts, topn, batch * topn, topn, batch, topn,
                                           out_keys, out_idx, out_score)
        torch.cuda.synchronize()
        got = out_idx.cpu().numpy().reshape(batch, topn)
        got_sc = out_score.cpu().numpy().reshape(batch, topn)
        for b, q in enumerate(qrows):
            ref_idx, ref_sc = whole.query_row_topn(int(q), topn)
            assert got[b].tolist() == ref_idx.tolist()
            want = oracle.scores(f, f[int(q)])
            assert_topn_matches(got[b], got_sc[b], want, int(q), topn, ref_idx=oracle.topn_heap(want, int(q), topn))
        for sh in shards:
            sh.close()


# ---- full BASELINE sizes ------------------------------------------------------

def test_reference_recorded_top3_at_1m_and_10m(Engine, golden_dir):
    """The HIP path reproduces what the REFERENCE returned for its own seeded
    1 M / 10 M catalogues — ids and 7-digit scores."""
    pins = json.loads((golden_dir / "survey_pins.json").read_text())
    for rows, key, topn in ((1_000_000, "1M_q0_top3", 10), (10_000_000, "10M_q0_top3", 100)):
        f = oracle.mt19937_uniform(12345, rows)
        with Engine(f) as eng:
            idx, sc = eng.query_row_topn(0, topn)
            for (want_i, want_s), got_i, got_s in zip(pins[key], idx[:3], sc[:3]):
                assert got_i == want_i
                assert f"{got_s:.7f}" == f"{want_s:.7f}"
            # and the whole list against the oracle on the same data
            want = oracle.scores(f, f[0], threads=0)
            assert_topn_matches(idx, sc, want, 0, topn, ref_idx=oracle.topn_heap(want, 0, topn))


def test_committed_full_size_fixture(Engine, golden_dir):
    """1 M / 10 M results against the COMMITTED fixture (tests/golden/
    seeded_top100.npz): ids, bit-exact scores and the CRC32 of the whole score
    vector — nothing here is recomputed by the oracle on the GPU box."""
    import zlib
    g = np.load(golden_dir / "seeded_top100.npz")
    for rows in (1_000_000, 10_000_000):
        f = oracle.mt19937_uniform(12345, rows)      # generator only (data, not results)
        with Engine(f) as eng:
            for q in g[f"queries_{rows}"]:
                idx, sc = eng.query_row_topn(int(q), 100)
                want_idx, want_sc = g[f"heap_idx_{rows}_{q}"], g[f"scores_{rows}_{q}"]
                assert np.array_equal(bits(sc), bits(want_sc + np.float32(0)))
                ties = np.concatenate([[False], want_sc[1:] == want_sc[:-1]]) | np.concatenate([want_sc[1:] == want_sc[:-1], [False]])
                assert np.array_equal(idx[~ties], want_idx[~ties])
                assert sorted(idx[ties].tolist()) == sorted(want_idx[ties].tolist()) or ties[-1]
                full = eng.scores_row(int(q))
                assert np.uint32(zlib.crc32(full.tobytes())) == g[f"crc32_{rows}_{q}"]


def test_full_size_10m_top100_properties(Engine, torch_cuda):
    """configs[2]: 10 M x 12, top-100, device-generated data.  Direct oracle
    comparison on a few queries plus size-independent properties."""
    torch = torch_cuda
    from spotify_recommender_amd.synth import query_rows, synthetic_catalogue
    n = 10_000_000
    t = synthetic_catalogue(n, seed=12345)
    f = t.cpu().numpy()
    with Engine(t) as eng:
        keys = torch.zeros(100, dtype=torch.int64, device="cuda")
        for q in query_rows(n, 4)[1:]:
            want = oracle.scores(f, f[q], threads=0)
            idx, sc = eng.query_row_topn(q, 100)
            assert_topn_matches(idx, sc, want, q, 100, ref_idx=oracle.topn_heap(want, q, 100))
            assert_canonical_order(idx, want)
            # idempotence + the async path returns the same keys
            idx2, _ = eng.query_row_topn(q, 100)
            assert idx2.tolist() == idx.tolist()
            eng.enqueue_row_keys(q, 100, keys)
            torch.cuda.synchronize()
            from spotify_recommender_amd.engine import unpack_keys
            k_rows, k_scores = unpack_keys(keys.cpu().numpy())
            assert k_rows.tolist() == idx.tolist()
            # top-10 is a prefix of top-100 (nested results)
            idx10, _ = eng.query_row_topn(q, 10)
            assert idx10.tolist() == idx[:10].tolist()
        # checksum of the full score vector against the oracle (bitwise)
        q = 7919
        got = eng.scores_row(q)
        want = oracle.scores(f, f[q], threads=0)
        assert np.array_equal(bits(got), bits(want))
        # the scans — fp32 rows, 8-bit replica (the default at this size) and, in experiment builds, the fp16 replica —
        # leave the same keys bit for bit, synchronously and as a stream (linearity of nothing, identity of everything)
        from spotify_recommender_amd import capi
        rows = query_rows(n, 12)
        per_mode = {}
        modes = (capi.REPLICA_OFF, capi.REPLICA_FP16, capi.REPLICA_ON) if capi.has_experiments() else (capi.REPLICA_OFF, capi.REPLICA_ON)
        for mode in modes:
            eng.set_replica(mode)
            out = torch.zeros((2 * len(rows), 100), dtype=torch.int64, device="cuda")
            for i, r in enumerate(rows):
                eng.enqueue_row_keys(int(r), 100, out[i])
            for i, r in enumerate(rows):
                eng.enqueue_row_keys_streamed(int(r), 100, out[len(rows) + i])
            eng.enqueue_flush()
            torch.cuda.synchronize()
            per_mode[mode] = out.cpu().numpy()
            assert np.array_equal(per_mode[mode][:len(rows)], per_mode[mode][len(rows):])
        for mode in modes[1:]:
            assert np.array_equal(per_mode[capi.REPLICA_OFF], per_mode[mode]), mode
        st = eng.stats()
        assert st.replica_single_row_bytes == 12 and st.replica_single_bytes_per_query == n // 4 * 48


def test_config5_shard_1024_query_batch(Engine, torch_cuda):
    """BASELINE configs[4] as seen by ONE of its 8 GPUs: a 12.5 M-row shard (row_base
    set as for rank 3), one batch of 1024 queries, top-100.  Served by the
    two-pass matrix-core path (csrc/batched.hip.h); sampled queries are checked against the oracle."""
    torch = torch_cuda
    from spotify_recommender_amd.engine import unpack_keys
    from spotify_recommender_amd.synth import synthetic_catalogue
    n_local, base, batch, topn = 12_500_000, 3 * 12_500_000, 1024, 100
    t = synthetic_catalogue(n_local, seed=5)
    f = t.cpu().numpy()
    rng = np.random.default_rng(55)
    local_rows = rng.integers(0, n_local, size=batch)
    queries = f[local_rows].copy()
    queries[1::2] = rng.random((batch // 2, 12), dtype=np.float32)      # half of them external
    excl = (local_rows + base).astype(np.int64)
    excl[1::2] = -1
    keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
    with Engine(t, row_base=base) as eng:
        eng.enqueue_batch_keys(queries, excl, topn, keys)
        torch.cuda.synchronize()
        got = keys.cpu().numpy().reshape(batch, topn)
        for b in (0, 1, 510, 1023):
            want = oracle.scores(f, queries[b], threads=0)
            rows, scores = unpack_keys(got[b])
            ex = int(excl[b]) - base if excl[b] >= 0 else -1
            assert_topn_matches(rows - base, scores, want, ex, topn, ref_idx=oracle.topn_heap(want, ex, topn))
        # every list is sorted in canonical order and carries global ids of this shard
        u = got.view(np.uint64)
        assert np.all(u[:, :-1] > u[:, 1:])
        all_rows, _ = unpack_keys(got.reshape(-1))
        assert all_rows.min() >= base and all_rows.max() < base + n_local


def test_100m_rows_on_one_gpu(Engine, torch_cuda):
    """The whole 100 M x 12 catalogue of BASELINE configs[4] (4.8 GB) resident on ONE
    MI355X: 64-bit row offsets, > 2^31 matrix elements, top-100 against the oracle."""
    torch = torch_cuda
    from spotify_recommender_amd.synth import synthetic_catalogue
    n = 100_000_000
    t = synthetic_catalogue(n, seed=100)
    f = t.cpu().numpy()
    with Engine(t) as eng:
        for q in (99_999_999, 53_687_091):        # beyond 2^31 / 48 and 2^32 / 48 bytes-offsets
            want = oracle.scores(f, f[q], threads=0)
            idx, sc = eng.query_row_topn(q, 100)
            assert_topn_matches(idx, sc, want, q, 100, ref_idx=oracle.topn_heap(want, q, 100))
        idx, sc, counts = eng.query_batch_topn(f[[7, 50_000_000]], [7, 50_000_000], 10)
        for b, q in enumerate((7, 50_000_000)):
            want = oracle.scores(f, f[q], threads=0)
            assert_topn_matches(idx[b], sc[b], want, q, 10)


def test_streamed_queries_defer_the_merge(Engine, torch_cuda):
    """mi355rec_enqueue_*_streamed: the merge of query k rides in the scan launch of query
    k + 1 (workgroup 0 of that launch), the last one is flushed.  Same keys as the plain
    path, with changing topn, interleaved plain calls and both query forms."""
    torch = torch_cuda
    from spotify_recommender_amd.engine import unpack_keys
    rng = np.random.default_rng(404)
    n = 700_001
    f = rng.random((n, 12), dtype=np.float32)
    f[123] = f[456]
    t = torch.from_numpy(f).cuda()
    rows = [int(r) for r in rng.integers(0, n, size=24)] + [123, 456]
    topns = [100, 100, 1, 10, 1000, 128, 100, 1024, 7] * 3
    with Engine(t) as eng:
        outs = [torch.zeros(1024, dtype=torch.int64, device="cuda") for _ in rows]
        plain = torch.zeros(1024, dtype=torch.int64, device="cuda")
        for i, r in enumerate(rows):
            k = topns[i]
            if i % 2 == 0:
                eng.enqueue_row_keys_streamed(r, k, outs[i])
            else:
                eng.enqueue_query_keys_streamed(f[r], r, k, outs[i])
            if i == 5:      # a plain query between two streamed ones must not disturb the pending lists
                eng.enqueue_row_keys(rows[0], 50, plain)
        eng.enqueue_flush()
        eng.enqueue_flush()   # nothing pending: no-op
        torch.cuda.synchronize()
        for i, r in enumerate(rows):
            k = topns[i]
            got_rows, got_sc = unpack_keys(outs[i][:k].cpu().numpy())
            want = oracle.scores(f, f[r])
            assert_topn_matches(got_rows, got_sc, want, r, k, ref_idx=oracle.topn_heap(want, r, k))
        p_rows, _ = unpack_keys(plain[:50].cpu().numpy())
        assert p_rows.tolist() == oracle.topn_canonical(oracle.scores(f, f[rows[0]]), rows[0], 50)[0].tolist()
    tiny = rng.random((3, 12), dtype=np.float32)      # fewer tiles than workgroups, topn > rows
    with Engine(tiny) as eng:
        o = [torch.zeros(8, dtype=torch.int64, device="cuda") for _ in range(3)]
        for i in range(3):
            eng.enqueue_row_keys_streamed(i, 8, o[i])
        eng.enqueue_flush()
        torch.cuda.synchronize()
        for i in range(3):
            got_rows, got_sc = unpack_keys(o[i].cpu().numpy())
            assert_topn_matches(got_rows, got_sc, oracle.scores(tiny, tiny[i]), i, 8)
