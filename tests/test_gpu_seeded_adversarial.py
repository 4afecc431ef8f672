"""Hostile catalogues at a size where the multi-query pass takes its SEEDED
path (mi355::seed_multi_kernel + seed_select_kernel: the one place where an
approximate value sets a chip-wide threshold).  The seed is used from 3 tiles
per workgroup up, i.e. above ~525 k rows with the 512-workgroup grid; every
case here has 700 003 rows, batches of 13-36 queries (so chains cross the
12-query pass boundary) and topn in {1, 100, 128}.  Everything is compared
with the oracle, query by query.

The sample is the first 512 rows of 512 regions spaced floor(n/512/64)*64 rows
apart; "in the sample" below means rows placed at those offsets.
"""
import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches

pytestmark = pytest.mark.gpu

N = 700_003
REGION = N // 512 // 64 * 64          # 1344: distance between sampled regions
SAMPLED = np.array([b * REGION + o for b in range(0, 512, 7) for o in (0, 5, 63, 64, 300, 511)])
UNSAMPLED = np.array([b * REGION + o for b in range(3, 512, 11) for o in (512, 700, REGION - 1)])


@pytest.fixture(scope="module")
def Engine():
    import torch
    assert torch.cuda.is_available()
    from spotify_recommender_amd.engine import CosineEngine
    return CosineEngine


def run_case(Engine, f, queries, excl, label, topns=(1, 100, 128), batches=(13, 24, 36), single=2):
    f = np.ascontiguousarray(f, dtype=np.float32)
    queries = np.ascontiguousarray(queries, dtype=np.float32)
    excl = np.asarray(excl, dtype=np.int64)
    want = [oracle.scores(f, q, threads=0) for q in queries]
    with Engine(f) as eng:
        # 1 = the exact multi-query passes with their sampled seed (what this file is about),
        # 2 = the batched matrix-core path (what AUTO picks for these batch sizes)
        for path in (1, 2):
            eng.set_batch_path(path)
            for topn, batch in zip(topns, batches):
                idx, sc, counts = eng.query_batch_topn(queries[:batch], excl[:batch], topn)
                for b in range(batch):
                    try:
                        assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want[b], int(excl[b]), topn,
                                            ref_idx=oracle.topn_heap(want[b], int(excl[b]), topn))
                    except AssertionError as e:
                        raise AssertionError(f"{label}: path {path} batch {batch} topn {topn} query {b}: {e}") from e
        eng.set_batch_path(0)
        for b in range(single):   # the single-query kernel on the same data
            idx, sc = eng.query_topn(queries[b], int(excl[b]), 100)
            assert_topn_matches(idx, sc, want[b], int(excl[b]), 100, ref_idx=oracle.topn_heap(want[b], int(excl[b]), 100))
        # ... and the scan over the fp16 replica (csrc/replica.hip.h; forced on: AUTO starts at 1 M rows):
        # its launch-wide cutoff also comes from a sample, of an approximation with a much wider margin
        eng.set_replica(2)
        for b in range(min(len(queries), 8)):
            for topn in (1, 100, 1000):
                idx, sc = eng.query_topn(queries[b], int(excl[b]), topn)
                try:
                    assert_topn_matches(idx, sc, want[b], int(excl[b]), topn, ref_idx=oracle.topn_heap(want[b], int(excl[b]), topn))
                except AssertionError as e:
                    raise AssertionError(f"{label}: replica scan, topn {topn} query {b}: {e}") from e


def perturbed_ones(rng, count):
    q = np.ones((count, 12), dtype=np.float32)
    q[:, 6:] += rng.random((count, 6), dtype=np.float32) * np.float32(0.02)
    return q


@pytest.mark.parametrize("descending", [False, True])
def test_catalogue_ordered_by_similarity(Engine, descending):
    """Similarity to every query grows (or falls) monotonically with the row index:
    an early sample says nothing about the far end."""
    rng = np.random.default_rng(1)
    t = np.linspace(0.0, 1.0, N, dtype=np.float32)
    if descending:
        t = t[::-1]
    f = np.ones((N, 12), dtype=np.float32)
    f[:, :6] = (0.1 + 0.9 * t)[:, None]
    f[:, 6:] += rng.random((N, 6), dtype=np.float32) * np.float32(1e-3)
    run_case(Engine, f, perturbed_ones(rng, 36), np.full(36, -1), f"ordered desc={descending}")


def test_duplicates_of_the_query_and_mass_ties_at_the_nth_score(Engine):
    rng = np.random.default_rng(2)
    f = rng.random((N, 12), dtype=np.float32)
    qrows = rng.choice(N, size=36, replace=False)
    # > topn exact duplicates of queries 0..3, inside and outside the sample
    for k in range(4):
        where = np.concatenate([rng.choice(SAMPLED, 80, replace=False) + k * 7, rng.choice(UNSAMPLED, 80, replace=False) + k])
        f[where % N] = f[qrows[k]]
    # a run of 5000 rows that tie EXACTLY (power-of-two multiples of one vector) just
    # below 60 better rows: the 100th / 128th best score sits inside the run
    v = rng.random(12, dtype=np.float32) + np.float32(0.5)
    tie_rows = rng.choice(N, size=5000, replace=False)
    f[tie_rows] = v[None, :] * (np.float32(2.0) ** rng.integers(-3, 4, size=5000)).astype(np.float32)[:, None]
    better = rng.choice(np.setdiff1d(np.arange(N), tie_rows), size=60, replace=False)
    tie_query = v * np.float32(1.0) + rng.random(12, dtype=np.float32) * np.float32(0.05)
    f[better] = tie_query[None, :] * (1 + rng.random((60, 12), dtype=np.float32) * np.float32(1e-3))
    queries = f[qrows].copy()
    excl = qrows.astype(np.int64)
    queries[4:8] = tie_query * np.array([1, 2, 0.5, 3], dtype=np.float32)[:, None]
    excl[4:8] = -1
    s = oracle.scores(f, queries[4], threads=0)
    top = np.sort(s)[::-1]
    assert top[99] == top[100] == top[127] == top[128], "construction: the N-th score must sit inside the tie run"
    run_case(Engine, f, queries, excl, "duplicates + ties")


def test_signed_wide_dynamic_range(Engine):
    rng = np.random.default_rng(3)
    f = (rng.normal(0, 1, size=(N, 12)) * 10.0 ** rng.integers(-3, 4, size=(N, 1))).astype(np.float32)
    qrows = rng.choice(N, size=36, replace=False)
    queries = f[qrows].copy()
    queries[::3] = (rng.normal(0, 1, size=(12, 12)) * 10.0 ** rng.integers(-3, 4, size=(12, 1))).astype(np.float32)
    excl = qrows.astype(np.int64)
    excl[::3] = -1
    run_case(Engine, f, queries, excl, "signed 1e-3..1e3")


def test_norm_products_straddling_the_1e8_threshold(Engine):
    """|row|*|q| on both sides of the reference's `> 1e-8f` test: rows below it score
    exactly 0 whatever their direction (Recommender.cu:271)."""
    rng = np.random.default_rng(4)
    f = rng.random((N, 12), dtype=np.float32)
    scale = (10.0 ** rng.uniform(-4.6, -3.4, size=N)).astype(np.float32)   # |row| ~ 5e-5 .. 8e-4
    f *= scale[:, None]
    normal = rng.choice(N, size=N // 20, replace=False)                    # 5 % ordinary rows
    f[normal] = rng.random((len(normal), 12), dtype=np.float32)
    f[SAMPLED[:40]] = rng.random((40, 12), dtype=np.float32) * np.float32(3e-5)
    queries = rng.random((36, 12), dtype=np.float32)
    queries[:18] *= (10.0 ** rng.uniform(-5.0, -4.2, size=18)).astype(np.float32)[:, None]   # ~half the rows land below 1e-8
    queries[18:24] *= np.float32(1e-6)                                     # everything but huge rows -> 0
    run_case(Engine, f, queries, np.full(36, -1), "norm product ~ 1e-8")


def test_special_rows_inside_the_sample(Engine):
    """NaN / inf / denormal / overflowing rows exactly where the seed looks, plus the
    mixed-sign 1e19 rows whose sequential sums overflow to inf/inf = NaN -> 1.0 in the
    reference while a reordered sum stays finite (the approximate chains must not be
    trusted there)."""
    rng = np.random.default_rng(5)
    f = rng.random((N, 12), dtype=np.float32)
    vals = [np.nan, np.inf, -np.inf, 1e-42, 3e19, -3e19, 0.0]
    for i, r in enumerate(SAMPLED):
        f[r, rng.integers(0, 12)] = vals[i % len(vals)]
    f[SAMPLED[::5] + 1] = 0.0
    f[SAMPLED[::9] + 2] = np.float32(1e-42)
    big = np.zeros(12, dtype=np.float32)
    big[:3] = (3e19, 3e19, -3e19)
    f[SAMPLED[::4] + 3] = big
    f[UNSAMPLED[::4]] = big * np.array([1, 1, 1] + [0] * 9, dtype=np.float32) + np.float32(0)
    qrows = rng.choice(N, size=36, replace=False)
    queries = f[qrows].copy()
    excl = qrows.astype(np.int64)
    qbig = np.zeros(12, dtype=np.float32)
    qbig[:3] = 1e19
    queries[0] = qbig                      # the ADVICE case: exact = NaN -> clamped to 1.0
    queries[1] = qbig * np.float32(0.1)
    queries[2] = f[SAMPLED[1]]             # a query with an inf component
    queries[3] = f[SAMPLED[0]]             # a query with a NaN component
    queries[4] = 0.0
    excl[:5] = -1
    want = oracle.scores(f, queries[0], threads=0)
    assert np.count_nonzero(want == 1.0) >= len(SAMPLED[::4]), "construction: the overflow rows must score 1.0"
    run_case(Engine, f, queries, excl, "special values in the sample")
