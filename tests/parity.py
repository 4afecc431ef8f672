"""Tie-aware comparison of a top-N result with the oracle.

Contract: the descending score list must be identical; ids must be identical
position by position except inside runs of exactly equal oracle scores, where
any permutation is accepted, and at a boundary tie (rank N score == rank N+1
score) where any of the tied rows is accepted.
"""
from __future__ import annotations

import numpy as np


def assert_topn_matches(got_idx, got_score, oracle_scores, exclude, topn, ref_idx=None):
    s = np.asarray(oracle_scores, dtype=np.float32)
    n = s.shape[0]
    got_idx = np.asarray(got_idx, dtype=np.int64)
    expect_count = min(topn, n - (1 if 0 <= exclude < n else 0))
    assert got_idx.shape[0] == expect_count, (got_idx.shape[0], expect_count)
    assert len(set(got_idx.tolist())) == expect_count, "duplicate rows in result"
    if 0 <= exclude < n:
        assert exclude not in set(got_idx.tolist()), "query row was not excluded"
    assert got_idx.min(initial=0) >= 0 and got_idx.max(initial=0) < n
    # scores of the returned rows, in order, must be the oracle's sorted top scores
    masked = s.astype(np.float64).copy()
    if 0 <= exclude < n:
        masked[exclude] = -np.inf
    want_sorted = np.sort(masked)[::-1][:expect_count].astype(np.float32)
    got_sorted = s[got_idx]
    assert np.array_equal(got_sorted + np.float32(0), want_sorted + np.float32(0)), "score list differs from the oracle"
    if got_score is not None:
        gs = np.asarray(got_score, dtype=np.float32)[:expect_count]
        # bit-exact (the engine reports -0.0 as +0.0)
        assert np.array_equal((gs + np.float32(0)).view(np.uint32), (got_sorted + np.float32(0)).view(np.uint32)), "reported scores differ"
    if ref_idx is not None:
        # positions outside exact-tie runs must agree with the reference order
        ref_idx = np.asarray(ref_idx, dtype=np.int64)
        assert ref_idx.shape[0] == expect_count
        rs = s[ref_idx]
        for p in range(expect_count):
            tied_prev = p > 0 and rs[p - 1] == rs[p]
            tied_next = p + 1 < expect_count and rs[p + 1] == rs[p]
            boundary = p == expect_count - 1 and np.count_nonzero(masked == rs[p]) > 1
            if not (tied_prev or tied_next or boundary):
                assert got_idx[p] == ref_idx[p], (p, got_idx[p], ref_idx[p])


def assert_canonical_order(got_idx, oracle_scores):
    """(score desc, index asc) — the engine's documented order."""
    s = np.asarray(oracle_scores, dtype=np.float32)[np.asarray(got_idx, dtype=np.int64)]
    idx = np.asarray(got_idx, dtype=np.int64)
    for p in range(1, len(idx)):
        assert s[p - 1] > s[p] or (s[p - 1] == s[p] and idx[p - 1] < idx[p]), p
