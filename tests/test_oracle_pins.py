"""The oracle against everything the reference left us to pin it with.

No reference tests exist; the pins are reference outputs the
survey recorded (tests/golden/survey_pins.json), the real libstdc++
priority_queue, and the committed golden catalogue.
"""
import json
import subprocess

import numpy as np
import pytest

from oracle import oracle


@pytest.fixture(scope="module")
def pins(golden_dir):
    return json.loads((golden_dir / "survey_pins.json").read_text())


def test_tie_orders_match_reference(pins):
    s = np.array(pins["tie_scores"], dtype=np.float32)
    q = pins["tie_query"]
    for k in (1, 3, 6, 8):
        assert oracle.topn_heap(s, q, k).tolist() == pins[f"tie_top{k}"]


def test_zero_query_order(pins):
    s = np.zeros(9, dtype=np.float32)
    assert oracle.topn_heap(s, 5, 3).tolist() == pins["zero_query_top3"]


@pytest.mark.parametrize("rows,key", [(1_000_000, "1M_q0_top3"), (10_000_000, "10M_q0_top3")])
def test_seeded_catalogue_top3_match_reference(pins, rows, key):
    f = oracle.mt19937_uniform(12345, rows)
    idx = oracle.recommend_by_index(f, 0, 100)
    sc = oracle.scores(f[idx[:3]], f[0])
    for (want_i, want_s), got_i, got_s in zip(pins[key], idx[:3], sc):
        assert got_i == want_i
        assert f"{got_s:.7f}" == f"{want_s:.7f}"
    # the OpenMP baseline variant returns the same rows
    idx2, sc2 = oracle.recommend_omp(f, 0, 100, threads=4)
    s_all = oracle.scores(f, f[0], threads=4)
    assert np.array_equal(np.sort(s_all[idx2])[::-1], np.sort(s_all[idx])[::-1])


def test_heap_replay_equals_libstdcxx(tmp_path):
    exe = tmp_path / "heap_check"
    src = __file__.rsplit("/", 1)[0] + "/heap_check.cpp"
    cmd = ["g++", "-std=c++11", "-O2", "-o", str(exe), src]
    from pathlib import Path
    if Path("/root/reference/Recommender.h").exists():
        # the reference's own Recommendation struct + comparator (Recommender.h:12-22)
        cmd += ["-DHEAP_CHECK_REFERENCE_HEADER", "-I", "/root/reference"]
    subprocess.run(cmd, check=True)
    rng = np.random.default_rng(7)
    for trial in range(40):
        n = int(rng.integers(1, 400))
        levels = int(rng.integers(1, 8))           # few distinct values -> many ties
        s = (rng.integers(0, levels, size=n) / np.float32(levels)).astype(np.float32)
        exclude = int(rng.integers(-1, n))
        topn = int(rng.integers(1, n + 3))
        out = subprocess.run([str(exe), str(n), str(exclude), str(topn)], input=s.tobytes(),
                             capture_output=True, check=True).stdout.decode().split()
        assert oracle.topn_heap(s, exclude, topn).tolist() == [int(v) for v in out]


def test_golden_catalogue_regenerates(golden_dir):
    g = np.load(golden_dir / "catalogue4096.npz")
    f = g["feats"]
    for i, q in enumerate(g["queries"]):
        s = oracle.scores(f, f[q])
        assert np.array_equal(s.view(np.uint32), g["scores"][i].view(np.uint32))
        for k in (1, 10, 100):
            assert oracle.topn_heap(s, int(q), k).tolist() == g[f"heap_top{k}"][i].tolist()
            assert oracle.topn_canonical(s, int(q), k)[0].tolist() == g[f"canon_top{k}"][i].tolist()


def test_observable_semantics(golden_dir):
    """the behaviours the drop-in must preserve."""
    g = np.load(golden_dir / "catalogue4096.npz")
    f, sc, qs = g["feats"], g["scores"], g["queries"].tolist()
    s5 = sc[qs.index(5)]
    assert s5[100] == 0.0                      # zero-norm row -> exactly 0
    assert s5[300] == 0.0                      # NaN feature -> 0
    assert np.all(sc[qs.index(100)] == 0.0)    # zero-norm query -> all 0
    assert np.all(sc[qs.index(300)] == 0.0)    # NaN query -> all 0
    assert sc[qs.index(7)][200] == 0.0         # orthogonal rows
    assert s5[10] == s5[11] == s5[12] == s5[5]  # duplicates of the query tie with it
    top = oracle.topn_heap(s5, 5, 10).tolist()
    assert 5 not in top and {10, 11, 12} <= set(top)  # excluded by index, duplicates kept
    assert np.all(np.abs(sc) <= 1.0)
    # AoS stride (sizeof(Song)==152 B == 38 floats) gives the same scores
    aos = np.zeros((64, 38), dtype=np.float32)
    aos[:, 26:38] = f[:64]
    assert np.array_equal(oracle.scores(aos[:, 26:38], f[5]), s5[:64])
    # topn > n-1 -> n-1 results
    assert len(oracle.topn_heap(s5[:4], 1, 10)) == 3


def test_seeded_top100_fixture_matches_survey_pins(golden_dir, pins):
    g = np.load(golden_dir / "seeded_top100.npz")
    for rows, key in ((1_000_000, "1M_q0_top3"), (10_000_000, "10M_q0_top3")):
        for (want_i, want_s), got_i, got_s in zip(pins[key], g[f"heap_idx_{rows}_0"][:3], g[f"scores_{rows}_0"][:3]):
            assert got_i == want_i and f"{got_s:.7f}" == f"{want_s:.7f}"
