"""The error bound of the 8-bit replica's pre-filter (csrc/replica_q8.hip.h), checked on the CPU with a
numpy model of exactly that arithmetic against the oracle's exact scores:

    r^ = r * rsqrt(|r|^2) (fp32),  k_j = rint(127 r^_j) in [-127, 127]            (signed bytes; 0x80 = "score exactly")
    q^ = q / |q| (fp32),  Q_j = rint(S q^_j) = 256 h_j + l_j  (S = 32000; balanced int8 digits h, l)
    D = 256 sum_j k_j h_j + sum_j k_j l_j                                         (int32: six v_dot4_i32_i8 and a shift-add)
    approx = D / (127 S)
    margin(q) = l1(Q) / (254 S) * (1 + 1e-5) + sqrt(12) / (2 S) + 3e-5

for every (row, query) pair the kernel claims the bound for: |r|^2 in [1.01e-8, 1e36], |q| in
[1.005e-4, 1e18].  The bound is per query; the test also checks that it is not vacuous, that the digits are what
v_dot4_i32_i8 needs (int8) and that the integer image of a cutoff (q8_threshold) never rules out a row the float
comparison would keep."""
import numpy as np

from oracle import oracle
from tests.test_batched_margin import catalogues

MIN_NORM2, MAX_NORM2 = 1.01e-8, 1e36
S = 32000
DOT_SCALE = np.float32(127.0 * S)


def q8_codes(rows):
    rows = rows.astype(np.float32)
    n2 = np.zeros(len(rows), np.float32)
    for j in range(12):
        n2 = n2 + rows[:, j] * rows[:, j]
    valid = (n2 >= np.float32(MIN_NORM2)) & (n2 <= np.float32(MAX_NORM2))
    inv = np.zeros_like(n2)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv[valid] = (np.float32(1) / np.sqrt(n2[valid])).astype(np.float32) * np.float32(127)
        k = np.clip(np.rint(rows * inv[:, None]), -127, 127)
    return np.nan_to_num(k).astype(np.int64), valid


def q8_digits(q):
    """(Q, h, l, ok) of a query: the 16-bit image and its balanced digits."""
    q = q.astype(np.float32)
    qn = np.float32(np.sqrt(np.sum(q * q, dtype=np.float32)))
    ok = bool(np.float32(1.005e-4) <= qn <= np.float32(1e18))
    if not ok:
        return None, None, None, False
    qhat = (q * (np.float32(1) / qn)).astype(np.float32)
    Q = np.clip(np.rint((qhat * np.float32(S)).astype(np.float32)), -S, S).astype(np.int64)
    h = (Q + 128) >> 8
    l = Q - 256 * h
    return Q, h, l, True


def q8_model(rows, q):
    k, valid = q8_codes(rows)
    Q, h, l, ok = q8_digits(q)
    if not ok:
        return None, valid, False, 0.0
    D = 256 * (k @ h) + (k @ l)
    assert np.array_equal(D, k @ Q)
    assert np.abs(D).max() < 2 ** 24            # converts to fp32 exactly
    approx = (D.astype(np.float32) * (np.float32(1) / DOT_SCALE)).astype(np.float32)
    margin = float(np.float32(np.abs(Q).sum()) * np.float32(1 / 254.0 / S) * np.float32(1 + 1e-5)
                   + np.float32(3.4642 * 0.5 / S) + np.float32(3e-5))
    return approx.astype(np.float64), valid, True, margin


def q8_threshold(cutoff):
    """csrc/replica_q8.hip.h q8_threshold: a row is ruled out iff D < this."""
    cutoff = np.float32(cutoff)
    if not (cutoff > np.float32(-2)):
        return -2 ** 31
    c = min(cutoff, np.float32(2))
    return int(np.floor(np.float32(c * DOT_SCALE))) - 1


def test_q8_prefilter_error_stays_inside_the_per_query_margin():
    rng = np.random.default_rng(2026)
    n = 60_000
    worst_ratio = 0.0
    for name, f in catalogues(rng, n):
        f = np.ascontiguousarray(f, dtype=np.float32)
        queries = [f[rng.integers(0, n)], f[rng.integers(0, n)] * np.float32(3), rng.random(12, dtype=np.float32),
                   rng.normal(0, 1, 12).astype(np.float32), np.eye(12, dtype=np.float32)[3],
                   -f[rng.integers(0, n)], np.full(12, 0.3, np.float32)]
        for q in queries:
            approx, valid, q_ok, margin = q8_model(f, q)
            if not q_ok:
                continue
            exact = oracle.scores(f, np.ascontiguousarray(q, dtype=np.float32)).astype(np.float64)
            err = np.abs(approx - exact)[valid]
            if err.size:
                worst_ratio = max(worst_ratio, float(err.max()) / margin)
                assert err.max() <= margin, (name, float(err.max()), margin)
    # the bound is not vacuous (a uniform quantisation error realises a good part of l1 / 254) and not exceeded
    assert 0.3 < worst_ratio <= 1.0, worst_ratio


def test_q8_codes_digits_and_special_rows():
    """k stays in [-127, 127] for valid rows (-128 is the special marker), an exactly-zero row is all 0; the query's digits are
    int8 with Q = 256 h + l."""
    rng = np.random.default_rng(3)
    f = (rng.normal(0, 1, (20_000, 12)) * 10.0 ** rng.integers(-3, 4, (20_000, 1))).astype(np.float32)
    f[5] = 0.0
    k, valid = q8_codes(f)
    assert k.min() >= -127 and k.max() <= 127
    assert (k[5] == 0).all()
    for q in (f[7], -f[9], np.eye(12, dtype=np.float32)[0], -np.eye(12, dtype=np.float32)[11], np.full(12, 1e-3, np.float32)):
        Q, h, l, ok = q8_digits(q)
        assert ok
        assert np.array_equal(Q, 256 * h + l)
        assert h.min() >= -125 and h.max() <= 125 and l.min() >= -128 and l.max() <= 127
        assert np.abs(Q).max() <= S


def test_q8_integer_threshold_is_conservative():
    """D < q8_threshold(cutoff) implies D / (127 S) < cutoff in real arithmetic, and the integer test gives away at most two
    units of D (5e-7 of a score)."""
    rng = np.random.default_rng(11)
    for cutoff in np.concatenate([rng.uniform(-1.1, 1.1, 20_000), [0.0, 1.0, -1.0, 0.9999999, 1e-9, -1e-9]]).astype(np.float32):
        t = q8_threshold(cutoff)
        real = float(cutoff) * 127.0 * S            # exact in float64
        assert t - 1 < real                         # the largest D that is ruled out (t - 1) lies below the cutoff
        assert real - t <= 2.3                      # ... and not by more than the guard
    assert q8_threshold(-np.inf) == -2 ** 31 and q8_threshold(np.nan) == -2 ** 31
    assert q8_threshold(np.float32(5.0)) == q8_threshold(np.float32(2.0))
