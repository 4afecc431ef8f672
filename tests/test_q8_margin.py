"""The error bound of the 8-bit replica's pre-filter (csrc/replica_q8.hip.h), checked on the CPU with a
numpy model of exactly that arithmetic against the oracle's exact scores:

    r^ = r * rsqrt(|r|^2) (fp32),  u_j = rint(127 r^_j) + 128 in [1, 255]
    q^ = q / |q| (fp32, NOT quantised),  s_j = q^_j / 127,  c = 128 sum_j s_j
    approx = fma-chain(sum_j s_j u_j) - c                        (fp32)
    margin(q) = l1(q^) / 254 * (1 + 1e-5) + 3e-5

for every (row, query) pair the kernel claims the bound for: |r|^2 in [1.01e-8, 1e36], |q| in
[1.005e-4, 1e18].  The bound is per query; the test also checks that it is not vacuous."""
import numpy as np

from oracle import oracle
from tests.test_batched_margin import catalogues

MIN_NORM2, MAX_NORM2 = 1.01e-8, 1e36


def q8_model(rows, q):
    rows = rows.astype(np.float32)
    n2 = np.zeros(len(rows), np.float32)
    for j in range(12):
        n2 = n2 + rows[:, j] * rows[:, j]
    valid = (n2 >= np.float32(MIN_NORM2)) & (n2 <= np.float32(MAX_NORM2))
    inv = np.zeros_like(n2)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv[valid] = (np.float32(1) / np.sqrt(n2[valid])).astype(np.float32) * np.float32(127)
        k = np.clip(np.rint(rows * inv[:, None]), -127, 127)
    u = (k + 128).astype(np.float32)
    q = q.astype(np.float32)
    qn = np.float32(np.sqrt(np.sum(q * q, dtype=np.float32)))
    q_ok = bool(np.float32(1.005e-4) <= qn <= np.float32(1e18))
    if not q_ok:
        return None, valid, False, 0.0
    qhat = (q / qn).astype(np.float32)
    s = (qhat * np.float32(1 / 127)).astype(np.float32)
    c = np.float32(128) * np.sum(s, dtype=np.float32)
    acc = np.full(len(rows), -c, np.float32)
    for j in range(12):   # fp32, one rounding per step (an fma rounds once; this model rounds the product too: not tighter)
        acc = (acc + u[:, j] * s[j]).astype(np.float32)
    margin = float(np.sum(np.abs(qhat), dtype=np.float32)) / 254.0 * (1 + 1e-5) + 3e-5
    return acc.astype(np.float64), valid, True, margin


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


def test_q8_codes_and_special_rows():
    """u stays in [1, 255] for valid rows (0 is the special marker), an exactly-zero row is all 128."""
    rng = np.random.default_rng(3)
    f = (rng.normal(0, 1, (20_000, 12)) * 10.0 ** rng.integers(-3, 4, (20_000, 1))).astype(np.float32)
    f[5] = 0.0
    n2 = np.sum(f.astype(np.float32) ** 2, axis=1, dtype=np.float32)
    inv = (np.float32(1) / np.sqrt(np.where(n2 > 0, n2, 1))).astype(np.float32) * np.float32(127)
    inv[n2 == 0] = 0
    u = np.clip(np.rint(f * inv[:, None]), -127, 127) + 128
    assert u.min() >= 1 and u.max() <= 255
    assert (u[5] == 128).all()
