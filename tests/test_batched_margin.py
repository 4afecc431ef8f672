"""The error bound of the batched path's fp16 pre-filter (csrc/batched.hip.h,
kBqMargin = 1.0e-3 with fp16 subnormals kept, kBqMarginFlush = 1.5e-3), checked on the CPU with a numpy model of exactly that
arithmetic against the oracle's exact scores:

    r^ = r * rsqrt(|r|^2), q^ = q / |q|         (fp32)
    approx = sum_j fp16(r^_j) * fp16(q^_j)      (products exact, fp32 accumulate)

for every (row, query) pair the kernel claims the bound for: |r|^2 and |q|^2 in
[1.01e-8, 1e36].  Two models of the fp16 conversion are checked: IEEE
round-to-nearest (what v_cvt_pk_f16_f32 does) and the same with subnormal
results flushed to zero (the bound is derived for that case too, so it does not
depend on how the matrix core treats fp16 subnormals).
"""
import numpy as np
import pytest

from oracle import oracle

MARGIN = {False: 1.0e-3, True: 1.5e-3}   # fp16 subnormals kept (gfx950, checked on the device) / flushed
MIN_NORM2, MAX_NORM2 = 1.01e-8, 1e36


def to_f16(x, flush):
    h = x.astype(np.float16)
    if flush:
        h = np.where(np.abs(h) < np.float16(6.104e-05), np.float16(0), h)
    return h.astype(np.float64)


def model(rows, q, flush):
    rows = rows.astype(np.float32)
    n2 = np.zeros(len(rows), np.float32)
    for j in range(12):
        n2 = n2 + rows[:, j] * rows[:, j]
    valid = (n2 >= np.float32(MIN_NORM2)) & (n2 <= np.float32(MAX_NORM2))
    inv = np.zeros_like(n2)
    inv[valid] = (np.float32(1) / np.sqrt(n2[valid])).astype(np.float32)
    rhat = (rows * inv[:, None]).astype(np.float32)
    rhat[~valid] = 0
    qn = np.float32(np.sqrt(np.sum(q.astype(np.float32) ** 2, dtype=np.float32)))
    qhat = (q.astype(np.float32) / qn).astype(np.float32)
    approx = to_f16(rhat, flush) @ to_f16(qhat, flush)
    return approx, valid, (1.005e-4 <= qn <= 1e18)


def catalogues(rng, n):
    yield "uniform", rng.random((n, 12), dtype=np.float32)
    yield "signed wide range", (rng.normal(0, 1, (n, 12)) * 10.0 ** rng.integers(-3, 4, (n, 1))).astype(np.float32)
    f = rng.random((n, 12), dtype=np.float32)
    f[rng.random((n, 12)) < 0.7] = 0
    yield "sparse", f
    # one dominant component, the others down in fp16's subnormal range after normalisation
    f = (rng.random((n, 12), dtype=np.float32) * 10.0 ** rng.uniform(-7.5, -4, (n, 12))).astype(np.float32)
    f[np.arange(n), rng.integers(0, 12, n)] = 1.0
    yield "subnormal tails", f
    # components that sit on fp16 rounding midpoints: (k + 0.5) ulp at every binade
    k = rng.integers(1024, 2048, (n, 12)).astype(np.float64) + 0.5
    f = (k * 2.0 ** rng.integers(-14, -10, (n, 12)) * rng.choice([-1, 1], (n, 12))).astype(np.float32)
    yield "rounding midpoints", f
    yield "tiny rows near the validity edge", (rng.random((n, 12), dtype=np.float32) * np.float32(6e-5))
    yield "huge rows", (rng.normal(0, 1, (n, 12)) * 1e17).astype(np.float32)


@pytest.mark.parametrize("flush", [False, True])
def test_prefilter_error_stays_inside_the_margin(flush):
    rng = np.random.default_rng(99)
    n = 60_000
    worst = 0.0
    for name, f in catalogues(rng, n):
        queries = [f[rng.integers(0, n)], f[rng.integers(0, n)] * np.float32(3), rng.random(12, dtype=np.float32),
                   rng.normal(0, 1, 12).astype(np.float32), np.eye(12, dtype=np.float32)[3]]
        if name.startswith("rounding"):
            queries += [f[rng.integers(0, n)] for _ in range(4)]
        for q in queries:
            approx, valid, q_ok = model(f, q, flush)
            if not q_ok:
                continue
            exact = oracle.scores(f, q.astype(np.float32)).astype(np.float64)
            err = np.abs(approx - exact)[valid]
            if err.size:
                worst = max(worst, float(err.max()))
                assert err.max() < MARGIN[flush], (name, flush, float(err.max()))
    # the bound is not vacuous: fp16 rounding of both operands really costs ~2^-11..2^-10
    assert 1e-4 < worst < MARGIN[flush], worst


def test_sequential_fp32_accumulation_stays_inside_the_slack():
    """The scan over the fp16 replica (csrc/replica.hip.h) accumulates the twelve exact fp16 x fp16
    products with fp32 FMAs in index order instead of inside the matrix core.  Against the exactly
    summed products that costs at most a few 1e-7 — far inside the 4e-6 of slack (kBqSlack) that the
    cutoff leaves on top of the margin."""
    rng = np.random.default_rng(5)
    n = 40_000
    worst = 0.0
    for name, f in catalogues(rng, n):
        q = f[rng.integers(0, n)]
        rows = f.astype(np.float32)
        n2 = np.zeros(n, np.float32)
        for j in range(12):
            n2 = n2 + rows[:, j] * rows[:, j]
        valid = (n2 >= np.float32(MIN_NORM2)) & (n2 <= np.float32(MAX_NORM2))
        inv = np.zeros_like(n2)
        inv[valid] = (np.float32(1) / np.sqrt(n2[valid])).astype(np.float32)
        rh = to_f16((rows * inv[:, None]).astype(np.float32), False)[valid]
        qn = np.float32(np.sqrt(np.sum(q.astype(np.float32) ** 2, dtype=np.float32)))
        if not (1.005e-4 <= qn <= 1e18):
            continue
        qh = to_f16((q.astype(np.float32) / qn).astype(np.float32), False)
        exact_sum = rh @ qh                                   # float64: the products and their sum are exact enough
        acc = np.zeros(len(rh), np.float32)
        for j in range(12):                                   # v_fma_mix_f32: fl32(x * y + acc), the product exact
            acc = (acc.astype(np.float64) + rh[:, j] * qh[j]).astype(np.float32)
        if len(rh):
            worst = max(worst, float(np.abs(acc.astype(np.float64) - exact_sum).max()))
    assert worst < 1e-6, worst
