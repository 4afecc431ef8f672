"""The batched matrix-core path (csrc/batched.hip.h, BASELINE configs[4]) against
the oracle and against the single-query path, through the C-ABI.

The fp16 MFMA pre-filter only PROPOSES candidates; every returned score comes
from the exact fp32 chain, so the bar is the same as everywhere else: scores
bit-exact, ids identical (tie-aware).  The diagnostics (candidates per query,
queries handed to the exact scan) are asserted too, so a run that silently
served everything through the fallback would fail.
"""
import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches
from tests.test_gpu_fuzz import make_catalogue

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def Engine(torch_cuda):
    from spotify_recommender_amd.engine import CosineEngine
    return CosineEngine


def mfma(eng):
    from spotify_recommender_amd import capi
    eng.set_batch_path(capi.BATCH_MFMA)
    return eng


def check_batch(eng, f, queries, excl, topn, label, sample=None, threads=0):
    idx, sc, counts = eng.query_batch_topn(queries, excl, topn)
    which = range(len(queries)) if sample is None else sample
    for b in which:
        want = oracle.scores(f, queries[b], threads=threads)
        ex = int(excl[b]) if excl is not None else -1
        try:
            assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, ex, topn,
                                ref_idx=oracle.topn_heap(want, ex, topn))
        except AssertionError as e:
            raise AssertionError(f"{label}: query {b} topn {topn}: {e}") from e
    return idx, sc, counts


@pytest.mark.parametrize("batch,topn", [(13, 1), (64, 10), (200, 100), (1024, 128), (1500, 100)])
def test_uniform_catalogue_matches_oracle(Engine, batch, topn):
    rng = np.random.default_rng(batch)
    n = 300_007
    f = rng.random((n, 12), dtype=np.float32)
    qrows = rng.integers(0, n, size=batch)
    queries = f[qrows].copy()
    queries[1::2] = rng.random((len(queries[1::2]), 12), dtype=np.float32)
    excl = qrows.astype(np.int64)
    excl[1::2] = -1
    with Engine(f) as eng:
        mfma(eng)
        sample = None if batch <= 200 else list(range(0, batch, 37)) + [batch - 1]
        idx, sc, counts = check_batch(eng, f, queries, excl, topn, f"uniform b{batch}", sample=sample)
        d = eng.batched_last_counters()
        assert d["queued_queries"] == 0 and d["special_rows"] == 0, d
        # gfx950 keeps fp16 subnormals (device self-check on first use): the tight bound applies
        assert abs(eng.stats().batched_margin - 1.0e-3) < 1e-9
        served = batch if batch <= 1024 else batch - 1024       # diagnostics cover the last chunk
        assert d["candidates_total"] >= served * min(topn, 1), d
        assert d["candidates_max"] <= 2048
        # every list equals the single-query path, bit for bit
        for b in range(0, batch, 7):
            si, ss = eng.query_topn(queries[b], int(excl[b]), topn)
            assert si.tolist() == idx[b][:counts[b]].tolist()
            assert np.array_equal(ss.view(np.uint32), sc[b][:counts[b]].view(np.uint32))


def test_hostile_distributions(Engine):
    import os
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "20261004")))
    for case in range(int(os.environ.get("BATCHED_FUZZ_CASES", "14"))):
        rows = int(rng.choice([70_000, 150_000, 400_000]))
        f = make_catalogue(rng, rows)
        batch = int(rng.integers(13, 120))
        topn = int(rng.choice([1, 7, 100, 128]))
        qrows = rng.integers(0, rows, size=batch)
        excl = np.where(rng.random(batch) < 0.8, qrows, -1).astype(np.int64)
        with Engine(f) as eng:
            mfma(eng)
            check_batch(eng, f, f[qrows], excl, topn, f"case {case} rows {rows}", threads=0)


def test_queries_the_bound_cannot_be_claimed_for(Engine):
    """Zero / tiny / huge / non-finite queries, and a query against which fewer than
    topn+1 groups are positive: served by the exact multi-query scan inside the call."""
    rng = np.random.default_rng(8)
    n = 200_000
    f = rng.random((n, 12), dtype=np.float32)
    f[::4000] *= np.float32(-1)             # only these 50 rows have a positive cosine with query 5
    queries = rng.random((40, 12), dtype=np.float32)
    queries[0] = 0.0
    queries[1] *= np.float32(1e-6)          # |q| below 1.005e-4
    queries[2] *= np.float32(1e19)          # |q| above 1e18
    queries[3, 4] = np.nan
    queries[4, 0] = np.inf
    queries[5] = -np.abs(queries[5]) - 1    # almost every cosine negative: fewer than topn+1 positive groups
    queries[6] *= np.float32(9e-5)          # |q| ~ 1.8e-4: just inside the claimed range
    excl = np.full(40, -1, dtype=np.int64)
    with Engine(f) as eng:
        mfma(eng)
        # queries 0..4 are always queued; query 5 has 50 rows with a positive cosine, so
        # topn + 1 positive groups exist for top-10 but cannot exist for top-100 / top-128
        for topn, queued in ((10, 5), (100, 6), (128, 6)):
            check_batch(eng, f, queries, excl, topn, "odd queries")
            d = eng.batched_last_counters()
            assert d["queued_queries"] == queued, (topn, d)


@pytest.mark.parametrize("clusters,served", [(100, True), (3, False)])
def test_clustered_catalogue_many_candidates_per_query(Engine, clusters, served):
    """Rows in tight clusters (spread 0.01; what min-max normalised audio features look like far more than uniform noise
    does): every row of a query's cluster passes the fp16 pre-filter.  100 clusters of ~13 000 rows: the per-query lists
    (32 768 rows at this size) hold them and the finalize workgroup works through them in chunks — several cuts, the
    floor of one cut filtering the next chunk; until the end of round 4 the cap was 2048 and all of these queries went
    to the exact queue (43 ms per 1024 queries at 10 M rows instead of 0.8).  3 clusters: past any list, the exact queue."""
    rng = np.random.default_rng(500 + clusters)
    n = 1_300_003
    centres = rng.random((clusters, 12), dtype=np.float32)
    centres[:, 2] = rng.integers(0, 12, clusters).astype(np.float32) / np.float32(11)
    centres[:, 4] = rng.integers(0, 2, clusters).astype(np.float32)
    which = rng.integers(0, clusters, n)
    noise = (rng.standard_normal((n, 12)) * 0.01).astype(np.float32)
    noise[:, [2, 4]] = 0
    f = np.clip(centres[which] + noise, 0, 1).astype(np.float32)
    dup = rng.integers(0, n, n // 50)
    f[dup] = f[rng.integers(0, n, n // 50)]            # exact duplicates: ties inside the lists
    batch = 96
    qrows = rng.integers(0, n, size=batch)
    queries = f[qrows].copy()
    excl = qrows.astype(np.int64)
    with Engine(f) as eng:
        mfma(eng)
        for topn in (100, 10):
            check_batch(eng, f, queries, excl, topn, f"{clusters} clusters", sample=range(0, batch, 7))
            d = eng.batched_last_counters()
            if served:
                assert d["queued_queries"] == 0 and 2048 < d["candidates_max"] <= 32768, d
            else:
                assert d["queued_queries"] == batch, d


def test_candidate_overflow_and_special_rows(Engine):
    """(a) every row identical: each query has 90 000 candidates at its threshold ->
    all queries overflow into the exact scan; (b) more special rows (tiny norms,
    inf, NaN) than the special list holds -> the whole chunk goes to the exact scan;
    (c) a few special rows: listed and scored exactly inside the batched path."""
    same = np.tile(np.linspace(0.05, 0.95, 12, dtype=np.float32), (90_000, 1))
    with Engine(same) as eng:
        mfma(eng)
        idx, sc, counts = eng.query_batch_topn(same[:20], np.arange(20), 64)
        for b in range(20):
            assert idx[b].tolist() == [i for i in range(65) if i != b][:64]
        assert eng.batched_last_counters()["queued_queries"] == 20
    rng = np.random.default_rng(12)
    n = 150_000
    f = rng.random((n, 12), dtype=np.float32)
    many = rng.choice(n, size=3000, replace=False)
    f[many[:1000]] *= np.float32(1e-5)
    f[many[1000:2000], 3] = np.inf
    f[many[2000:], 7] = np.nan
    queries = f[rng.integers(0, n, size=30)].copy()
    queries[:10] = rng.random((10, 12), dtype=np.float32) * np.float32(1e-3)   # tiny rows matter for tiny queries
    with Engine(f) as eng:
        mfma(eng)
        check_batch(eng, f, queries, None, 100, "many special rows")
        d = eng.batched_last_counters()
        assert d["special_rows"] > 1024 and d["queued_queries"] == 30, d
    g = rng.random((n, 12), dtype=np.float32)
    few = rng.choice(n, size=300, replace=False)
    g[few[:100]] *= np.float32(3e-5)
    g[few[100:200], 1] = -np.inf
    g[few[200:250]] = np.float32(3e19)
    g[few[250:], 0] = np.nan
    g[rng.choice(n, size=5000, replace=False)] = 0.0       # zero rows are skipped, not listed
    queries = g[rng.integers(0, n, size=30)].copy()
    queries[:6] = rng.random((6, 12), dtype=np.float32) * np.float32(3e-4)     # |q| ~ 6e-4: |q||row| straddles 1e-8 for the 3e-5 rows
    with Engine(g) as eng:
        mfma(eng)
        check_batch(eng, g, queries, None, 100, "few special rows")
        d = eng.batched_last_counters()
        assert 250 <= d["special_rows"] <= 300, d


def test_device_resident_queries_and_sharded_layout(Engine, torch_cuda):
    """mi355rec_enqueue_batch_keys_dev on two shards with row_base: the [rank][query][key]
    buffer goes through the batched merge and equals the whole-catalogue result."""
    torch = torch_cuda
    from spotify_recommender_amd.engine import shard_bounds, unpack_keys
    rng = np.random.default_rng(31)
    n, batch, topn = 500_003, 96, 50
    f = rng.random((n, 12), dtype=np.float32)
    t = torch.from_numpy(f).cuda()
    qrows = rng.integers(0, n, size=batch)
    qd = t[torch.from_numpy(qrows).cuda()].contiguous()
    ed = torch.from_numpy(qrows.astype(np.int64)).cuda()
    parts = 2
    shards = [Engine(t[lo:hi], row_base=lo) for lo, hi in (shard_bounds(n, parts, r) for r in range(parts))]
    gathered = torch.zeros(parts * batch * topn, dtype=torch.int64, device="cuda")
    for r, sh in enumerate(shards):
        sh.enqueue_batch_keys_dev(qd, ed, topn, gathered[r * batch * topn:(r + 1) * batch * topn])
    out_keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
    out_idx = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
    out_score = torch.zeros(batch * topn, dtype=torch.float32, device="cuda")
    shards[0].enqueue_merge_keys_batch(gathered, parts, topn, batch * topn, topn, batch, topn, out_keys, out_idx, out_score)
    torch.cuda.synchronize()
    gi = out_idx.cpu().numpy().reshape(batch, topn)
    gs = out_score.cpu().numpy().reshape(batch, topn)
    for b in range(batch):
        want = oracle.scores(f, f[qrows[b]], threads=0)
        assert_topn_matches(gi[b], gs[b], want, int(qrows[b]), topn, ref_idx=oracle.topn_heap(want, int(qrows[b]), topn))
    for sh in shards:
        assert sh.batched_last_counters()["queued_queries"] == 0
        sh.close()


def test_config5_shard_1024_queries_every_list(Engine, torch_cuda):
    """BASELINE configs[4] as one of its 8 GPUs sees it: a 12.5 M-row shard (row_base of
    rank 3), ONE 1024-query batch, top-100.  Every list must equal the single-query
    path; sampled queries go against the oracle; the pre-filter must have served all
    of them with a few hundred candidates each."""
    torch = torch_cuda
    from spotify_recommender_amd.engine import unpack_keys
    from spotify_recommender_amd.synth import synthetic_catalogue
    n_local, base, batch, topn = 12_500_000, 3 * 12_500_000, 1024, 100
    t = synthetic_catalogue(n_local, seed=5)
    f = t.cpu().numpy()
    rng = np.random.default_rng(55)
    local_rows = rng.integers(0, n_local, size=batch)
    queries = f[local_rows].copy()
    queries[1::2] = rng.random((batch // 2, 12), dtype=np.float32)
    excl = (local_rows + base).astype(np.int64)
    excl[1::2] = -1
    keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
    single = torch.zeros(topn, dtype=torch.int64, device="cuda")
    with Engine(t, row_base=base) as eng:
        eng.enqueue_batch_keys(queries, excl, topn, keys)       # AUTO routes 1024 queries to the batched path
        torch.cuda.synchronize()
        d = eng.batched_last_counters()
        assert d["queued_queries"] == 0 and d["special_rows"] == 0, d
        assert batch * topn <= d["candidates_total"] <= batch * 24 * topn, d
        got = keys.cpu().numpy().reshape(batch, topn)
        for b in range(batch):
            eng.enqueue_query_keys(queries[b], int(excl[b]), topn, single)
            assert np.array_equal(single.cpu().numpy(), got[b]), f"query {b} differs from the single-query path"
        for b in (0, 1, 510, 1023):
            want = oracle.scores(f, queries[b], threads=0)
            rows, scores = unpack_keys(got[b])
            ex = int(excl[b]) - base if excl[b] >= 0 else -1
            assert_topn_matches(rows - base, scores, want, ex, topn, ref_idx=oracle.topn_heap(want, ex, topn))


def test_batched_call_replays_from_a_hip_graph(Engine, torch_cuda):
    """The asynchronous batched call allocates nothing and never synchronises after its
    first use on a handle, so it can be captured into a hipGraph (here through torch's
    CUDAGraph on a side stream) and replayed with new queries in the same device buffers."""
    torch = torch_cuda
    from spotify_recommender_amd.engine import unpack_keys
    rng = np.random.default_rng(77)
    n, batch, topn = 400_000, 200, 20
    f = rng.random((n, 12), dtype=np.float32)
    t = torch.from_numpy(f).cuda()
    side = torch.cuda.Stream()
    qd = torch.zeros((batch, 12), dtype=torch.float32, device="cuda")
    ed = torch.full((batch,), -1, dtype=torch.int64, device="cuda")
    keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
    with Engine(t) as eng:
        rows0 = rng.integers(0, n, size=batch)
        qd.copy_(t[torch.from_numpy(rows0).cuda()])
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            eng.enqueue_batch_keys_dev(qd, ed, topn, keys)     # first call: allocates, on the capture stream
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            eng.enqueue_batch_keys_dev(qd, ed, topn, keys)
        for trial in range(2):
            rows = rng.integers(0, n, size=batch)
            qd.copy_(t[torch.from_numpy(rows).cuda()])
            ed.copy_(torch.from_numpy(rows.astype(np.int64)).cuda())
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            got = keys.cpu().numpy().reshape(batch, topn)
            for b in range(0, batch, 17):
                want = oracle.scores(f, f[rows[b]])
                r_, s_ = unpack_keys(got[b])
                assert_topn_matches(r_, s_, want, int(rows[b]), topn, ref_idx=oracle.topn_heap(want, int(rows[b]), topn))


def test_fp32_sourced_passes_give_the_same_keys(Engine, torch_cuda):
    """By default the passes read the fp16 replica; with the replica switched off they normalise and
    convert the fp32 rows themselves.  Same arithmetic by construction: same candidates, same keys."""
    torch = torch_cuda
    rng = np.random.default_rng(91)
    n, batch, topn = 500_003, 96, 50
    f = rng.random((n, 12), dtype=np.float32)
    f[::5000] *= np.float32(1e-7)                  # some rows the bound is not claimed for
    f[7::9000] = 0.0
    qrows = rng.integers(0, n, size=batch)
    queries = f[qrows].copy()
    excl = qrows.astype(np.int64)
    keys = {}
    cands = {}
    with Engine(f) as eng:
        mfma(eng)
        for mode in (0, 1):                         # REPLICA_AUTO (replica-sourced), REPLICA_OFF (fp32-sourced)
            eng.set_replica(mode)
            k = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys(queries, excl, topn, k)
            torch.cuda.synchronize()
            keys[mode] = k.cpu().numpy()
            cands[mode] = eng.batched_last_counters()
        assert np.array_equal(keys[0], keys[1])
        assert cands[0] == cands[1], (cands[0], cands[1])
        assert cands[0]["special_rows"] > 0
        eng.set_replica(0)
        check_batch(eng, f, queries, excl, topn, "replica-sourced", sample=range(0, batch, 11))
