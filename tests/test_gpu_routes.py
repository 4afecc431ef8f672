"""Which route a call takes under AUTO, cell by cell (DESIGN.md §6 "Routes"): every
cell makes ONE call and asserts the route counter of mi355rec_stats_t that moved — and that no other did.
(seven scan routes + the two-pass matrix-core path, and nothing reported which one a call took.)
Results are checked against the oracle for one query per cell: a route that is taken must also be right."""
import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches

pytestmark = pytest.mark.gpu

ROUTES = ("route_fp32", "route_fp16", "route_q8", "route_q8_lone", "route_multi_fp32", "route_multi_fp16",
          "route_multi_q8", "route_mfma_two_pass")


def routes(eng):
    st = eng.stats()
    return {r: int(getattr(st, r)) for r in ROUTES}


@pytest.fixture(scope="module")
def catalogues():
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(77)
    return {n: rng.random((n, 12), dtype=np.float32) for n in (30_000, 300_000, 1_100_000, 1_500_000, 4_500_000)}


# (rows, how, batch, topn) -> {route: launches}.  `how`: sync = mi355rec_query_row_topn, keys = _enqueue_row_keys,
# stream = one streamed query + flush, batch = mi355rec_query_batch_topn, dev = _enqueue_batch_keys_dev
CELLS = [
    # single queries: the fp32 rows below 1 M rows; the 8-bit replica from there on; a lone synchronous query on a
    # shard of >= 4 M rows in ONE launch that also merges and signals
    (30_000, "sync", 1, 10, {"route_fp32": 1}),
    (300_000, "sync", 1, 100, {"route_fp32": 1}),
    (300_000, "stream", 1, 100, {"route_fp32": 1}),
    (1_100_000, "sync", 1, 10, {"route_q8": 1}),
    (1_100_000, "keys", 1, 10, {"route_q8": 1}),
    (1_100_000, "stream", 1, 10, {"route_q8": 1}),
    (1_500_000, "sync", 1, 100, {"route_q8": 1}),
    (1_500_000, "keys", 1, 1000, {"route_q8": 1}),
    (1_500_000, "stream", 1, 100, {"route_q8": 1}),
    (4_500_000, "sync", 1, 100, {"route_q8_lone": 1}),
    (4_500_000, "keys", 1, 100, {"route_q8": 1}),
    (1_500_000, "sync", 1, 2500, {"route_q8": 1, "route_fp32": 2}),       # topn > 1024: rounds; the later ones read the fp32 rows
    # batches on a shard WITHOUT a replica (< 65536 rows): exact 12-query passes, whatever the count
    (30_000, "batch", 2, 10, {"route_multi_fp32": 1}),
    (30_000, "batch", 40, 10, {"route_multi_fp32": 4}),
    # batches with a replica: 2 ... 32 queries -> one pass over the fp16 replica; 33 and more -> the two-pass matrix-core
    # path (chunks of 1024)
    (300_000, "batch", 2, 100, {"route_multi_fp16": 1}),
    (300_000, "batch", 3, 100, {"route_multi_fp16": 1}),
    (300_000, "batch", 16, 100, {"route_multi_fp16": 1}),
    (300_000, "batch", 17, 100, {"route_multi_fp16": 1}),
    (300_000, "batch", 32, 100, {"route_multi_fp16": 1}),
    (300_000, "batch", 33, 100, {"route_mfma_two_pass": 1}),
    (1_500_000, "batch", 1200, 50, {"route_mfma_two_pass": 2}),
    (300_000, "dev", 64, 100, {"route_mfma_two_pass": 1}),
    # topn above 128: no multi-query pass holds that many keys per query -> one scan per query
    (300_000, "batch", 3, 200, {"route_fp32": 3}),
    (1_500_000, "batch", 3, 200, {"route_q8": 3}),
]


@pytest.mark.parametrize("rows,how,batch,topn,want", CELLS, ids=[f"{c[0]}-{c[1]}-{c[2]}-top{c[3]}" for c in CELLS])
def test_auto_route(catalogues, rows, how, batch, topn, want):
    import torch
    from spotify_recommender_amd.engine import CosineEngine, unpack_keys
    f = catalogues[rows]
    rng = np.random.default_rng(rows + batch)
    qrows = rng.integers(0, rows, size=batch)
    with CosineEngine(f) as eng:
        before = routes(eng)
        if how == "sync":
            idx, sc = eng.query_row_topn(int(qrows[0]), topn)
        elif how in ("keys", "stream"):
            keys = torch.zeros(topn, dtype=torch.int64, device="cuda")
            if how == "keys":
                eng.enqueue_row_keys(int(qrows[0]), topn, keys)
            else:
                eng.enqueue_row_keys_streamed(int(qrows[0]), topn, keys)
                eng.enqueue_flush()
            torch.cuda.synchronize()
            idx, sc = unpack_keys(keys.cpu().numpy())
        elif how == "batch":
            bi, bs, counts = eng.query_batch_topn(f[qrows], qrows, topn)
            idx, sc = bi[0][:counts[0]], bs[0][:counts[0]]
        else:
            qd = torch.from_numpy(f[qrows]).cuda()
            ed = torch.from_numpy(qrows).cuda()
            keys = torch.zeros(batch * topn, dtype=torch.int64, device="cuda")
            eng.enqueue_batch_keys_dev(qd, ed, topn, keys)
            torch.cuda.synchronize()
            idx, sc = unpack_keys(keys[:topn].cpu().numpy())
        after = routes(eng)
        moved = {r: after[r] - before[r] for r in ROUTES if after[r] != before[r]}
        assert moved == want, moved
        s = oracle.scores(f, f[qrows[0]], threads=0)
        assert_topn_matches(idx, sc, s, int(qrows[0]), topn)
        assert eng.stats().route_exact_queue == 0


def test_forced_routes_and_the_exact_queue(catalogues):
    """The A/B knobs move the counters they name, and queries the matrix-core path cannot serve are counted where they
    go: to the exact scan, on the device."""
    from spotify_recommender_amd import capi
    from spotify_recommender_amd.engine import CosineEngine
    f = catalogues[1_500_000]
    rng = np.random.default_rng(5)
    with CosineEngine(f) as eng:
        r0 = routes(eng)
        exp = capi.has_experiments()   # the A/B routes of MI355REC_EXPERIMENTS builds: refused by the product library
        if exp:
            eng.set_replica(capi.REPLICA_FP16)
            eng.query_row_topn(7, 10)
        else:
            with pytest.raises(capi.Mi355Error):
                eng.set_replica(capi.REPLICA_FP16)
            with pytest.raises(capi.Mi355Error):
                eng.set_batch_path(capi.BATCH_Q8)
        eng.set_replica(capi.REPLICA_OFF)
        eng.query_row_topn(7, 10)
        eng.set_replica(capi.REPLICA_AUTO)
        if exp:
            eng.set_batch_path(capi.BATCH_Q8)
            eng.query_batch_topn(f[[1, 2, 3, 4, 5]], np.array([1, 2, 3, 4, 5]), 10)
        eng.set_batch_path(capi.BATCH_MULTI)
        eng.query_batch_topn(f[[1, 2, 3]], np.array([1, 2, 3]), 10)
        eng.set_batch_path(capi.BATCH_MFMA_NOSKIP)
        eng.query_batch_topn(f[[1, 2, 3]], np.array([1, 2, 3]), 10)
        r1 = routes(eng)
        moved = {r: r1[r] - r0[r] for r in ROUTES if r1[r] != r0[r]}
        want = {"route_fp32": 1, "route_multi_fp32": 1, "route_mfma_two_pass": 1}
        if exp:
            want.update({"route_fp16": 1, "route_multi_q8": 1})
        assert moved == want, moved
        # a zero query and a huge one cannot claim the pre-filter's bound: the matrix-core path queues them
        eng.set_batch_path(capi.BATCH_MFMA)
        q = f[rng.integers(0, len(f), size=40)].copy()
        q[3] = 0.0
        q[17] = np.float32(1e30)
        bi, bs, counts = eng.query_batch_topn(q, np.full(40, -1), 10)
        assert eng.stats().route_exact_queue == 2
        for b in (3, 17, 20):
            assert_topn_matches(bi[b][:counts[b]], bs[b][:counts[b]], oracle.scores(f, q[b], threads=0), -1, 10)


def test_stats_for_a_caller_built_against_a_shorter_struct(catalogues):
    """mi355rec_stats_sized copies no more than the caller's struct holds (mi355rec_stats_t grew without a size
    field): the first 64 bytes equal those of the full struct, the bytes behind them stay untouched."""
    import ctypes
    from spotify_recommender_amd import capi
    from spotify_recommender_amd.engine import CosineEngine
    with CosineEngine(catalogues[30_000]) as eng:
        eng.query_row_topn(5, 10)
        full = eng.stats()
        size = ctypes.sizeof(full)
        buf = (ctypes.c_ubyte * (size + 64))(*([0xAB] * (size + 64)))
        wrote = ctypes.c_size_t(0)
        capi.check(capi.lib().mi355rec_stats_sized(eng._h, buf, 64, ctypes.byref(wrote)), eng._h)
        assert wrote.value == 64
        assert bytes(buf[:64]) == bytes(full)[:64] and set(buf[64:]) == {0xAB}
        capi.check(capi.lib().mi355rec_stats_sized(eng._h, buf, size + 64, ctypes.byref(wrote)), eng._h)
        assert wrote.value == size and set(buf[size:]) == {0xAB}
        assert capi.lib().mi355rec_stats_sized(eng._h, None, 64, None) == capi.ERR_INVALID_ARG
