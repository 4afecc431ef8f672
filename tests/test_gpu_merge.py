"""merge_kernel on crafted key lists, through the C-ABI (mi355rec_enqueue_merge_keys / _batch): every branch of the merge
— one load phase over few non-empty lists, the threshold from the list heads, the threshold one level down (from the ends of
the first chunks) when too few lists hold keys, the walk through lists, the exact radix fallback when the survivors
overflow — against the plain definition: sort every key, keep the best topn.  The shapes are the ones scans leave behind:
768 lists of which most are empty (a launch-wide bound), a few dozen long ones (a catalogue of few large clusters: the
case that sent every third launch of a streamed fp32 scan through the fallback in round 5), keys that differ only in
their lowest bits (tight clusters), mass ties."""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def lists_of(rng, n_lists, list_len, fill, lo_bits, hi_bits):
    """fill[l] keys in list l, unique keys, score images in [lo_bits, hi_bits), each list sorted descending, 0-padded"""
    total = int(np.sum(fill))
    scores = rng.integers(lo_bits, hi_bits, size=total, dtype=np.uint64)
    rows = rng.permutation(total).astype(np.uint64)     # unique rows -> unique keys
    keys = (scores << np.uint64(32)) | ((~rows) & np.uint64(0xFFFFFFFF))
    out = np.zeros((n_lists, list_len), dtype=np.uint64)
    at = 0
    for l, f in enumerate(fill):
        out[l, :f] = np.sort(keys[at:at + f])[::-1]
        at += f
    return out


def expected(lists, topn):
    flat = lists.reshape(-1)
    flat = np.sort(flat[flat != 0])[::-1][:topn]
    out = np.zeros(topn, dtype=np.uint64)
    out[:len(flat)] = flat
    return out


# (score images: the ordered image of scores just below 1.0 is just below 0xBF800000; the merge only compares keys)
TIGHT = (0xBF7FF000, 0xBF800000)
WIDE = (0x80000001, 0xBF800000)

CASES = []
for name, n_lists, list_len, topn, fill_fn, rng_bits in [
    # few non-empty lists among many: one load phase
    ("sparse-8-long", 768, 100, 100, lambda r, n: np.where(np.arange(n) % 96 == 5, 100, 0), TIGHT),
    # 65 lists of 30 ... 100 keys (few large clusters): fewer heads than topn, too many keys for one phase -> one level down
    ("deep-65", 768, 100, 100, lambda r, n: np.where(np.arange(n) % 11 == 3, r.integers(30, 101, n), 0), TIGHT),
    # 22 long lists + 40 short ones: neither -> the walk with thr = 1
    ("walk-22", 768, 100, 100, lambda r, n: np.where(np.arange(n) % 35 == 1, 100, np.where(np.arange(n) % 19 == 2, 2, 0)), TIGHT),
    # every list holds keys: heads give the threshold (uniform catalogue)
    ("heads-all", 768, 100, 100, lambda r, n: r.integers(1, 6, n), WIDE),
    ("heads-full-lists", 512, 128, 128, lambda r, n: np.full(n, 128), WIDE),
    # top-10 lists of a small shard
    ("top10", 326, 10, 10, lambda r, n: r.integers(0, 11, n), WIDE),
    # per-rank lists after an all-gather: 8 lists, probe depth > 1
    ("ranks-8", 8, 100, 100, lambda r, n: np.full(n, 100), WIDE),
    ("ranks-3-ragged", 3, 100, 100, lambda r, n: np.array([100, 37, 0]), TIGHT),
    # fewer keys than topn in total
    ("short", 768, 100, 100, lambda r, n: np.where(np.arange(n) % 50 == 0, 3, 0), WIDE),
    ("empty", 64, 100, 100, lambda r, n: np.zeros(n, dtype=np.int64), WIDE),
    # max sizes
    ("max-lists-topn", 2048, 1024, 1024, lambda r, n: np.where(np.arange(n) % 4 == 0, r.integers(0, 1025, n), 0), WIDE),
]:
    CASES.append((name, n_lists, list_len, topn, fill_fn, rng_bits))


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available()
    from spotify_recommender_amd.engine import CosineEngine
    f = np.random.default_rng(1).random((4096, 12), dtype=np.float32)
    with CosineEngine(f) as e:
        yield e


@pytest.mark.parametrize("name,n_lists,list_len,topn,fill_fn,bits", CASES, ids=[c[0] for c in CASES])
def test_merge_of_crafted_lists(eng, name, n_lists, list_len, topn, fill_fn, bits):
    import torch
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    fill = np.minimum(np.asarray(fill_fn(rng, n_lists), dtype=np.int64), list_len)
    lists = lists_of(rng, n_lists, list_len, fill, *bits)
    d = torch.from_numpy(lists.view(np.int64)).cuda()
    out = torch.full((topn,), -7, dtype=torch.int64, device="cuda")
    idx = torch.zeros(topn, dtype=torch.int64, device="cuda")
    sc = torch.zeros(topn, dtype=torch.float32, device="cuda")
    eng.enqueue_merge_keys(d, n_lists, list_len, topn, out, idx, sc)
    torch.cuda.synchronize()
    want = expected(lists, topn)
    got = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want), (name, int(np.argmax(got != want)))
    cnt = int(np.count_nonzero(want))
    assert (idx.cpu().numpy()[cnt:] == -1).all()
    assert np.array_equal(idx.cpu().numpy()[:cnt], ((~want[:cnt]) & np.uint64(0xFFFFFFFF)).astype(np.int64))


def test_merge_fallback_when_survivors_overflow(eng):
    """100 lists full of keys ABOVE every head of the other 668: the 100th largest head lets 10 000 keys through, more
    than the survivor buffer holds -> the exact radix select over the non-empty lists; with keys that share their upper
    bytes (tight cluster) and with 50-fold ties in the score (the row breaks them)."""
    import torch
    rng = np.random.default_rng(5)
    for tie_scores in (False, True):
        n_lists, list_len, topn = 768, 100, 100
        lists = np.zeros((n_lists, list_len), dtype=np.uint64)
        rows = rng.permutation(n_lists * list_len).astype(np.uint64)
        at = 0
        for l in range(n_lists):
            if l % 7 == 0 and l < 700:
                sc = rng.integers(0xBF7FFF00, 0xBF7FFFF0, size=list_len, dtype=np.uint64)
                if tie_scores:
                    sc = np.uint64(0xBF7FFF00) + (sc % np.uint64(2))
            else:
                sc = rng.integers(0xBF000000, 0xBF7FFF00, size=list_len, dtype=np.uint64)
            keys = (sc << np.uint64(32)) | ((~rows[at:at + list_len]) & np.uint64(0xFFFFFFFF))
            lists[l] = np.sort(keys)[::-1]
            at += list_len
        d = torch.from_numpy(lists.view(np.int64)).cuda()
        out = torch.zeros(topn, dtype=torch.int64, device="cuda")
        eng.enqueue_merge_keys(d, n_lists, list_len, topn, out)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64), expected(lists, topn)), tie_scores


def test_batched_merge_layout_with_sparse_and_deep_queries(eng):
    """[list][query][key] layout (what an all-gather of per-rank batch results leaves): queries of one batch take
    different branches of the merge in the same launch."""
    import torch
    rng = np.random.default_rng(9)
    n_lists, batch, topn = 8, 5, 64
    per_query = []
    for b in range(batch):
        fill = [np.full(n_lists, topn), np.array([topn, 0, 0, 0, 3, 0, 0, 1]), np.zeros(n_lists, dtype=np.int64),
                rng.integers(0, topn + 1, n_lists), np.full(n_lists, 1)][b]
        per_query.append(lists_of(rng, n_lists, topn, fill, *(TIGHT if b % 2 else WIDE)))
    buf = np.zeros((n_lists, batch, topn), dtype=np.uint64)
    for b in range(batch):
        buf[:, b, :] = per_query[b]
    d = torch.from_numpy(buf.view(np.int64)).cuda()
    out = torch.zeros((batch, topn), dtype=torch.int64, device="cuda")
    eng.enqueue_merge_keys_batch(d, n_lists, topn, batch * topn, topn, batch, topn, out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    for b in range(batch):
        assert np.array_equal(got[b], expected(per_query[b], topn)), b
