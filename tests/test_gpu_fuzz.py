"""Randomised parity sweep: many small/medium catalogues with hostile score
distributions (heavy ties, clusters, signed, sparse special values), random
topN and batch sizes, every result checked against the oracle.  FUZZ_CASES
(environment) scales the number of cases; the default keeps the test short."""
import os

import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches

pytestmark = pytest.mark.gpu


def make_catalogue(rng, rows):
    kind = rng.integers(0, 7)
    if kind == 0:
        f = rng.random((rows, 12), dtype=np.float32)
    elif kind == 1:   # few distinct values per feature -> massive exact ties
        f = (rng.integers(0, 3, size=(rows, 12)) / np.float32(2)).astype(np.float32)
    elif kind == 2:   # tight clusters around a handful of centres
        centres = rng.random((5, 12), dtype=np.float32)
        f = centres[rng.integers(0, 5, size=rows)] * (1 + rng.normal(0, 1e-6, size=(rows, 12))).astype(np.float32)
    elif kind == 3:   # signed, wide dynamic range
        f = (rng.normal(0, 1, size=(rows, 12)) * 10.0 ** rng.integers(-3, 4, size=(rows, 1))).astype(np.float32)
    elif kind == 4:   # sorted by similarity to the all-ones vector (ascending): adversarial for thresholds
        t = np.sort(rng.random(rows)).astype(np.float32)[:, None]
        f = np.ones((rows, 12), np.float32)
        f[:, :6] = 0.1 + 0.9 * t
    elif kind == 5:   # mostly duplicates of a few rows
        base = rng.random((3, 12), dtype=np.float32)
        f = base[rng.integers(0, 3, size=rows)].copy()
        k = max(1, rows // 50)
        f[rng.integers(0, rows, size=k)] = rng.random((k, 12), dtype=np.float32)
    else:             # sparse rows + special values
        f = rng.random((rows, 12), dtype=np.float32)
        f[rng.random((rows, 12)) < 0.6] = 0.0
        for val in (np.nan, np.inf, -np.inf, 1e-42, 3e19):
            if rows > 4:
                f[rng.integers(0, rows), rng.integers(0, 12)] = val
    return np.ascontiguousarray(f.astype(np.float32))


def test_randomised_parity():
    import torch
    assert torch.cuda.is_available()
    from spotify_recommender_amd.engine import CosineEngine
    cases = int(os.environ.get("FUZZ_CASES", "60"))
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "20251017")))
    for case in range(cases):
        rows = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 513, 2049, 7000, 40_000, 150_000, 600_000],
                              p=[.03, .03, .05, .05, .05, .05, .1, .1, .1, .14, .15, .1, .05]))
        f = make_catalogue(rng, rows)
        with CosineEngine(f) as eng:
            # single queries over the 8-bit replica (forced on: AUTO starts at 1 M rows) / the fp32 rows / AUTO — and, in
            # MI355REC_EXPERIMENTS builds, the fp16 replica in place of the fp32 rows
            from spotify_recommender_amd import capi
            eng.set_replica((2, 3 if capi.has_experiments() else 1, 0)[case % 3])
            for _ in range(3):
                topn = int(rng.choice([1, 2, 10, 100, 128, 129, 1000, 1024, 1025, 2500]))
                q = int(rng.integers(0, rows))
                want = oracle.scores(f, f[q])
                got = eng.scores_row(q)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (case, rows, q)
                idx, sc = eng.query_row_topn(q, topn)
                try:
                    assert_topn_matches(idx, sc, want, q, topn, ref_idx=oracle.topn_heap(want, q, topn))
                except AssertionError as e:  # pragma: no cover - keep the failing case reproducible
                    raise AssertionError(f"case {case}: rows {rows} q {q} topn {topn}: {e}") from e
            batch = int(rng.integers(2, 20))
            topn = int(rng.choice([1, 7, 100, 128]))
            qrows = rng.integers(0, rows, size=batch)
            excl = np.where(rng.random(batch) < 0.8, qrows, -1).astype(np.int64)
            # a short stream of single queries (merge riding, over the replica one call behind): same keys as
            # the synchronous path
            s_topn = int(rng.choice([1, 10, 100, 700]))
            s_rows = [int(r) for r in rng.integers(0, rows, size=3)]
            s_keys = [torch.zeros(s_topn, dtype=torch.int64, device="cuda") for _ in s_rows]
            for r, k in zip(s_rows, s_keys):
                eng.enqueue_row_keys_streamed(r, s_topn, k)
            eng.enqueue_flush()
            torch.cuda.synchronize()
            for r, k in zip(s_rows, s_keys):
                want = oracle.scores(f, f[r])
                got = k.cpu().numpy().view(np.uint64)
                cnt = min(s_topn, rows - 1)
                idx = (~got[:cnt] & np.uint64(0xffffffff)).astype(np.int64)
                try:
                    assert_topn_matches(idx, None, want, r, s_topn, ref_idx=oracle.topn_heap(want, r, s_topn))
                    assert not got[cnt:].any()
                except AssertionError as e:  # pragma: no cover
                    raise AssertionError(f"case {case} streamed: rows {rows} q {r} topn {s_topn}: {e}") from e
            # queries BY VALUE (round 6: their launch-wide bound comes from their anchor's neighbourhood, csrc/handoff.hip.h): a
            # catalogue row passed by value, a perturbed one, noise; nothing excluded or an arbitrary row excluded; alone and streamed
            v_topn = int(rng.choice([1, 10, 100, 300]))
            v_sync_topn = int(rng.choice([v_topn, 1500]))   # (1500: rounds of 1024 through the synchronous by-value call)
            v_q = [f[int(rng.integers(0, rows))].copy(),
                   (f[int(rng.integers(0, rows))] * np.float32(1.0 + 0.01 * rng.standard_normal())).astype(np.float32),
                   rng.random(12, dtype=np.float32)]
            v_ex = [-1, int(rng.integers(0, rows)), -1]
            v_keys = [torch.zeros(v_topn, dtype=torch.int64, device="cuda") for _ in v_q]
            for q, ex, k in zip(v_q, v_ex, v_keys):
                eng.enqueue_query_keys_streamed(q, ex, v_topn, k)
            eng.enqueue_flush()
            torch.cuda.synchronize()
            for q, ex, k in zip(v_q, v_ex, v_keys):
                want = oracle.scores(f, np.ascontiguousarray(q))
                got = k.cpu().numpy().view(np.uint64)
                cnt = min(v_topn, rows - (1 if ex >= 0 else 0))
                idx = (~got[:cnt] & np.uint64(0xffffffff)).astype(np.int64)
                try:
                    assert_topn_matches(idx, None, want, ex, v_topn, ref_idx=oracle.topn_heap(want, ex, v_topn))
                    assert not got[cnt:].any()
                    i2, s2 = eng.query_topn(q, ex, v_sync_topn)
                    assert_topn_matches(i2, s2, want, ex, v_sync_topn, ref_idx=oracle.topn_heap(want, ex, v_sync_topn))
                except AssertionError as e:  # pragma: no cover
                    raise AssertionError(f"case {case} by value: rows {rows} excl {ex} topn {v_topn}: {e}") from e
            # the same kind of stream dealt over two LANES of the handle (shared rows and replicas, own stream state), each on
            # the stream the library created with it
            if case % 3 == 0:
                lane = eng.lane()
                pair, pstreams = [eng, lane], [eng.own_stream(), lane.own_stream()]
                l_rows = [int(r) for r in rng.integers(0, rows, size=5)]
                l_keys = [torch.zeros(s_topn, dtype=torch.int64, device="cuda") for _ in l_rows]
                torch.cuda.synchronize()
                for i, (r, k) in enumerate(zip(l_rows, l_keys)):
                    pair[i % 2].enqueue_row_keys_streamed(r, s_topn, k, stream=pstreams[i % 2])
                for e, ps in zip(pair, pstreams):
                    e.enqueue_flush(stream=ps)
                torch.cuda.synchronize()
                lane.close()
                for r, k in zip(l_rows, l_keys):
                    want = oracle.scores(f, f[r])
                    got = k.cpu().numpy().view(np.uint64)
                    cnt = min(s_topn, rows - 1)
                    idx = (~got[:cnt] & np.uint64(0xffffffff)).astype(np.int64)
                    try:
                        assert_topn_matches(idx, None, want, r, s_topn, ref_idx=oracle.topn_heap(want, r, s_topn))
                        assert not got[cnt:].any()
                    except AssertionError as e:  # pragma: no cover
                        raise AssertionError(f"case {case} lanes: rows {rows} q {r} topn {s_topn}: {e}") from e
            eng.set_batch_path(1 if case % 2 else 0)   # odd cases: the exact multi-query pass; even: AUTO
            idx, sc, counts = eng.query_batch_topn(f[qrows], excl, topn)
            for b in range(batch):
                want = oracle.scores(f, f[qrows[b]])
                try:
                    assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, int(excl[b]), topn)
                except AssertionError as e:  # pragma: no cover
                    raise AssertionError(f"case {case} batch {b}: rows {rows} q {qrows[b]} excl {excl[b]} topn {topn}: {e}") from e


def test_two_handles_on_two_streams_share_one_matrix():
    """Concurrency the documented way: one handle per stream over the same
    borrowed device matrix; interleaved queries must not disturb each other."""
    import torch
    from spotify_recommender_amd.engine import CosineEngine, unpack_keys
    rng = np.random.default_rng(9)
    f = rng.random((500_000, 12), dtype=np.float32)
    t = torch.from_numpy(f).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    rows = rng.integers(0, f.shape[0], size=24).tolist()
    with CosineEngine(t) as a, CosineEngine(t) as b:
        out = torch.zeros(len(rows) * 50, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        for i, r in enumerate(rows):
            eng, st = (a, s1) if i % 2 == 0 else (b, s2)
            eng.enqueue_row_keys(r, 50, out[i * 50:(i + 1) * 50], stream=st)
        torch.cuda.synchronize()
        got = out.cpu().numpy().reshape(len(rows), 50)
        for i, r in enumerate(rows):
            want = oracle.scores(f, f[r])
            ci, _ = oracle.topn_canonical(want, r, 50)
            assert unpack_keys(got[i])[0].tolist() == ci.tolist()


def test_randomised_multi_query_and_sharded_streams():
    """The round-3 paths under the same hostile catalogues: the multi-query pass over the fp16 replica
    (forced: MI355REC_BATCH_HALF), streams of batches (merge + next sample riding in the launch), and the
    row-sharded stream with worker threads in both window modes — random shard counts, windows, batch sizes
    and topn, special values included; every result against the oracle."""
    import torch
    assert torch.cuda.is_available()
    from spotify_recommender_amd.engine import CosineEngine, NodeEngine, unpack_keys
    cases = int(os.environ.get("FUZZ_CASES2", "14"))
    rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "20261004")))
    for case in range(cases):
        rows = int(rng.choice([66_000, 90_000, 150_000, 400_000, 1_100_000], p=[.2, .25, .25, .2, .1]))
        f = make_catalogue(rng, rows)
        topn = int(rng.choice([1, 5, 50, 100, 128]))
        # ---- one handle: forced multi-query passes, then a stream of batches
        with CosineEngine(f) as eng:
            from spotify_recommender_amd import capi
            # the multi-query pass forced: rows from the fp16 replica (experiment builds: every other case from the 8-bit one)
            eng.set_batch_path(3 + (case % 2 if capi.has_experiments() else 0))
            batch = int(rng.integers(2, 45))
            qrows = rng.integers(0, rows, size=batch)
            queries = f[qrows].copy()
            excl = np.where(rng.random(batch) < 0.8, qrows, -1).astype(np.int64)
            if batch > 3:
                queries[3] = rng.normal(0, 1, 12).astype(np.float32)
                excl[3] = -1
            idx, sc, counts = eng.query_batch_topn(queries, excl, topn)
            for b in range(batch):
                want = oracle.scores(f, queries[b])
                try:
                    assert_topn_matches(idx[b][:counts[b]], sc[b][:counts[b]], want, int(excl[b]), topn)
                except AssertionError as e:  # pragma: no cover
                    raise AssertionError(f"case {case} multi-query pass: rows {rows} batch {batch} b {b} topn {topn}: {e}") from e
            eng.set_batch_path(0)
            outs = []
            for _ in range(4):
                nb = int(rng.integers(1, 34))
                br = rng.integers(0, rows, size=nb)
                keys = torch.zeros(nb * topn, dtype=torch.int64, device="cuda")
                eng.enqueue_batch_keys_streamed(f[br], br.astype(np.int64), topn, keys)
                outs.append((br, keys))
            eng.enqueue_flush()
            torch.cuda.synchronize()
            for br, keys in outs:
                got = keys.cpu().numpy().reshape(len(br), topn)
                for b, r in enumerate(br):
                    want = oracle.scores(f, f[r])
                    i2, s2 = unpack_keys(got[b])
                    try:
                        assert_topn_matches(i2, s2, want, int(r), topn)
                    except AssertionError as e:  # pragma: no cover
                        raise AssertionError(f"case {case} stream of batches: rows {rows} q {r} topn {topn}: {e}") from e
        # ---- the sharded stream, both window modes
        shards = int(rng.choice([1, 2, 3, 5]))
        window = int(rng.choice([1, 3, 8, 16, 40]))
        with NodeEngine(f, devices=[0] * shards) as node:
            node.set_window(window)
            for batched in (True, False):
                node.set_window_mode(batched)
                nq = int(rng.integers(1, 2 * window + 3))
                qr = rng.integers(0, rows, size=nq).tolist()
                tickets = [node.enqueue_row(r, topn) for r in qr]
                vec = rng.random(12, dtype=np.float32)
                tv = node.enqueue_query(vec, -1, topn)
                if rng.random() < 0.5:
                    node.enqueue_flush()
                keep = qr[-2 * window:] if window > 1 else qr[-2:]          # what the ring of 4 windows certainly still holds
                for t, r in list(zip(tickets, qr))[-len(keep):]:
                    i3, s3 = node.wait(t, topn)
                    want = oracle.scores(f, f[r])
                    try:
                        assert_topn_matches(i3, s3, want, r, topn)
                    except AssertionError as e:  # pragma: no cover
                        raise AssertionError(f"case {case} sharded stream: rows {rows} shards {shards} window {window} batched {batched} q {r}: {e}") from e
                i3, s3 = node.wait(tv, topn)
                assert_topn_matches(i3, s3, oracle.scores(f, vec), -1, topn)
            # and one synchronous query through the workers
            r = int(rng.integers(0, rows))
            i4, s4 = node.query_row_topn(r, topn)
            assert_topn_matches(i4, s4, oracle.scores(f, f[r]), r, topn)
