"""Lanes (mi355rec_create_lane): further handles over the SAME device rows and replicas, each with its own stream state.  Results
of every lane are the oracle's on every route; lanes run interleaved on their own streams; the group outlives its parent."""
import numpy as np
import pytest

from oracle import oracle
from tests.parity import assert_topn_matches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def catalogue():
    import torch
    assert torch.cuda.is_available()
    rng = np.random.default_rng(123)
    f = rng.random((1_300_000, 12), dtype=np.float32)      # >= 1 M rows: single queries stream the 8-bit replica
    return f, torch.from_numpy(f).cuda()


def check_keys(f, row, keys, topn):
    from spotify_recommender_amd.engine import unpack_keys
    idx, sc = unpack_keys(keys)
    assert_topn_matches(idx, sc, oracle.scores(f, f[row]), row, topn)


def test_lanes_interleaved_on_their_own_streams(catalogue):
    import torch
    from spotify_recommender_amd.engine import CosineEngine
    f, t = catalogue
    topn = 50
    with CosineEngine(t) as eng:
        lanes = [eng, eng.lane(), eng.lane()]
        streams = [ln.own_stream() for ln in lanes]
        st0 = eng.stats()
        rows = [(k * 7919 + 5) % f.shape[0] for k in range(30)]
        outs = [torch.zeros(topn, dtype=torch.int64, device="cuda") for _ in rows]
        torch.cuda.synchronize()
        for k, r in enumerate(rows):
            lanes[k % 3].enqueue_row_keys_streamed(r, topn, outs[k], stream=streams[k % 3])
        for ln, s in zip(lanes, streams):
            ln.enqueue_flush(stream=s)
        torch.cuda.synchronize()
        for k, r in enumerate(rows):
            check_keys(f, r, outs[k].cpu().numpy(), topn)
        # every lane took the 8-bit route over the SHARED replica (nothing was rebuilt: a lane reports no build time)
        for ln in lanes[1:]:
            st = ln.stats()
            assert st.route_q8 == 10 and st.replica_active == 1 and st.replica_build_ms == 0.0
        assert lanes[0].stats().route_q8 - st0.route_q8 == 10
        # the other routes of a lane: a query alone, a multi-query pass, a 1024-query batch, the fp32 rows
        ln = lanes[1]
        idx, sc = ln.query_row_topn(rows[3], topn)
        assert_topn_matches(idx, sc, oracle.scores(f, f[rows[3]]), rows[3], topn)
        qr = np.array(rows[:12], dtype=np.int64)
        bi, bs, cnt = ln.query_batch_topn(f[qr], qr, topn)
        for b in range(12):
            assert_topn_matches(bi[b][:cnt[b]], bs[b][:cnt[b]], oracle.scores(f, f[qr[b]]), int(qr[b]), topn)
        qr = (np.arange(70, dtype=np.int64) * 104729) % f.shape[0]
        bi, bs, cnt = lanes[2].query_batch_topn(f[qr], qr, topn)
        for b in (0, 33, 69):
            assert_topn_matches(bi[b][:cnt[b]], bs[b][:cnt[b]], oracle.scores(f, f[qr[b]]), int(qr[b]), topn)
        from spotify_recommender_amd import capi
        ln.set_replica(capi.REPLICA_OFF)
        idx, sc = ln.query_row_topn(rows[4], topn)
        assert_topn_matches(idx, sc, oracle.scores(f, f[rows[4]]), rows[4], topn)
        for extra in lanes[1:]:
            extra.close()


def test_the_group_outlives_its_parent_and_refuses_a_rebuild(catalogue):
    import torch
    from spotify_recommender_amd import capi
    from spotify_recommender_amd.engine import CosineEngine
    f, _ = catalogue
    topn = 10
    eng = CosineEngine(f[:1_050_000])            # host rows: the parent OWNS the device copy
    lane = eng.lane()
    with pytest.raises(capi.Mi355Error):
        eng.rebuild_replica()                    # the replicas are shared
    row = 777_777
    want = oracle.scores(f[:1_050_000], f[row])
    idx, sc = lane.query_row_topn(row, topn)
    assert_topn_matches(idx, sc, want, row, topn)
    eng.close()                                  # the parent goes first: rows and replicas stay with the lane
    out = torch.zeros(topn, dtype=torch.int64, device="cuda")
    lane.enqueue_row_keys_streamed(row, topn, out)
    lane.enqueue_flush()
    torch.cuda.synchronize()
    check_keys(f[:1_050_000], row, out.cpu().numpy(), topn)
    lane2 = lane.lane()                          # a lane of a lane: the same group
    idx, sc = lane2.query_row_topn(row, topn)
    assert_topn_matches(idx, sc, want, row, topn)
    lane.close()
    idx, sc = lane2.query_row_topn(row + 1, topn)
    assert_topn_matches(idx, sc, oracle.scores(f[:1_050_000], f[row + 1]), row + 1, topn)
    lane2.close()


def test_a_lane_gets_a_stream_that_runs_beside_its_parents(catalogue):
    """mi355rec_create_lane times a small kernel on the parent's stream alone and on both streams at once and goes through
    streams until the pair overlaps (a process has four hardware queues; streams that share one serialise): every lane of a
    group of five reports a stream that overlaps the parent's, found within six attempts — and a torch-pool stream pair made
    here, with nine streams alive, is allowed to collide (that is the trap this guards against)."""
    from spotify_recommender_amd.engine import CosineEngine
    _, t = catalogue
    with CosineEngine(t) as eng:
        assert eng.lane_status() == {"stream_attempts": 0, "overlaps_parent": -1}
        lanes = [eng.lane() for _ in range(4)]
        stats = [ln.lane_status() for ln in lanes]
        for ln in lanes:
            ln.close()
        for st in stats:
            assert st["overlaps_parent"] in (0, 1) and 1 <= st["stream_attempts"] <= 6, st
        if not all(st["overlaps_parent"] == 1 for st in stats):
            # an environment in which no two streams of the process run side by side (one hardware queue, a tracer that serialises
            # dispatches): the library said so, which is what it is for — lanes buy nothing here, and nothing is wrong with them
            pytest.xfail(f"no stream runs beside the parent's on this box: {stats}")


def test_a_group_without_replicas_builds_them_on_demand_and_lanes_inherit_the_modes():
    """ADVICE r5 (medium).  A shard under 65 536 rows gets no replica at create.  (1) Any member of a group of lanes may then
    build it (mi355rec_set_replica(ON)); the other members take the group's over when they are told the same; once the group has
    replicas a rebuild is refused as before.  (2) A lane starts in its parent's replica mode and batch path.  (3) The node handle:
    REPLICATED {0, 0} (the second replica is a lane of the first) at 30 000 rows, mi355rec_sharded_set_replica(ON), queries."""
    import torch
    from spotify_recommender_amd import capi
    from spotify_recommender_amd.engine import CosineEngine, NodeEngine
    rng = np.random.default_rng(77)
    n, topn = 30_000, 20
    f = rng.random((n, 12), dtype=np.float32)
    f[n - 2] = f[5]
    with CosineEngine(f) as eng:
        assert eng.stats().device_bytes_per_row == 48            # no replica at this size
        lane = eng.lane()
        lane.set_replica(capi.REPLICA_ON)                        # a LANE builds the group's replicas
        assert lane.stats().device_bytes_per_row == 84 and lane.stats().replica_active == 1
        assert eng.stats().device_bytes_per_row == 48            # the parent has not been told yet
        eng.set_replica(capi.REPLICA_ON)                         # ... and takes the group's over, nothing rebuilt
        assert eng.stats().device_bytes_per_row == 84 and eng.stats().replica_active == 1
        with pytest.raises(capi.Mi355Error):
            eng.rebuild_replica()                                # shared replicas are not rebuilt under a lane
        for h in (eng, lane):
            for row in (5, n - 1, 12_345):
                idx, sc = h.query_row_topn(row, topn)
                assert_topn_matches(idx, sc, oracle.scores(f, f[row]), row, topn)
            before = h.stats()
            out = torch.zeros(topn, dtype=torch.int64, device="cuda")
            h.enqueue_row_keys_streamed(4_321, topn, out)
            h.enqueue_flush()
            torch.cuda.synchronize()
            check_keys(f, 4_321, out.cpu().numpy(), topn)
            assert h.stats().route_q8 + h.stats().route_q8_lone > before.route_q8 + before.route_q8_lone   # forced ON: over the replica
        lane2 = eng.lane()                                       # made after the replicas: has them, and the parent's mode
        assert lane2.stats().replica_active == 1
        eng.set_replica(capi.REPLICA_OFF)
        eng.set_batch_path(capi.BATCH_MULTI)
        lane3 = eng.lane()                                       # the parent's forced modes are the lane's
        s0 = lane3.stats()
        assert s0.replica_active == 0
        qr = rng.integers(0, n, size=14)
        bi, bs, cnt = lane3.query_batch_topn(f[qr], qr, topn)
        s1 = lane3.stats()
        assert s1.route_multi_fp32 > s0.route_multi_fp32 and s1.route_multi_fp16 == s0.route_multi_fp16
        for b in (0, 13):
            assert_topn_matches(bi[b][:cnt[b]], bs[b][:cnt[b]], oracle.scores(f, f[qr[b]]), int(qr[b]), topn)
        for h in (lane, lane2, lane3):
            h.close()
    with NodeEngine(f, devices=[0, 0], placement=capi.PLACEMENT_REPLICATED) as node:
        node.set_replica(capi.REPLICA_ON)                        # used to fail on the second replica (a lane)
        assert node.shard_stats(0).replica_active == 1 and node.shard_stats(1).replica_active == 1
        for row in (5, n - 2, 999, 29_999):                      # synchronous calls take the replicas in turn
            idx, sc = node.query_row_topn(row, topn)
            want = oracle.scores(f, f[row])
            assert_topn_matches(idx, sc, want, row, topn, ref_idx=oracle.topn_heap(want, row, topn))
        tickets = [node.enqueue_row(r, topn) for r in (1, 2, 3, 4, 5, 6)]
        node.enqueue_flush()
        for t, r in zip(tickets, (1, 2, 3, 4, 5, 6)):
            idx, sc = node.wait(t, topn)
            assert_topn_matches(idx, sc, oracle.scores(f, f[r]), r, topn)
