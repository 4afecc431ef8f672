// sharded.hip — the row-sharded engine behind the C-ABI (include/mi355rec.h,
// "row-sharded catalogue" section): ONE host process drives every GPU of the node.
//
// Replaces the reference's single-device residency (cudaSetDevice(0),
// Recommender.cu:124): shard r owns the contiguous rows [lo_r, hi_r) on device
// devices[r] (an ordinary mi355rec handle created with row_base = lo_r, so its
// keys carry GLOBAL row ids).  Per query every shard runs the fused scan + local
// merge on its own stream, the per-shard top-N key lists meet on device
// devices[0], and the same merge kernel that merges per-workgroup lists merges
// the G per-shard lists.  The result does not depend on G.
//
// Exchange (the only inter-GPU step; 800 B per shard at top-100):
//   PEER  every shard's merge kernel STORES its keys straight into the gather
//         buffer on device 0 through a peer mapping (xGMI point-to-point), an
//         event per shard orders device 0's merge behind them.  No collective, no
//         copy launch.  Default when every device can map device 0's memory.
//   RCCL  one ncclAllGather of topn x batch uint64 keys per rank, issued by that rank's
//         WORKER THREAD on its own stream (communicators from ncclCommInitAll: one per
//         device, one thread per communicator — RCCL's one-thread-per-rank pattern, so no
//         ncclGroupStart/End is involved; the exchange BASELINE.json's north_star names).
//         librccl is opened lazily, only for this transport.
// This layer is host orchestration over the single-device C-ABI and the HIP
// runtime: it launches no kernel of its own.
//
// Two ways in:
//   * synchronous calls (mi355rec_sharded_query_*): what Recommender::recommendByIndex
//     maps to.  One shard -> straight to that shard's handle.  More: scan + local merge per
//     shard, exchange, merge on the first device straight into mapped host memory, ONE
//     host wait (the first device's stream; everything it depends on is ordered by events).
//   * the STREAM (mi355rec_sharded_enqueue_* / _flush / _wait): a serving loop.  One streamed
//     scan launch per shard per query, one exchange + one batched merge per WINDOW of
//     queries, tickets instead of host waits.
// Host threads.  With more than one shard every shard has a WORKER thread that owns its device context,
// its stream and its single-device handle: the caller's thread only posts small task records (a
// single-producer ring per worker) and the workers issue the HIP launches in parallel — one launch per
// shard per query costs the caller ~0.1 us instead of G x ~5 us, and a synchronous query on G GPUs pays
// one shard's launch chain, not G of them.  Cross-shard ordering stays on the device (one event per shard,
// or the collective); the only host-side hand-off is "shard r has RECORDED its event" before the first
// device's worker makes its stream wait for it.  With one shard there are no threads at all.
// A query by ROW never comes back to the host: every shard's kernels read its 48 bytes from
// the owning shard's memory through the peer mapping (checked once at create time against
// the by-value path; without all-pairs peer access the row is fetched once per query).
//
// PLACEMENT (mi355rec_create_placed).  Row sharding is what north_star specifies and the only way to a lower
// latency once a scan is longer than its launches; but a shard below ~4 M rows is launch-bound (measured on one
// MI355X, streamed queries over the 8-bit replica: 12.1 us per query at 1.25 M rows, 15.4 at 3 M, 24.4 at 10 M, 207
// at 100 M — 11 us of fixed cost + 1.95 us per million rows), and every query of a sharded catalogue pays the
// exchange on top.  So
//   SHARDED     rows split over G devices; G given, or (AUTO) as many as keep >= 4 M rows per shard: 1 for the 114 k
//               catalogue of BASELINE configs[0] and up to 7.9 M rows, 2 at 10 M, all 8 from 32 M rows on;
//   REPLICATED  every device holds ALL rows (840 MB at 10 M rows with both replicas) and serves whole WINDOWS of the
//               stream by itself, round-robin: no exchange, no merge across devices, queries/s scale with the
//               devices for any catalogue that fits one.  Synchronous calls go to the replicas in turn.
// A "shard" of a replicated handle is a full replica (lo = 0, hi = n).
#include "node_stream.hip.h"

extern "C" {

const char* mi355rec_sharded_last_error(const mi355rec_sharded_t* h) {
    return h ? h->err.c_str() : g_sharded_error.c_str();
}

void mi355rec_sharded_destroy(mi355rec_sharded_t* h) {
    if (!h) return;
    if (h->cpu) {
        mi355cpu::node_destroy(h->cpu);
        delete h;
        return;
    }
    DeviceRestore restore;
    stop_workers(h);   // they finish what has been posted first
    for (Shard& s : h->shards) {
        if (hipSetDevice(s.device) != hipSuccess) continue;
        if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    free_stream(h);
    for (size_t i = 0; i < h->r_merged.size(); ++i) {
        if (!h->r_merged[i]) continue;
        if (hipSetDevice(h->shards[i % h->shards.size()].device) == hipSuccess) (void)hipEventDestroy(h->r_merged[i]);
    }
    for (Shard& s : h->shards) {
        if (hipSetDevice(s.device) != hipSuccess) continue;
        if (s.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s.comm);
        if (s.engine) mi355rec_destroy(s.engine);
        if (s.local_keys) (void)hipFree(s.local_keys);
        if (s.gathered) (void)hipFree(s.gathered);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream && s.owns_stream) (void)hipStreamDestroy(s.stream);
    }
    if (!h->shards.empty() && hipSetDevice(h->shards[0].device) == hipSuccess) {
        if (h->gather0) (void)hipFree(h->gather0);
        if (h->d_keys) (void)hipFree(h->d_keys);
        if (h->d_idx) (void)hipFree(h->d_idx);
        if (h->d_score) (void)hipFree(h->d_score);
        if (h->h_idx) (void)hipHostFree(h->h_idx);
        if (h->h_score) (void)hipHostFree(h->h_score);
        for (Window& w : h->win)
            if (w.merged) (void)hipEventDestroy(w.merged);
    }
    delete h;
}

}  // extern "C"

namespace {

// Shard r of a SHARDED handle holds its balanced block of rows; of a REPLICATED one, all of them.
int create_on(const float* feats_host, int64_t n, int dim, const int* devices, int n_shards, bool replicated,
              mi355rec_sharded_t** out) {
    if (out) *out = nullptr;
    if (!out || !feats_host || !devices) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null argument");
    if (dim != MI355REC_DIM) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "dim must be %d, got %d", MI355REC_DIM, dim);
    if (n < 1 || n > 0xfffffffell) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "rows %lld out of range", (long long)n);
    if (n_shards < 1 || n_shards > MI355REC_MAX_SHARDS)
        return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "n_shards must be in [1, %d], got %d", MI355REC_MAX_SHARDS, n_shards);
    const int visible = mi355rec_device_count();
    if (visible <= 0)
        return sfail(nullptr, MI355REC_ERR_NO_DEVICE, "no HIP device visible: the MI355X engine has no CPU fallback");
    for (int r = 0; r < n_shards; ++r)
        if (devices[r] < 0 || devices[r] >= visible)
            return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "device %d not in [0,%d)", devices[r], visible);

    DeviceRestore restore;
    mi355rec_sharded* h = new mi355rec_sharded();
    h->n = n;
    h->replicated = replicated;
    h->shards.resize(n_shards);
    auto bail = [&](int code) {
        const std::string keep = g_sharded_error;
        mi355rec_sharded_destroy(h);
        g_sharded_error = keep;
        return code;
    };
    const int root = devices[0];
    for (int r = 0; r < n_shards; ++r) {
        Shard& s = h->shards[r];
        s.device = devices[r];
        if (replicated) {
            s.lo = 0;
            s.hi = n;
        } else {
            bounds(n, n_shards, r, s.lo, s.hi);
        }
        if (hipSetDevice(s.device) != hipSuccess) return bail(sfail(nullptr, MI355REC_ERR_HIP, "hipSetDevice(%d) failed", s.device));
        // a replica on a device that already holds one is a LANE of it (mi355rec_create_lane: the same rows and replicas,
        // own stream state): listing a device twice buys the overlap of two chains of launches, not a second copy
        int same = -1;
        for (int p = 0; p < r && replicated && same < 0; ++p)
            if (h->shards[p].device == s.device) same = p;
        const int rc = same >= 0 ? mi355rec_create_lane(h->shards[same].engine, &s.engine)
                                 : mi355rec_create(s.hi > s.lo ? feats_host + s.lo * MI355REC_DIM : nullptr, s.hi - s.lo, dim, s.device,
                                                   s.lo, &s.engine);
        if (rc != MI355REC_OK) return bail(sfail(nullptr, rc, "shard %d on device %d: %s", r, s.device, mi355rec_last_global_error()));
        // The shard works on the stream its engine was created with: for a lane that is the stream mi355rec_create_lane CHOSE
        // because kernels on it run beside the first replica's (a hardware queue of its own) — a stream made here would be
        // bound to whichever queue is next, possibly the same one.
        s.stream = static_cast<hipStream_t>(mi355rec_own_stream(s.engine));
        s.owns_stream = false;
        if (!s.stream || hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)
            return bail(sfail(nullptr, MI355REC_ERR_HIP, "stream / event creation on device %d failed", s.device));
    }
    // Peer mappings.  Stores into the first device's gather buffers need ITS memory mapped on every
    // other device (PEER transport); queries by row need every device's rows mapped on every other.
    // (Replicas never read or write each other's memory.)
    for (int a = 0; a < n_shards && !replicated; ++a) {
        for (int b = 0; b < n_shards; ++b) {
            const int da = devices[a], db = devices[b];
            if (da == db) continue;
            bool ok = false;
            int can = 0;
            if (hipSetDevice(da) == hipSuccess && hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();
            if (!ok) {
                h->peer_rows = false;
                if (db == root) h->peer_ok = false;
            }
        }
    }
    if (hipSetDevice(root) != hipSuccess) return bail(sfail(nullptr, MI355REC_ERR_HIP, "hipSetDevice(%d) failed", root));
    for (Window& w : h->win)
        if (hipEventCreateWithFlags(&w.merged, hipEventDisableTiming) != hipSuccess)
            return bail(sfail(nullptr, MI355REC_ERR_HIP, "event creation on device %d failed", root));
    h->transport = h->peer_ok ? MI355REC_TRANSPORT_PEER : MI355REC_TRANSPORT_RCCL;
    if (replicated) {   // one "results are in host memory" event per (ring entry, replica), on the replica's device
        h->r_merged.assign(static_cast<size_t>(kStreamDepth) * n_shards, nullptr);
        for (int w = 0; w < kStreamDepth; ++w)
            for (int r = 0; r < n_shards; ++r)
                if (hipSetDevice(devices[r]) != hipSuccess ||
                    hipEventCreateWithFlags(&h->r_merged[static_cast<size_t>(w) * n_shards + r], hipEventDisableTiming) != hipSuccess)
                    return bail(sfail(nullptr, MI355REC_ERR_HIP, "event creation on device %d failed", devices[r]));
    }

    if (n_shards > 1) {   // one worker per shard (none for a single shard: its calls are made by the caller)
        for (int r = 0; r < n_shards; ++r) {
            h->workers.emplace_back(new Worker());
            h->workers.back()->ring.resize(Worker::kCap);
        }
        for (int r = 0; r < n_shards; ++r) h->workers[r]->th = std::thread(worker_main, h, r);
    }

    // Queries by row through the peer mapping have to give the by-value path's keys: one query
    // per shard boundary, checked here, once, on real multi-device placements (a mismatch or an
    // error switches the pointer path off for this handle; mi355rec_sharded_info's note says so).
    bool distinct = false;
    for (int r = 1; r < n_shards; ++r) distinct = distinct || devices[r] != devices[0];
    if (replicated) {
        // nothing to check: a replica reads its own rows
    } else if (distinct && h->peer_rows && n >= 2) {
        const int topn = n - 1 < 16 ? static_cast<int>(n - 1) : 16;
        std::vector<int64_t> i_ptr(topn), i_val(topn);
        std::vector<float> s_ptr(topn), s_val(topn);
        for (int r = 0; r < n_shards && h->peer_rows; ++r) {
            const Shard& s = h->shards[r];
            if (s.hi == s.lo) continue;
            const int64_t row = s.hi - 1;
            int c0 = 0, c1 = 0;
            int rc = mi355rec_sharded_query_row_topn(h, row, topn, i_ptr.data(), s_ptr.data(), &c0);
            h->peer_rows = false;
            const int rc2 = mi355rec_sharded_query_row_topn(h, row, topn, i_val.data(), s_val.data(), &c1);
            h->peer_rows = true;
            if (rc2 != MI355REC_OK) return bail(sfail(nullptr, rc2, "self-check query failed: %s", h->err.c_str()));
            if (rc != MI355REC_OK || c0 != c1 || i_ptr != i_val ||
                std::memcmp(s_ptr.data(), s_val.data(), sizeof(float) * topn) != 0) {
                h->peer_rows = false;
                h->note = "queries by row travel by value: reading a row through the peer mapping did not reproduce the by-value result";
            }
        }
    } else if (!h->peer_rows) {
        h->note = "queries by row travel by value: no all-pairs peer access between the shards' devices";
    }
    *out = h;
    return MI355REC_OK;
}

}  // namespace

extern "C" {

int mi355rec_create_sharded_on(const float* feats_host, int64_t n, int dim, const int* devices, int n_shards,
                               mi355rec_sharded_t** out) {
    return create_on(feats_host, n, dim, devices, n_shards, false, out);
}

// How many devices a row-sharded catalogue of n rows is spread over when the caller does not say (see "PLACEMENT"
// at the top): as many as keep at least kRowsPerShardAuto rows per shard — a smaller shard is launch-bound, and
// every further shard adds to the exchange.
constexpr int64_t kRowsPerShardAuto = 4000000;

int mi355rec_auto_shards(int64_t n, int visible_devices) {
    if (visible_devices < 1) return 0;
    int64_t g = n / kRowsPerShardAuto;
    if (g < 1) g = 1;
    if (g > visible_devices) g = visible_devices;
    if (g > MI355REC_MAX_SHARDS) g = MI355REC_MAX_SHARDS;
    return static_cast<int>(g);
}

int mi355rec_create_placed(const float* feats_host, int64_t n, int dim, const int* devices, int n_devices, int placement,
                           mi355rec_sharded_t** out) {
    if (out) *out = nullptr;
    if (placement != MI355REC_PLACEMENT_AUTO && placement != MI355REC_PLACEMENT_SHARDED && placement != MI355REC_PLACEMENT_REPLICATED)
        return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "unknown placement %d", placement);
    const int visible = mi355rec_device_count();
    if (visible <= 0) {
        // No device at all.  An explicit device list cannot be honoured; otherwise the catalogue is served by the
        // product's own CPU backend, as the reference falls back to its CPU loop (Recommender.cu:117-127,176-181).
        if (devices || n_devices > 0)
            return sfail(nullptr, MI355REC_ERR_NO_DEVICE, "no HIP device visible, and %d device(s) were asked for", n_devices);
        if (!out || !feats_host) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null argument");
        if (dim != MI355REC_DIM) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "dim must be %d, got %d", MI355REC_DIM, dim);
        if (n < 1 || n > 0xfffffffell) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "rows %lld out of range", (long long)n);
        mi355rec_sharded* h = nullptr;
        try {   // (nothing may leave a C entry point as an exception: the handle and its note allocate)
            h = new mi355rec_sharded();
            h->n = n;
            h->peer_rows = false;
            h->cpu = mi355cpu::node_create(feats_host, n);
            if (!h->cpu) {
                delete h;
                return sfail(nullptr, MI355REC_ERR_OUT_OF_MEMORY, "CPU backend: cannot hold %lld rows", (long long)n);
            }
            h->note = "CPU backend: no HIP device visible (" + std::to_string(mi355cpu::threads(mi355cpu::node_catalogue(h->cpu))) +
                      " OpenMP thread(s))";
        } catch (const std::bad_alloc&) {
            if (h && h->cpu) mi355cpu::node_destroy(h->cpu);
            if (h) h->cpu = nullptr;
            delete h;
            return sfail(nullptr, MI355REC_ERR_OUT_OF_MEMORY, "out of host memory for the CPU backend's handle");
        }
        *out = h;
        return MI355REC_OK;
    }
    std::vector<int> devs;
    if (devices) {   // an explicit list (a device may repeat: virtual shards / replicas on a one-GPU box)
        if (n_devices < 1 || n_devices > MI355REC_MAX_SHARDS)
            return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "n_devices must be in [1, %d] with an explicit list, got %d", MI355REC_MAX_SHARDS, n_devices);
        devs.assign(devices, devices + n_devices);
    } else {
        if (n_devices < 0 || n_devices > visible)
            return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "n_devices %d but %d device(s) visible", n_devices, visible);
        int g = n_devices;
        if (g == 0)   // the library decides: replicas on every device; shards by the size of the catalogue
            g = placement == MI355REC_PLACEMENT_REPLICATED ? (visible < MI355REC_MAX_SHARDS ? visible : MI355REC_MAX_SHARDS)
                                                            : mi355rec_auto_shards(n, visible);
        devs.resize(g);
        for (int d = 0; d < g; ++d) devs[d] = d;
    }
    return create_on(feats_host, n, dim, devs.data(), static_cast<int>(devs.size()), placement == MI355REC_PLACEMENT_REPLICATED, out);
}

int mi355rec_create_sharded(const float* feats_host, int64_t n, int dim, int n_devices, mi355rec_sharded_t** out) {
    return mi355rec_create_placed(feats_host, n, dim, nullptr, n_devices, MI355REC_PLACEMENT_SHARDED, out);
}

int mi355rec_sharded_placement(const mi355rec_sharded_t* h) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return MI355REC_PLACEMENT_CPU;
    return h->replicated ? MI355REC_PLACEMENT_REPLICATED : MI355REC_PLACEMENT_SHARDED;
}

int mi355rec_sharded_set_transport(mi355rec_sharded_t* h, int transport) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->replicated || h->cpu) {   // nothing is ever exchanged (replicas; the CPU backend): either value is accepted and ignored
        if (transport != MI355REC_TRANSPORT_PEER && transport != MI355REC_TRANSPORT_RCCL)
            return sfail(h, MI355REC_ERR_INVALID_ARG, "unknown transport %d", transport);
        return MI355REC_OK;
    }
    if (transport == MI355REC_TRANSPORT_PEER) {
        if (!h->peer_ok) return sfail(h, MI355REC_ERR_INVALID_ARG, "peer access to device %d is not available from every shard", h->shards[0].device);
    } else if (transport != MI355REC_TRANSPORT_RCCL) {
        return sfail(h, MI355REC_ERR_INVALID_ARG, "unknown transport %d", transport);
    }
    if (transport != h->transport) {   // an open stream window is closed under the transport it was filled with
        DeviceRestore restore;
        int rc = stream_flush(h);
        if (rc) return rc;
        rc = drain_workers(h);
        if (rc) return rc;
    }
    h->transport = transport;
    return MI355REC_OK;
}

int mi355rec_sharded_info(const mi355rec_sharded_t* h, int* n_shards, int* transport, int64_t* rows,
                          int* devices_out, int64_t* shard_rows_out) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (n_shards) *n_shards = static_cast<int>(h->shards.size());
    if (transport) *transport = h->transport;
    if (rows) *rows = h->n;
    for (size_t r = 0; r < h->shards.size(); ++r) {
        if (devices_out) devices_out[r] = h->shards[r].device;
        if (shard_rows_out) shard_rows_out[r] = h->shards[r].hi - h->shards[r].lo;
    }
    return MI355REC_OK;
}

const char* mi355rec_sharded_note(const mi355rec_sharded_t* h) { return h ? h->note.c_str() : ""; }

// What RCCL itself says about the communicators of the RCCL transport (include/mi355rec_diag.h): *comms = communicators this
// handle holds (one per shard; 0 until the transport has been used), *ranks = ncclCommCount of the first one, *ranks_agree = 1
// when every communicator reports that same count and its own shard index as its rank.
int mi355rec_sharded_rccl_ranks(const mi355rec_sharded_t* h, int* comms, int* ranks, int* ranks_agree) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    int n_comms = 0, first = 0, agree = 1;
    if (!h->cpu && h->rccl_ready && g_rccl.CommCount) {
        for (size_t r = 0; r < h->shards.size(); ++r) {
            if (!h->shards[r].comm) continue;
            int count = 0, rank = -1;
            if (g_rccl.CommCount(h->shards[r].comm, &count) != 0) count = -1;
            if (g_rccl.CommUserRank && g_rccl.CommUserRank(h->shards[r].comm, &rank) != 0) rank = -1;
            if (n_comms == 0) first = count;
            if (count != first || (g_rccl.CommUserRank && rank != static_cast<int>(r))) agree = 0;
            ++n_comms;
        }
    }
    if (comms) *comms = n_comms;
    if (ranks) *ranks = first;
    if (ranks_agree) *ranks_agree = n_comms > 0 ? agree : 0;
    return MI355REC_OK;
}

int mi355rec_sharded_set_timing(mi355rec_sharded_t* h, int enabled) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return MI355REC_OK;   // (no kernels to time)
    const int drc = drain_workers(h);
    if (drc) return drc;
    for (Shard& s : h->shards) S_ENG(h, s, mi355rec_set_timing(s.engine, enabled));
    return MI355REC_OK;
}

int mi355rec_sharded_shard_stats(const mi355rec_sharded_t* hc, int shard, mi355rec_stats_t* out) {
    mi355rec_sharded_t* h = const_cast<mi355rec_sharded_t*>(hc);
    if (!h || !out) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (shard < 0 || shard >= static_cast<int>(h->shards.size()))
        return sfail(h, MI355REC_ERR_INVALID_ARG, h->cpu ? "the CPU backend has no device shard %d" : "no shard %d", shard);
    const int drc = drain_workers(h);
    if (drc) return drc;
    const Shard& s = h->shards[shard];
    S_ENG(h, s, mi355rec_stats(s.engine, out));
    return MI355REC_OK;
}

int mi355rec_sharded_set_replica(mi355rec_sharded_t* h, int mode) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return MI355REC_OK;   // (the CPU backend scans the fp32 rows, always)
    DeviceRestore restore;
    int rc = stream_flush(h);
    if (rc) return rc;
    rc = drain_workers(h);
    if (rc) return rc;
    for (Shard& s : h->shards) {
        if (s.hi == s.lo) continue;
        S_ENG(h, s, mi355rec_set_replica(s.engine, mode));
    }
    return MI355REC_OK;
}

int mi355rec_sharded_rows_by_pointer(const mi355rec_sharded_t* h) { return h && h->peer_rows ? 1 : 0; }

int mi355rec_sharded_query_batch_topn(mi355rec_sharded_t* h, const float* queries, int batch,
                                      const int64_t* exclude_global, int topn, int64_t* out_idx, float* out_score,
                                      int* out_count) {
    if (!h || !queries || !out_idx) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return sfail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    if (topn <= 0) return sfail(h, MI355REC_ERR_INVALID_ARG, "topn must be positive, got %d", topn);
    if (h->cpu) {
        for (int b = 0; b < batch; ++b) {
            const char* why = nullptr;
            const int rc = mi355cpu::node_query(h->cpu, queries + static_cast<size_t>(b) * MI355REC_DIM, exclude_global ? exclude_global[b] : -1,
                                                topn, out_idx + static_cast<size_t>(b) * topn,
                                                out_score ? out_score + static_cast<size_t>(b) * topn : nullptr, out_count ? out_count + b : nullptr, &why);
            if (rc) return cpu_result(h, rc, why);
        }
        return MI355REC_OK;
    }
    if (h->shards.size() == 1) {   // one shard: its own handle is the whole engine (row_base 0)
        const int rc = mi355rec_query_batch_topn(h->shards[0].engine, queries, batch, exclude_global, topn, out_idx, out_score, out_count);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "%s", mi355rec_last_error(h->shards[0].engine));
    }
    if (h->replicated) {   // any replica is the whole engine
        DeviceRestore restore;
        Shard* s = nullptr;
        int rc = take_replica(h, &s);
        if (rc) return rc;
        rc = mi355rec_query_batch_topn(s->engine, queries, batch, exclude_global, topn, out_idx, out_score, out_count);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "replica on device %d: %s", s->device, mi355rec_last_error(s->engine));
    }
    // lists are at most n long
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    DeviceRestore restore;
    if (eff > MI355REC_MAX_TOPN_FAST) return run_queries_large(h, queries, exclude_global, batch, eff, topn, out_idx, out_score, out_count);
    const int rc = run_queries(h, queries, nullptr, exclude_global, batch, eff);
    if (rc) return rc;
    copy_rows(h->h_idx, h->h_score, batch, eff, topn, out_idx, out_score, out_count);
    return MI355REC_OK;
}

int mi355rec_sharded_query_topn(mi355rec_sharded_t* h, const float* query12, int64_t exclude_global, int topn,
                                int64_t* out_idx, float* out_score, int* out_count) {
    return mi355rec_sharded_query_batch_topn(h, query12, 1, &exclude_global, topn, out_idx, out_score, out_count);
}

int mi355rec_sharded_query_row_topn(mi355rec_sharded_t* h, int64_t global_row, int topn, int64_t* out_idx,
                                    float* out_score, int* out_count) {
    if (!h || !out_idx) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (global_row < 0 || global_row >= h->n)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)global_row);
    if (topn <= 0) return sfail(h, MI355REC_ERR_INVALID_ARG, "topn must be positive, got %d", topn);
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_query(h->cpu, mi355cpu::row(mi355cpu::node_catalogue(h->cpu), global_row), global_row, topn,
                                                  out_idx, out_score, out_count, &why), why);
    }
    if (h->shards.size() == 1) {   // what Recommender::recommendByIndex costs on a one-GPU box: exactly mi355rec_query_row_topn
        const int rc = mi355rec_query_row_topn(h->shards[0].engine, global_row, topn, out_idx, out_score, out_count);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "%s", mi355rec_last_error(h->shards[0].engine));
    }
    if (h->replicated) {
        DeviceRestore restore;
        Shard* s = nullptr;
        int rc = take_replica(h, &s);
        if (rc) return rc;
        rc = mi355rec_query_row_topn(s->engine, global_row, topn, out_idx, out_score, out_count);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "replica on device %d: %s", s->device, mi355rec_last_error(s->engine));
    }
    DeviceRestore restore;
    float q[MI355REC_DIM];
    const float* qptr = nullptr;
    int rc = locate_row(h, global_row, &qptr, q);
    if (rc) return rc;
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    if (eff > MI355REC_MAX_TOPN_FAST) {
        if (qptr) {   // cold path: by value
            const Shard* own = owner_of(h, global_row);
            rc = mi355rec_fetch_row(own->engine, global_row - own->lo, q);
            if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", own->device, mi355rec_last_error(own->engine));
        }
        return run_queries_large(h, q, &global_row, 1, eff, topn, out_idx, out_score, out_count);
    }
    rc = run_queries(h, q, qptr, &global_row, 1, eff);
    if (rc) return rc;
    copy_rows(h->h_idx, h->h_score, 1, eff, topn, out_idx, out_score, out_count);
    return MI355REC_OK;
}

int mi355rec_sharded_scores_row(mi355rec_sharded_t* h, int64_t global_row, float* out_host) {
    if (!h || !out_host) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (global_row < 0 || global_row >= h->n)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)global_row);
    if (h->cpu) {
        const mi355cpu::Catalogue* c = mi355cpu::node_catalogue(h->cpu);
        mi355cpu::scores(c, mi355cpu::row(c, global_row), out_host);
        return MI355REC_OK;
    }
    DeviceRestore restore;
    if (h->replicated) {
        Shard* s = nullptr;
        int rc = take_replica(h, &s);
        if (rc) return rc;
        rc = mi355rec_scores_row(s->engine, global_row, out_host);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "replica on device %d: %s", s->device, mi355rec_last_error(s->engine));
    }
    float q[MI355REC_DIM];
    const Shard* own = owner_of(h, global_row);
    int rc = drain_workers(h);   // cold path: the caller's thread drives every shard itself
    if (rc) return rc;
    rc = mi355rec_fetch_row(own->engine, global_row - own->lo, q);
    if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", own->device, mi355rec_last_error(own->engine));
    for (const Shard& s : h->shards) {
        if (s.hi == s.lo) continue;
        rc = mi355rec_scores(s.engine, q, out_host + s.lo);
        if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", s.device, mi355rec_last_error(s.engine));
    }
    return MI355REC_OK;
}

// ---- the stream ---------------------------------------------------------------------------

int mi355rec_sharded_set_window(mi355rec_sharded_t* h, int window) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (window < 1 || window > kMaxWindow) return sfail(h, MI355REC_ERR_INVALID_ARG, "window must be in [1, %d], got %d", kMaxWindow, window);
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_set_window(h->cpu, window, &why), why);
    }
    if (window == h->s_window) return MI355REC_OK;
    DeviceRestore restore;
    int rc = stream_flush(h);   // the open window is closed in the old geometry
    if (rc) return rc;
    rc = drain_workers(h);
    if (rc) return rc;
    // buffers are re-made by the next enqueue (ensure_stream sees s_alloc_window != s_window)
    h->next_ticket = (h->next_ticket + window - 1) / window * window;
    h->issued_upto = h->next_ticket;
    for (Shard& s : h->shards) {
        S_HIP(h, hipSetDevice(s.device));
        S_HIP(h, hipStreamSynchronize(s.stream));
    }
    free_stream(h);
    for (Window& w : h->win) {
        w.abs = -1;
        w.count = 0;
        w.handed = false;
        w.issued = false;
    }
    h->s_window = window;
    return MI355REC_OK;
}

int mi355rec_sharded_set_window_mode(mi355rec_sharded_t* h, int batched) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return MI355REC_OK;   // (a query is computed when it is enqueued, either way)
    if ((batched != 0) == h->batched_windows) return MI355REC_OK;
    DeviceRestore restore;
    int rc = stream_flush(h);
    if (rc) return rc;
    rc = drain_workers(h);
    if (rc) return rc;
    for (Shard& s : h->shards) {
        S_HIP(h, hipSetDevice(s.device));
        S_HIP(h, hipStreamSynchronize(s.stream));
    }
    free_stream(h);   // the next enqueue re-makes the buffers and decides the mode again
    for (Window& w : h->win) {
        w.abs = -1;
        w.count = 0;
        w.handed = false;
        w.issued = false;
    }
    h->batched_windows = batched != 0;
    return MI355REC_OK;
}

int mi355rec_sharded_enqueue_query(mi355rec_sharded_t* h, const float* query12, int64_t exclude_global, int topn,
                                   int64_t* ticket) {
    if (!h || !query12) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_enqueue(h->cpu, query12, exclude_global, topn, ticket, &why), why);
    }
    DeviceRestore restore;
    return stream_enqueue(h, nullptr, query12, exclude_global, topn, ticket);
}

int mi355rec_sharded_enqueue_row(mi355rec_sharded_t* h, int64_t global_row, int topn, int64_t* ticket) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (global_row < 0 || global_row >= h->n)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)global_row);
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_enqueue(h->cpu, mi355cpu::row(mi355cpu::node_catalogue(h->cpu), global_row), global_row, topn,
                                                    ticket, &why), why);
    }
    DeviceRestore restore;
    float q[MI355REC_DIM] = {0};
    const float* qptr = nullptr;
    if (!h->replicated) {
        const int rc = locate_row(h, global_row, &qptr, q);
        if (rc) return rc;
    }
    return stream_enqueue(h, qptr, q, global_row, topn, ticket, global_row);
}

int mi355rec_sharded_enqueue_flush(mi355rec_sharded_t* h) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return mi355cpu::node_flush(h->cpu);
    DeviceRestore restore;
    const int64_t t0 = now_ns();
    const int rc = stream_flush(h);
    h->st_host_ns += now_ns() - t0;
    return rc;
}

int mi355rec_sharded_wait(mi355rec_sharded_t* h, int64_t ticket, int64_t* out_idx, float* out_score, int* out_count) {
    if (!h || !out_idx) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_wait(h->cpu, ticket, out_idx, out_score, out_count, &why), why);
    }
    if (!h->s_topn || ticket < 0 || ticket >= h->next_ticket)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "ticket %lld was never handed out", (long long)ticket);
    DeviceRestore restore;
    const int W = h->s_window;
    const int64_t abs = ticket / W;
    const int w = static_cast<int>(abs % kStreamDepth);
    const int slot = static_cast<int>(ticket % W);
    Window& win = h->win[w];
    if (win.abs != abs || slot >= win.count)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "the results of ticket %lld are no longer kept (ring of %d windows of %d)",
                     (long long)ticket, kStreamDepth, W);
    if (!win.issued) {   // its window is still open (or waits for the lag): close it now
        const int rc = stream_flush(h);
        if (rc) return rc;
    }
    {
        const int rc = wait_window(h, w);   // the merge has been enqueued, `merged` recorded, and the results are in host memory
        if (rc) return rc;
    }
    const int topn = h->s_topn;
    const size_t off = (static_cast<size_t>(w) * W + slot) * topn;
    copy_rows(h->s_hidx + off, h->s_hscore + off, 1, topn, topn, out_idx, out_score, out_count);
    return MI355REC_OK;
}

int mi355rec_sharded_stream_stats(const mi355rec_sharded_t* h, int64_t* queries, int64_t* exchanges, int64_t* host_ns) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) {
        mi355cpu::node_stream_stats(h->cpu, queries, exchanges, host_ns);   // ("exchanges": windows opened)
        return MI355REC_OK;
    }
    if (queries) *queries = h->st_queries;
    if (exchanges) *exchanges = h->st_exchanges;
    if (host_ns) *host_ns = h->st_host_ns;
    return MI355REC_OK;
}

}  // extern "C"
