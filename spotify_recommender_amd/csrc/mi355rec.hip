// mi355rec.hip — the C-ABI (include/mi355rec.h) over the gfx950 kernels: every entry point validates its arguments, picks
// the device and the stream order, and hands over to the engine (engine_state / engine_single / engine_batch .hip.h, all part
// of this one translation unit).  Replaces the reference's Recommender internals (Recommender.cu:100-318) behind
// include/Recommender.h.  No CPU fallback anywhere in this file.
#include "engine_batch.hip.h"

extern "C" {

int mi355rec_build_flags(void) {
    int flags = 0;
#ifdef MI355REC_EXPERIMENTS
    flags |= MI355REC_BUILD_EXPERIMENTS;
#endif
#ifdef MI355REC_PHASE_CLOCK
    flags |= MI355REC_BUILD_PHASE_CLOCK;
#endif
#ifdef MI355REC_TEST_HOOKS
    flags |= MI355REC_BUILD_TEST_HOOKS;
#endif
    return flags;
}

int mi355rec_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

const char* mi355rec_last_global_error(void) { return g_last_error.c_str(); }

namespace {
// The buffers of the STREAMED entry points (two alternating sets of per-workgroup lists for single queries; lists, sample
// values, cutoffs and arrival counters of the streams of batches) are allocated with the handle, not by the first streamed
// call: nothing in the serving path allocates or synchronises after create (the six hipMallocs + a stream
// synchronise of ensure_mstream used to land inside whatever call came first).  A failure here is not the create's: the
// lazy paths (ensure_streamed / ensure_mstream) try again on first use and report it there.
void preallocate_stream_state(mi355rec* h) {
    if (!h || h->n < 1) return;
    DeviceGuard guard(h->device);
    const std::string kept = h->err;
    bool ok = ensure_streamed(h) == MI355REC_OK;
    if (ok && h->d_half) ok = ensure_mstream(h) == MI355REC_OK;
    if (!ok) {
        (void)hipGetLastError();
        h->err = kept;
    }
}

int create_and_preallocate(const float* feats, bool on_device, int64_t n, int dim, int device, int64_t row_base, int flags,
                           mi355rec_t** out) {
    const int rc = create_common(feats, on_device, n, dim, device, row_base, flags, out);
    if (rc == MI355REC_OK) preallocate_stream_state(*out);
    return rc;
}
}  // namespace

int mi355rec_create(const float* feats_host, int64_t n, int dim, int device, int64_t row_base,
                    mi355rec_t** out) {
    return create_and_preallocate(feats_host, false, n, dim, device, row_base, 0, out);
}

int mi355rec_create_device(const float* feats_dev, int64_t n, int dim, int device,
                           int64_t row_base, mi355rec_t** out) {
    return create_and_preallocate(feats_dev, true, n, dim, device, row_base, 0, out);
}

int mi355rec_create_ex(const float* feats_host, int64_t n, int dim, int device, int64_t row_base, int flags,
                       mi355rec_t** out) {
    return create_and_preallocate(feats_host, false, n, dim, device, row_base, flags, out);
}

int mi355rec_create_device_ex(const float* feats_dev, int64_t n, int dim, int device, int64_t row_base, int flags,
                              mi355rec_t** out) {
    return create_and_preallocate(feats_dev, true, n, dim, device, row_base, flags, out);
}

void mi355rec_destroy(mi355rec_t* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t e : h->ev_scan) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_merge) (void)hipEventDestroy(e);
    free_bq(h);
    for (int i = 0; i < 2; ++i) {
        if (h->d_mstream_lists[i]) (void)hipFree(h->d_mstream_lists[i]);
        if (h->d_mstream_seed[i]) (void)hipFree(h->d_mstream_seed[i]);
    }
    if (h->d_mstream_cuts) (void)hipFree(h->d_mstream_cuts);
    if (h->d_mstream_ctl) (void)hipFree(h->d_mstream_ctl);
    for (hipEvent_t e : h->ev_pass) (void)hipEventDestroy(e);
    if (h->owned_feats && !h->shared) (void)hipFree(h->owned_feats);
    if (h->d_block_lists) (void)hipFree(h->d_block_lists);
    if (h->d_lone_ctr) (void)hipFree(h->d_lone_ctr);
    {
        void* bufs[] = {h->d_half_seed, h->d_stream_seed[0], h->d_stream_seed[1], h->d_stream_ctl, h->d_lone_ctl};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
    }
    if (h->d_stream_lists[0]) (void)hipFree(h->d_stream_lists[0]);
    if (h->d_stream_lists[1]) (void)hipFree(h->d_stream_lists[1]);
    if (h->d_anchor) (void)hipFree(h->d_anchor);
    if (h->d_seed_keys) (void)hipFree(h->d_seed_keys);
    if (h->d_seed_vals) (void)hipFree(h->d_seed_vals);
    free_replica(h);
    if (h->d_keys) (void)hipFree(h->d_keys);
    if (h->d_idx) (void)hipFree(h->d_idx);
    if (h->d_score) (void)hipFree(h->d_score);
    if (h->d_scores_full) (void)hipFree(h->d_scores_full);
    if (h->h_idx) (void)hipHostFree(h->h_idx);
    if (h->h_score) (void)hipHostFree(h->h_score);
    if (h->h_done) (void)hipHostFree(h->h_done);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    if (h->shared && h->shared->refs.fetch_sub(1) == 1) {   // the last of a group of lanes: the rows and the replicas go with it
        void* bufs[] = {h->shared->owned_feats, h->shared->d_half, h->shared->d_q8};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
        delete h->shared;
    }
    delete h;
}

void* mi355rec_own_stream(mi355rec_t* h) { return h ? static_cast<void*>(h->stream) : nullptr; }

namespace {
// Do kernels on a's own stream and on b's run side by side?  A process has few hardware queues (four by default) and a stream
// is bound to one of them when it is created; streams that share a queue run their kernels strictly one after the other,
// and then two lanes are worth exactly one handle.  Measured, not guessed: a one-workgroup read of 480 KB of the rows
// (~30 us) on a's stream alone, then one on each stream at once; side by side the pair takes what one takes.
// Returns 1 / 0, or -1 when it cannot tell (a shard of under 10 000 rows, or a HIP error).
int streams_overlap(mi355rec* a, mi355rec* b) {
    constexpr int64_t kVec = 30000;   // float4s: 10 000 rows
    if (a->n * 3 < kVec || !a->d_seed_keys || !b->d_seed_keys) return -1;
    hipEvent_t e[3] = {nullptr, nullptr, nullptr};
    for (hipEvent_t& x : e)
        if (hipEventCreate(&x) != hipSuccess) {
            for (hipEvent_t y : e)
                if (y) (void)hipEventDestroy(y);
            return -1;
        }
    const float4* rows = reinterpret_cast<const float4*>(a->d_feats);
    auto once = [&](bool both) -> float {
        (void)hipEventRecord(e[0], a->stream);
        hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(kProbeBlock), 0, a->stream, rows, kVec, reinterpret_cast<uint32_t*>(a->d_seed_keys));
        if (both)
            hipLaunchKernelGGL(stream_probe_kernel, dim3(1), dim3(kProbeBlock), 0, b->stream, rows, kVec, reinterpret_cast<uint32_t*>(b->d_seed_keys));
        (void)hipEventRecord(e[1], a->stream);
        if (both) (void)hipEventRecord(e[2], b->stream);
        if (hipStreamSynchronize(a->stream) != hipSuccess || hipStreamSynchronize(b->stream) != hipSuccess) return -1.0f;
        float t1 = 0.f, t2 = 0.f;
        if (hipEventElapsedTime(&t1, e[0], e[1]) != hipSuccess) return -1.0f;
        if (both && hipEventElapsedTime(&t2, e[0], e[2]) != hipSuccess) return -1.0f;
        return t1 > t2 ? t1 : t2;
    };
    (void)once(true);   // (warm: code objects, the rows in the cache)
    float alone = once(false), pair = once(true);
    for (int k = 0; k < 4; ++k) {   // the better of five (with other processes on the GPU three were noise-driven)
        const float a2 = once(false), p2 = once(true);
        if (a2 > 0.f && (alone <= 0.f || a2 < alone)) alone = a2;
        if (p2 > 0.f && (pair <= 0.f || p2 < pair)) pair = p2;
    }
    for (hipEvent_t x : e) (void)hipEventDestroy(x);
    (void)hipGetLastError();
    if (alone <= 0.f || pair <= 0.f) return -1;
    return pair < 1.5f * alone ? 1 : 0;
}
}  // namespace

int mi355rec_lane_status(const mi355rec_t* h, int* stream_attempts, int* overlaps_parent) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (stream_attempts) *stream_attempts = h->lane_stream_attempts;
    if (overlaps_parent) *overlaps_parent = h->lane_overlaps;
    return MI355REC_OK;
}

int mi355rec_create_lane(mi355rec_t* parent, mi355rec_t** out) {
    if (out) *out = nullptr;
    if (!parent || !out) return fail(parent, MI355REC_ERR_INVALID_ARG, "null argument");
    if (parent->n < 1) return fail(parent, MI355REC_ERR_INVALID_ARG, "an empty shard has no lanes");
    DeviceGuard guard(parent->device);
    mi355rec_t* lane = nullptr;
    // the parent's rows, borrowed; no replica of its own (it takes the parent's below)
    const int rc = create_common(parent->d_feats, true, parent->n, kDim, parent->device, parent->row_base, MI355REC_CREATE_NO_REPLICA, &lane);
    if (rc != MI355REC_OK) {
        parent->err = g_last_error;
        return rc;
    }
    lane->is_lane = true;
    lane->replica_allowed = parent->replica_allowed;
    // a lane takes the routes its parent takes: the modes the parent was forced into (mi355rec_set_replica / _set_batch_path)
    // are the lane's too, until it is told otherwise itself
    lane->replica_mode = parent->replica_mode;
    lane->batch_path = parent->batch_path;
    if (parent->d_half) {
        const int src = alloc_replica_state(lane);
        if (src != MI355REC_OK) {
            parent->err = lane->err;
            mi355rec_destroy(lane);
            return src;
        }
        if (hipStreamSynchronize(lane->stream) != hipSuccess) {
            mi355rec_destroy(lane);
            return fail(parent, MI355REC_ERR_HIP, "hipStreamSynchronize(lane)");
        }
    }
    if (!parent->shared) {   // from now on the group owns what the parent owned
        mi355rec::SharedRows* group = new (std::nothrow) mi355rec::SharedRows();   // (no exception may cross the C-ABI)
        if (!group) {
            mi355rec_destroy(lane);
            return fail(parent, MI355REC_ERR_OUT_OF_MEMORY, "out of host memory for the lane group");
        }
        parent->shared = group;
        group->owned_feats = parent->owned_feats;
        group->d_half = parent->d_half;
        group->d_q8 = parent->d_q8;
        group->margin_mix = parent->margin_mix;
        group->margin_mfma = parent->margin_mfma;
    }
    parent->shared->refs.fetch_add(1);
    lane->shared = parent->shared;
    lane->d_half = parent->d_half;
    lane->d_q8 = parent->d_q8;
    lane->margin_mix = parent->margin_mix;
    lane->margin_mfma = parent->margin_mfma;
    lane->replica_build_ms = 0.f;
    // A stream for the lane that really runs beside the parent's: test the one create_common made, and while the two share a
    // hardware queue replace it (every new stream is bound to the next queue), at most once round all of them.  The stream the
    // lane ends up with is always one that was TESTED: after the last attempt nothing is replaced any more (`lane_overlaps` 0
    // then says no stream was found that runs beside the parent's).  The verdict is the better of three timings of a ~30 us
    // kernel: with other processes on the same GPU it can come out differently from run to run — a lane is correct on any stream.
    constexpr int kLaneStreamAttempts = 6;
    for (int attempt = 1; attempt <= kLaneStreamAttempts; ++attempt) {
        lane->lane_stream_attempts = attempt;
        lane->lane_overlaps = streams_overlap(parent, lane);
        if (lane->lane_overlaps != 0 || attempt == kLaneStreamAttempts) break;   // side by side, cannot tell, or out of attempts
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
        (void)hipStreamDestroy(lane->stream);
        lane->stream = fresh;
    }
    preallocate_stream_state(lane);
    *out = lane;
    return MI355REC_OK;
}

const char* mi355rec_last_error(const mi355rec_t* h) {
    return h ? h->err.c_str() : g_last_error.c_str();
}

int mi355rec_set_timing(mi355rec_t* h, int enabled) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    h->timing = enabled != 0;
    h->timing_stride = enabled > 1 ? enabled : 1;
    h->n_scan_pairs = 0;
    h->n_merge_pairs = 0;
    h->n_pass_pairs = 0;
    h->scan_launches = 0;
    h->merge_launches = 0;
    h->pass_launches = 0;
    return MI355REC_OK;
}

int mi355rec_stats(const mi355rec_t* hc, mi355rec_stats_t* out) {
    mi355rec_t* h = const_cast<mi355rec_t*>(hc);
    if (!h || !out) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(h->device);
    // average over the pairs recorded since set_timing(1)
    auto avg = [&](std::vector<hipEvent_t>& evs, int pairs, float& last) {
        if (pairs <= 0) return;
        double sum = 0.0;
        int good = 0;
        for (int i = 0; i < pairs; ++i) {
            float ms = 0.f;
            if (hipEventSynchronize(evs[2 * i + 1]) == hipSuccess &&
                hipEventElapsedTime(&ms, evs[2 * i], evs[2 * i + 1]) == hipSuccess) {
                sum += ms;
                ++good;
            }
        }
        if (good) last = static_cast<float>(sum / good);
    };
    avg(h->ev_scan, h->n_scan_pairs, h->last_scan_ms);
    avg(h->ev_merge, h->n_merge_pairs, h->last_merge_ms);
    avg(h->ev_pass, h->n_pass_pairs, h->last_pass_ms);
    out->rows = h->n;
    out->row_base = h->row_base;
    out->device = h->device;
    out->compute_units = h->cus;
    out->grid_blocks = h->grid;
    out->block_threads = kScanBlock;
    out->bytes_per_query = h->n * kDim * static_cast<int64_t>(sizeof(float));
    out->last_scan_ms = h->last_scan_ms;
    out->last_merge_ms = h->last_merge_ms;
    out->last_pass_ms = h->last_pass_ms;
    out->batched_grid_blocks = h->bq.ready ? h->bq.grid : 0;
    out->batched_margin = h->bq.ready ? h->bq.margin : 0.0f;
    out->replica_bytes_per_query = h->d_half ? ((h->n + 1) / 2) * 48 : 0;
    out->replica_active = use_half(h, nullptr) ? 1 : 0;
    out->replica_grid_blocks = h->d_half ? (use_q8(h) ? h->qg.grid : h->hg.grid) : 0;
    out->replica_build_ms = h->replica_build_ms;
    out->replica_margin_single = h->d_half ? h->margin_mix : 0.0f;
    out->replica_margin_multi = h->d_half ? h->margin_mfma : 0.0f;
    out->replica_single_row_bytes = !h->d_half ? 0 : (use_q8(h) ? 12 : 24);
    out->replica_single_bytes_per_query = !h->d_half ? 0 : (use_q8(h) ? ((h->n + 3) / 4) * 48 : ((h->n + 1) / 2) * 48);
    out->lone_fused_queries = static_cast<int32_t>(h->lone_fused & 0x7fffffff);
    out->route_fp32 = h->routes.fp32;
    out->route_fp16 = h->routes.fp16;
    out->route_q8 = h->routes.q8;
    out->route_q8_lone = h->routes.q8_lone;
    out->route_multi_fp32 = h->routes.multi_fp32;
    out->route_multi_fp16 = h->routes.multi_fp16;
    out->route_multi_q8 = h->routes.multi_q8;
    out->route_mfma_two_pass = h->routes.mfma_two_pass;
    out->route_exact_queue = 0;
    if (h->bq.ready) {
        // queries the batched path handed to the exact scan: counted on the device (counters[5], never reset).  Read on the
        // handle's own stream (behind what the synchronous API enqueued there) — not a device-wide synchronisation: a
        // serving loop that polls the statistics does not stall the other streams.  Work still in flight on a caller's
        // stream is not in the figure yet.
        int queued = 0;
        if (hipMemcpyAsync(&queued, h->bq.counters + 5, sizeof queued, hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
            hipStreamSynchronize(h->stream) == hipSuccess)
            out->route_exact_queue = queued;
    }
    out->device_bytes_per_row = 48 + (h->d_half ? 24 : 0) + (h->d_q8 ? 12 : 0);
    return MI355REC_OK;
}

int mi355rec_stats_sized(const mi355rec_t* h, void* out, size_t out_size, size_t* written) {
    if (!out || out_size == 0) return fail(const_cast<mi355rec_t*>(h), MI355REC_ERR_INVALID_ARG, "null argument");
    mi355rec_stats_t full;
    std::memset(&full, 0, sizeof full);
    const int rc = mi355rec_stats(h, &full);
    if (rc) return rc;
    const size_t n = out_size < sizeof full ? out_size : sizeof full;
    std::memcpy(out, &full, n);
    if (written) *written = n;
    return MI355REC_OK;
}

// ---- asynchronous device API -------------------------------------------------

int mi355rec_enqueue_row_keys(mi355rec_t* h, int64_t local_row, int topn,
                              mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    rc = order_stream(h, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    return enqueue_query(h, h->d_feats + local_row * kDim, nullptr, h->row_base + local_row, topn, out_keys_dev, nullptr,
                         nullptr, static_cast<hipStream_t>(stream));
}

int mi355rec_enqueue_query_keys(mi355rec_t* h, const float* query12, int64_t exclude_global,
                                int topn, mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev || !query12) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    rc = order_stream(h, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    return enqueue_query(h, nullptr, query12, exclude_global, topn, out_keys_dev, nullptr, nullptr,
                         static_cast<hipStream_t>(stream));
}

int mi355rec_enqueue_row_keys_streamed(mi355rec_t* h, int64_t local_row, int topn, mi355rec_key_t* out_keys_dev,
                                       void* stream) {
    if (!h || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    return enqueue_streamed(h, h->d_feats + local_row * kDim, nullptr, h->row_base + local_row, topn, out_keys_dev, s);
}

int mi355rec_enqueue_query_keys_streamed(mi355rec_t* h, const float* query12, int64_t exclude_global, int topn,
                                         mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev || !query12) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    if (h->n < 1) {   // an EMPTY shard answers with an all-empty list at once (it still takes part in the merge)
        HIP_TRY(h, hipMemsetAsync(out_keys_dev, 0, sizeof(uint64_t) * static_cast<size_t>(topn), s));
        return MI355REC_OK;
    }
    return enqueue_streamed(h, nullptr, query12, exclude_global, topn, out_keys_dev, s);
}

int mi355rec_enqueue_batch_keys_streamed(mi355rec_t* h, const float* queries, const int64_t* exclude_global, int batch,
                                         int topn, mi355rec_key_t* out_keys_dev, void* stream) {
    if (!queries) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    return mi355rec_enqueue_batch_mixed_keys_streamed(h, queries, nullptr, exclude_global, batch, topn, out_keys_dev, stream);
}

int mi355rec_enqueue_batch_mixed_keys_streamed(mi355rec_t* h, const float* queries, const float* const* query_ptrs_dev,
                                               const int64_t* exclude_global, int batch, int topn,
                                               mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev || (!queries && !query_ptrs_dev)) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    const bool streamable = h->n >= kBqMinRows && half_multi_ok(h, topn) && h->batch_path != MI355REC_BATCH_MULTI &&
                            h->batch_path != MI355REC_BATCH_MFMA && h->batch_path != MI355REC_BATCH_MFMA_NOSKIP;
    if (!streamable) {
        // nothing to stream on: the batch is served at once (complete in stream order behind this call)
        if (query_ptrs_dev) return mi355rec_enqueue_batch_mixed_keys(h, queries, query_ptrs_dev, exclude_global, batch, topn, out_keys_dev, stream);
        return mi355rec_enqueue_batch_keys(h, queries, exclude_global, batch, topn, out_keys_dev, stream);
    }
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    rc = flush_streamed(h, s);   // a stream of SINGLE queries on this handle is closed first
    if (rc) return rc;
    for (int g0 = 0; g0 < batch; g0 += kHmQueries) {
        const int nq = batch - g0 < kHmQueries ? batch - g0 : kHmQueries;
        rc = enqueue_mstream(h, queries, query_ptrs_dev, exclude_global, g0, nq, topn, out_keys_dev + static_cast<size_t>(g0) * topn, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// ---- a batch whose queries are partly vectors, partly pointers (the sharded engine's windows) ----

int mi355rec_batch_pointers_ok(const mi355rec_t* h, int topn) {
    return h && h->n >= kBqMinRows && half_multi_ok(h, topn) ? 1 : 0;
}

int mi355rec_enqueue_batch_mixed_keys(mi355rec_t* h, const float* queries, const float* const* query_ptrs_dev,
                                      const int64_t* exclude_global, int batch, int topn, mi355rec_key_t* out_keys_dev,
                                      void* stream) {
    if (!h || !out_keys_dev || (!queries && !query_ptrs_dev)) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    if (h->n < 1) {   // an empty shard answers with all-empty lists
        HIP_TRY(h, hipMemsetAsync(out_keys_dev, 0, sizeof(uint64_t) * static_cast<size_t>(batch) * topn, s));
        return MI355REC_OK;
    }
    if (batch >= 2 && mi355rec_batch_pointers_ok(h, topn) && h->batch_path != MI355REC_BATCH_MULTI) {
        for (int b = 0; b < batch; b += kMultiChain) {
            const int count = batch - b < kMultiChain ? batch - b : kMultiChain;
            rc = enqueue_half_multi(h, queries ? queries + static_cast<size_t>(b) * kDim : nullptr,
                                    query_ptrs_dev ? query_ptrs_dev + b : nullptr, exclude_global ? exclude_global + b : nullptr,
                                    count, topn, out_keys_dev + static_cast<size_t>(b) * topn, nullptr, nullptr, s);
            if (rc) return rc;
        }
        return MI355REC_OK;
    }
    // no multi-query pass over a replica here: one scan per query, either kind of query
    for (int b = 0; b < batch; ++b) {
        const float* ptr = query_ptrs_dev ? query_ptrs_dev[b] : nullptr;
        if (!ptr && !queries) return fail(h, MI355REC_ERR_INVALID_ARG, "query %d has neither a vector nor a pointer", b);
        rc = enqueue_query(h, ptr, queries ? queries + static_cast<size_t>(b) * kDim : nullptr,
                           exclude_global ? exclude_global[b] : -1, topn, out_keys_dev + static_cast<size_t>(b) * topn, nullptr,
                           nullptr, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// ---- queries whose 12 floats already live in device-readable memory -----------------

int mi355rec_row_ptr(mi355rec_t* h, int64_t local_row, const float** out_dev) {
    if (!h || !out_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    *out_dev = h->d_feats + local_row * kDim;
    return MI355REC_OK;
}

int mi355rec_enqueue_ptr_keys(mi355rec_t* h, const float* query12_dev, int64_t exclude_global, int topn,
                              mi355rec_key_t* out_keys_dev, int64_t* out_idx_dev, float* out_score_dev, void* stream) {
    if (!h || !out_keys_dev || !query12_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    return enqueue_query(h, h->n > 0 ? query12_dev : nullptr, query12_dev, exclude_global, topn, out_keys_dev, out_idx_dev,
                         out_score_dev, s);
}

int mi355rec_enqueue_ptr_keys_streamed(mi355rec_t* h, const float* query12_dev, int64_t exclude_global, int topn,
                                       mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev || !query12_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    if (h->n < 1) {
        HIP_TRY(h, hipMemsetAsync(out_keys_dev, 0, sizeof(uint64_t) * static_cast<size_t>(topn), s));
        return MI355REC_OK;
    }
    return enqueue_streamed(h, query12_dev, nullptr, exclude_global, topn, out_keys_dev, s);
}

int mi355rec_enqueue_flush(mi355rec_t* h, void* stream) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = order_stream(h, s);
    if (rc) return rc;
    rc = flush_streamed(h, s);
    if (rc) return rc;
    return flush_mstream(h, s);
}

int mi355rec_enqueue_batch_keys(mi355rec_t* h, const float* queries, const int64_t* exclude_global,
                                int batch, int topn, mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !queries || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    return enqueue_batch(h, queries, exclude_global, batch, topn, out_keys_dev, nullptr, nullptr, s);
}

int mi355rec_enqueue_batch_keys_dev(mi355rec_t* h, const float* queries_dev, const int64_t* exclude_global_dev,
                                    int batch, int topn, mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !queries_dev || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    if (topn <= 0 || topn > kMultiMaxTopK)
        return fail(h, MI355REC_ERR_INVALID_ARG, "topn must be in [1, %d] for device-resident batches, got %d",
                    kMultiMaxTopK, topn);
    if (h->n < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "empty shard: use the host-query entry point");
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = order_stream(h, s);
    if (rc) return rc;
    rc = ensure_bq(h);
    if (rc) return rc;
    static_assert(sizeof(long long) == sizeof(int64_t), "exclude ids are passed through unchanged");
    for (int b0 = 0; b0 < batch; b0 += kBqMaxQueries) {
        const int count = batch - b0 < kBqMaxQueries ? batch - b0 : kBqMaxQueries;
        rc = enqueue_bq_chunk(h, queries_dev + static_cast<size_t>(b0) * kDim,
                              exclude_global_dev ? reinterpret_cast<const long long*>(exclude_global_dev) + b0 : nullptr,
                              count, topn, out_keys_dev + static_cast<size_t>(b0) * topn, nullptr, nullptr, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

int mi355rec_batched_last_counters(mi355rec_t* h, int32_t* special_rows, int32_t* queued_queries,
                                   int64_t* candidates_total, int32_t* candidates_max) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (!h->bq.ready) return fail(h, MI355REC_ERR_INVALID_ARG, "no batched call has been made on this handle");
    DeviceGuard guard(h->device);
    HIP_TRY(h, hipDeviceSynchronize());
    int counters[4] = {0, 0, 0, 0};
    std::vector<int> cand(kBqMaxQueries);
    std::vector<uint32_t> flags(kBqMaxQueries);
    HIP_TRY(h, hipMemcpy(counters, h->bq.counters, sizeof counters, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(cand.data(), h->bq.cand_examined, sizeof(int) * kBqMaxQueries, hipMemcpyDeviceToHost));   // rows, not records
    HIP_TRY(h, hipMemcpy(flags.data(), h->bq.qflags, sizeof(uint32_t) * kBqMaxQueries, hipMemcpyDeviceToHost));
    int64_t total = 0;
    int mx = 0;
    for (int q = 0; q < h->bq.last_count; ++q) {   // slots past the last chunk hold an earlier chunk's values, or nothing
        if (flags[q] != kBqFlagOk) continue;       // queued to the exact scan
        total += cand[q];
        if (cand[q] > mx) mx = cand[q];
    }
    if (special_rows) *special_rows = counters[0];
    if (queued_queries) *queued_queries = counters[1];
    if (candidates_total) *candidates_total = total;
    if (candidates_max) *candidates_max = mx;
    return MI355REC_OK;
}

int mi355rec_batched_pass2_pairs(mi355rec_t* h, int64_t* pairs_done, int64_t* pairs_total) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (!h->bq.ready) return fail(h, MI355REC_ERR_INVALID_ARG, "no batched call has been made on this handle");
    DeviceGuard guard(h->device);
    HIP_TRY(h, hipDeviceSynchronize());
    int counters[4] = {0, 0, 0, 0};
    HIP_TRY(h, hipMemcpy(counters, h->bq.counters, sizeof counters, hipMemcpyDeviceToHost));
    const int blocks = (h->bq.last_count + 31) / 32;
    int nb = 1;
    while (nb < blocks) nb *= 2;
    const int64_t total = ((h->n + 63) / 64) * nb;
    if (pairs_total) *pairs_total = total;
    // (0 = the last chunk ran without tile maxima: every pair was looked at)
    if (pairs_done) *pairs_done = counters[3] > 0 ? counters[3] : total;
    return MI355REC_OK;
}

int mi355rec_set_batch_path(mi355rec_t* h, int path) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (path != MI355REC_BATCH_AUTO && path != MI355REC_BATCH_MULTI && path != MI355REC_BATCH_MFMA && path != MI355REC_BATCH_HALF &&
        path != MI355REC_BATCH_Q8 && path != MI355REC_BATCH_MFMA_NOSKIP)
        return fail(h, MI355REC_ERR_INVALID_ARG, "unknown batch path %d", path);
#ifndef MI355REC_EXPERIMENTS
    if (path == MI355REC_BATCH_Q8)
        return fail(h, MI355REC_ERR_INVALID_ARG, "MI355REC_BATCH_Q8 (the 8-bit front end of the multi-query pass) exists in MI355REC_EXPERIMENTS builds only");
#endif
    h->batch_path = path;
    return MI355REC_OK;
}

namespace {
// A member of a group of lanes whose group got its replicas AFTER this member was made (built on demand by another member):
// take them — the member's own sample / cutoff state, the group's rows and margins.  No-op otherwise.
int adopt_group_replica(mi355rec* h) {
    if (!h->shared || h->d_half || !h->shared->d_half) return MI355REC_OK;
    DeviceGuard guard(h->device);
    const int rc = alloc_replica_state(h);
    if (rc == MI355REC_OK && hipStreamSynchronize(h->stream) != hipSuccess)
        return fail(h, MI355REC_ERR_HIP, "hipStreamSynchronize(replica state)");
    if (rc != MI355REC_OK) {
        void* bufs[] = {h->d_half_mseed, h->d_half_rescored, h->d_half_mcuts, h->d_half_mctl};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
        h->d_half_mseed = nullptr;
        h->d_half_rescored = nullptr;
        h->d_half_mcuts = nullptr;
        h->d_half_mctl = nullptr;
        return rc;
    }
    h->d_half = static_cast<decltype(h->d_half)>(h->shared->d_half);
    h->d_q8 = static_cast<decltype(h->d_q8)>(h->shared->d_q8);
    h->margin_mix = h->shared->margin_mix;
    h->margin_mfma = h->shared->margin_mfma;
    return MI355REC_OK;
}
}  // namespace

int mi355rec_set_replica(mi355rec_t* h, int mode) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (mode != MI355REC_REPLICA_AUTO && mode != MI355REC_REPLICA_OFF && mode != MI355REC_REPLICA_ON && mode != MI355REC_REPLICA_FP16)
        return fail(h, MI355REC_ERR_INVALID_ARG, "unknown replica mode %d", mode);
#ifndef MI355REC_EXPERIMENTS
    if (mode == MI355REC_REPLICA_FP16)
        return fail(h, MI355REC_ERR_INVALID_ARG, "MI355REC_REPLICA_FP16 (single queries over the fp16 replica) exists in MI355REC_EXPERIMENTS builds only");
#endif
    if ((mode == MI355REC_REPLICA_ON || mode == MI355REC_REPLICA_FP16) && !h->d_half && h->n > 0) {
        if (!h->replica_allowed)
            return fail(h, MI355REC_ERR_INVALID_ARG, "this handle was created without a replica (MI355REC_CREATE_NO_REPLICA)");
        // a small shard (or one whose replica could not be allocated at create): build it now — or, in a group of lanes
        // where another member has built it since, take the group's
        int rc = adopt_group_replica(h);
        if (rc) return rc;
        if (!h->d_half) {
            rc = mi355rec_rebuild_replica(h);
            if (rc) return rc;
        }
        preallocate_stream_state(h);
    }
    h->replica_mode = mode;
    return MI355REC_OK;
}

#ifdef MI355REC_TEST_HOOKS   // libmi355rec_testhooks.so / the experiments build (build.py); NOT in the product library
// Test hook for the hand-offs that must fail safe (replica.hip.h): see include/mi355rec_diag.h.
int mi355rec_debug_handoff(mi355rec_t* h, int flags) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    DeviceGuard guard(h->device);
    if (flags & MI355REC_DEBUG_HANDOFF_POISON) {
        HIP_TRY(h, hipDeviceSynchronize());
        // what a reader would find if the stores it depends on had not landed: the values of EARLIER queries (the
        // buffers alternate), here the most hostile ones — a perfect score under each of the last three epochs
        const uint32_t one = score_to_ordered(1.0f);
        const size_t per = static_cast<size_t>(kSampleSlots);   // (the neighbourhood's slot included)
        std::vector<unsigned long long> vals(static_cast<size_t>(kHmSampleSlots));
        for (size_t i = 0; i < vals.size(); ++i)
            vals[i] = (static_cast<unsigned long long>(h->epoch_ctr - 1u - static_cast<uint32_t>(i % 3)) << 32) | one;
        unsigned long long* single[] = {h->d_half_seed, h->d_stream_seed[0], h->d_stream_seed[1]};
        for (unsigned long long* b : single)
            if (b) HIP_TRY(h, hipMemcpy(b, vals.data(), sizeof(unsigned long long) * per, hipMemcpyHostToDevice));
        unsigned long long* multi[] = {h->d_half_mseed, h->d_mstream_seed[0], h->d_mstream_seed[1]};
        for (unsigned long long* b : multi)
            if (b) HIP_TRY(h, hipMemcpy(b, vals.data(), sizeof(unsigned long long) * vals.size(), hipMemcpyHostToDevice));
        // ... and a cutoff of +1.0 (it would rule out every row) under the epoch before the current one
        unsigned int one_bits;
        const float onef = 1.0f;
        std::memcpy(&one_bits, &onef, sizeof one_bits);
        const unsigned long long stale_cut = (static_cast<unsigned long long>(h->epoch_ctr - 1u) << 32) | one_bits;
        if (h->d_stream_ctl) {
            SeedCtl ctl[2];
            HIP_TRY(h, hipMemcpy(ctl, h->d_stream_ctl, sizeof ctl, hipMemcpyDeviceToHost));
            ctl[0].cutoff = ctl[1].cutoff = stale_cut;   // (the arrival counters stay what they are)
            HIP_TRY(h, hipMemcpy(h->d_stream_ctl, ctl, sizeof ctl, hipMemcpyHostToDevice));
        }
        if (h->d_lone_ctl) {
            SeedCtl ctl;
            HIP_TRY(h, hipMemcpy(&ctl, h->d_lone_ctl, sizeof ctl, hipMemcpyDeviceToHost));
            ctl.cutoff = stale_cut;
            HIP_TRY(h, hipMemcpy(h->d_lone_ctl, &ctl, sizeof ctl, hipMemcpyHostToDevice));
        }
        std::vector<unsigned long long> cuts(2 * kHmQueries, stale_cut);
        if (h->d_mstream_cuts)
            HIP_TRY(h, hipMemcpy(h->d_mstream_cuts, cuts.data(), sizeof(unsigned long long) * cuts.size(), hipMemcpyHostToDevice));
        if (h->d_half_mcuts)
            HIP_TRY(h, hipMemcpy(h->d_half_mcuts, cuts.data(), sizeof(unsigned long long) * kHmQueries, hipMemcpyHostToDevice));
    }
    if (flags & MI355REC_DEBUG_HANDOFF_DROP_STORES) h->dbg_skip_regions = kHalfSeedMaxGrid / 2;
    if (flags & MI355REC_DEBUG_HANDOFF_NO_LAST_RIDER) h->dbg_no_last = true;
    return MI355REC_OK;
}

#endif   // MI355REC_TEST_HOOKS

#ifdef MI355REC_PHASE_CLOCK   // tools/phase_clock.py builds only
int mi355rec_debug_phase_clock(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return MI355REC_ERR_HIP;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mi355::g_phase_clock), sizeof(unsigned long long) * n_words) == hipSuccess
               ? MI355REC_OK : MI355REC_ERR_HIP;
}
#endif

int mi355rec_replica_counters(mi355rec_t* h, int64_t* scans, int64_t* rescored_rows) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (scans) *scans = h->half_scans;
    if (rescored_rows) *rescored_rows = 0;
    if (!h->d_half_rescored || !rescored_rows) return MI355REC_OK;
    DeviceGuard guard(h->device);
    std::vector<unsigned long long> slots(kRideMaxLists);
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(slots.data(), h->d_half_rescored, sizeof(unsigned long long) * kRideMaxLists, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (unsigned long long v : slots) sum += v;
    *rescored_rows = static_cast<int64_t>(sum);
    return MI355REC_OK;
}

int mi355rec_rebuild_replica(mi355rec_t* h) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->n == 0) return MI355REC_OK;
    if (!h->replica_allowed)
        return fail(h, MI355REC_ERR_INVALID_ARG, "this handle was created without a replica (MI355REC_CREATE_NO_REPLICA)");
    // In a group of lanes the replicas belong to the group.  While the group HAS none (a shard under 65 536 rows gets none at
    // create) any member may build them — the others take them over when they are next told to use them
    // (adopt_group_replica); once it has them they are never rebuilt under the other members' feet.
    if (h->shared && h->shared->d_half)
        return fail(h, MI355REC_ERR_INVALID_ARG, "the replicas are shared with lanes (mi355rec_create_lane): destroy the lanes, rebuild, create them again");
    DeviceGuard guard(h->device);
    int rc = sync_api_begin(h);
    if (rc) return rc;
    // A stashed streamed query carries a sample taken from the OLD replica, and the pending one's
    // lists wait for their merge: both are completed first (on the handle's own stream).
    rc = flush_streamed(h, h->stream);
    if (rc) return rc;
    rc = flush_mstream(h, h->stream);
    if (rc) return rc;
    rc = build_anchors(h);   // (the other snapshot of the rows)
    if (rc) return rc;
    rc = build_replica(h);
    if (rc == MI355REC_OK && h->shared) {   // published to the group
        h->shared->d_half = h->d_half;
        h->shared->d_q8 = h->d_q8;
        h->shared->margin_mix = h->margin_mix;
        h->shared->margin_mfma = h->margin_mfma;
    }
    return rc;
}

int mi355rec_enqueue_merge_keys(mi355rec_t* h, const mi355rec_key_t* lists_dev, int n_lists,
                                int list_len, int topn, mi355rec_key_t* out_keys_dev,
                                int64_t* out_idx_dev, float* out_score_dev, void* stream) {
    if (!h || !lists_dev || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (n_lists < 1 || n_lists > kMergeMaxLists || list_len < 1)
        return fail(h, MI355REC_ERR_INVALID_ARG, "n_lists %d / list_len %d out of range", n_lists, list_len);
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    rc = order_stream(h, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    return enqueue_merge(h, lists_dev, n_lists, list_len, topn, out_keys_dev, out_idx_dev,
                         out_score_dev, static_cast<hipStream_t>(stream));
}

int mi355rec_enqueue_merge_keys_batch(mi355rec_t* h, const mi355rec_key_t* lists_dev, int n_lists,
                                      int list_len, int64_t list_stride, int64_t query_stride, int batch,
                                      int topn, mi355rec_key_t* out_keys_dev, int64_t* out_idx_dev,
                                      float* out_score_dev, void* stream) {
    if (!h || !lists_dev || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (n_lists < 1 || n_lists > kMergeMaxLists || list_len < 1 || batch < 1 || list_stride < list_len)
        return fail(h, MI355REC_ERR_INVALID_ARG, "merge geometry out of range");
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    hipLaunchKernelGGL(merge_kernel, dim3(batch), dim3(kMergeBlock), 0, s, lists_dev, n_lists, list_len,
                       list_stride, query_stride, topn, out_keys_dev, out_idx_dev, out_score_dev,
                       static_cast<int64_t>(topn));
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

int mi355rec_enqueue_scores(mi355rec_t* h, int64_t local_row, const float* query12,
                            float* out_scores_dev, void* stream) {
    if (!h || !out_scores_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row >= h->n) return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    if (local_row < 0 && !query12) return fail(h, MI355REC_ERR_INVALID_ARG, "null query");
    if (h->n == 0) return MI355REC_OK;  // empty shard: no scores
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = order_stream(h, s);
    if (rc) return rc;
    QueryArg qa;
    std::memset(&qa, 0, sizeof qa);
    qa.margin = h->margin_mix;
    NextSeed no_next;
    std::memset(&no_next, 0, sizeof no_next);
    if (local_row >= 0) {
        hipLaunchKernelGGL((scan_kernel<ScanConfig, true, true>), dim3(h->grid), dim3(kScanBlock), 0, s, h->d_feats,
                           h->n, h->rows_per_block, h->iters, h->row_base, qa, h->d_feats + local_row * kDim,
                           static_cast<int64_t>(-1), 1, static_cast<uint64_t*>(nullptr), out_scores_dev,
                           static_cast<const uint64_t*>(nullptr), PrevMerge{nullptr, 0, 0, nullptr},
                           static_cast<const unsigned long long*>(nullptr), static_cast<const unsigned long long*>(nullptr), 0u, no_next);
    } else {
        std::memcpy(qa.q, query12, sizeof qa.q);
        hipLaunchKernelGGL((scan_kernel<ScanConfig, false, true>), dim3(h->grid), dim3(kScanBlock), 0, s, h->d_feats,
                           h->n, h->rows_per_block, h->iters, h->row_base, qa, kNoQueryPtr,
                           static_cast<int64_t>(-1), 1, static_cast<uint64_t*>(nullptr), out_scores_dev,
                           static_cast<const uint64_t*>(nullptr), PrevMerge{nullptr, 0, 0, nullptr},
                           static_cast<const unsigned long long*>(nullptr), static_cast<const unsigned long long*>(nullptr), 0u, no_next);
    }
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

int mi355rec_enqueue_stream_probe(mi355rec_t* h, uint32_t* sink_dev, void* stream) {
    return mi355rec_enqueue_stream_probe_of(h, MI355REC_PROBE_FP32_ROWS, sink_dev, stream);
}

int mi355rec_enqueue_stream_probe_of(mi355rec_t* h, int which, uint32_t* sink_dev, void* stream) {
    if (!h || !sink_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (h->n == 0) return MI355REC_OK;
    const float4* buf = nullptr;
    int64_t n_vec = 0;   // 16-byte vectors
    if (which == MI355REC_PROBE_FP32_ROWS) {
        buf = reinterpret_cast<const float4*>(h->d_feats);
        n_vec = h->n * 3;
    } else if (which == MI355REC_PROBE_FP16_REPLICA && h->d_half) {
        buf = reinterpret_cast<const float4*>(h->d_half);
        n_vec = ((h->n + 1) / 2) * 3;
    } else if (which == MI355REC_PROBE_Q8_REPLICA && h->d_q8) {
        buf = reinterpret_cast<const float4*>(h->d_q8);
        n_vec = ((h->n + 3) / 4) * 3;
    } else {
        return fail(h, MI355REC_ERR_INVALID_ARG, "no such buffer to probe (%d)", which);
    }
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, stream_probe_kernel,
                 dim3(h->cus), dim3(kProbeBlock), s, buf, n_vec, sink_dev);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

// ---- synchronous host API ------------------------------------------------------

static int scores_common(mi355rec_t* h, int64_t local_row, const float* query12, float* out_host) {
    if (!h || !out_host) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(h->device);
    if (h->n == 0) return MI355REC_OK;
    if (!h->d_scores_full) HIP_TRY(h, hipMalloc(&h->d_scores_full, sizeof(float) * static_cast<size_t>(h->n)));
    int rc = mi355rec_enqueue_scores(h, local_row, query12, h->d_scores_full, h->stream);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(out_host, h->d_scores_full, sizeof(float) * static_cast<size_t>(h->n),
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MI355REC_OK;
}

int mi355rec_fetch_row(mi355rec_t* h, int64_t local_row, float* out12_host) {
    if (!h || !out12_host) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    DeviceGuard guard(h->device);
    HIP_TRY(h, hipMemcpy(out12_host, h->d_feats + local_row * kDim, sizeof(float) * kDim, hipMemcpyDeviceToHost));
    return MI355REC_OK;
}

int mi355rec_scores_row(mi355rec_t* h, int64_t local_row, float* out_host) {
    if (h && (local_row < 0 || local_row >= h->n))
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    return scores_common(h, local_row, nullptr, out_host);
}

int mi355rec_scores(mi355rec_t* h, const float* query12, float* out_host) {
    if (!query12) return fail(h, MI355REC_ERR_INVALID_ARG, "null query");
    return scores_common(h, -1, query12, out_host);
}

int mi355rec_query_batch_topn(mi355rec_t* h, const float* queries, int batch,
                              const int64_t* exclude_global, int topn, int64_t* out_idx,
                              float* out_score, int* out_count) {
    if (!h || !queries || !out_idx) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    // internal lists are `eff` long: a shard of n rows cannot return more
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    const size_t cnt = static_cast<size_t>(batch) * eff;
    rc = ensure_slots(h, cnt);
    if (rc) return rc;
    rc = sync_api_begin(h);
    if (rc) return rc;
    const bool direct = cnt <= static_cast<size_t>(kDirectResultSlots);
    rc = enqueue_batch(h, queries, exclude_global, batch, eff, h->d_keys, direct ? h->hd_idx : h->d_idx,
                       direct ? h->hd_score : h->d_score, h->stream);
    if (rc) return rc;
    if (!direct) {
        HIP_TRY(h, hipMemcpyAsync(h->h_idx, h->d_idx, cnt * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->h_score, h->d_score, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < batch; ++b) {
        const int64_t* src_i = h->h_idx + static_cast<size_t>(b) * eff;
        const float* src_s = h->h_score + static_cast<size_t>(b) * eff;
        int64_t* dst_i = out_idx + static_cast<size_t>(b) * topn;
        std::memcpy(dst_i, src_i, static_cast<size_t>(eff) * sizeof(int64_t));
        for (int i = eff; i < topn; ++i) dst_i[i] = -1;
        if (out_score) {
            float* dst_s = out_score + static_cast<size_t>(b) * topn;
            std::memcpy(dst_s, src_s, static_cast<size_t>(eff) * sizeof(float));
            for (int i = eff; i < topn; ++i) dst_s[i] = 0.0f;
        }
        if (out_count) {
            int c = 0;
            while (c < eff && src_i[c] >= 0) ++c;
            out_count[b] = c;
        }
    }
    return MI355REC_OK;
}

namespace {
// One query, synchronously, results in the caller's host buffers: the query is resident memory (`qptr`: a row of this handle)
// or 12 floats by value (`query12`).  What Recommender::recommendByIndex sits on (Recommender.cu:275-318) — and, since round 6,
// mi355rec_query_topn as well: a query by value used to go through the batch entry point with a batch of one (no completion
// word, a stream synchronise, two D2H copies: ~9 us more per call at 10 M rows).
int sync_single_query(mi355rec_t* h, const float* qptr, const float* query12, int64_t exclude_global, int topn, int64_t* out_idx,
                      float* out_score, int* out_count) {
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    // device / pinned slots are sized by what the shard can return, not by topn
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    rc = ensure_slots(h, static_cast<size_t>(eff));
    if (rc) return rc;
    rc = sync_api_begin(h);
    if (rc) return rc;
    // Small results go straight into the pinned host buffers from the merge kernel
    // (zero-copy stores over PCIe: no D2H copy launches on the latency path).
    const bool direct = eff <= kDirectResultSlots;
    // one round (eff <= 1024): the merge kernel stores the results AND a completion word in pinned host
    // memory; the host spins on the word
    const bool notify = direct && eff <= kMaxTopK && eff > 0;
    const uint32_t want = notify ? (++h->done_seq ? h->done_seq : ++h->done_seq) : 0u;   // never 0
    rc = enqueue_query(h, qptr, query12, exclude_global, eff, h->d_keys, direct ? h->hd_idx : h->d_idx,
                       direct ? h->hd_score : h->d_score, h->stream, want);
    if (rc) return rc;
    if (!direct) {
        HIP_TRY(h, hipMemcpyAsync(h->h_idx, h->d_idx, eff * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->h_score, h->d_score, eff * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    if (notify) {
        rc = wait_done(h, want);
        if (rc) return rc;
    } else {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    int c = 0;
    while (c < eff && h->h_idx[c] >= 0) ++c;
    std::memcpy(out_idx, h->h_idx, static_cast<size_t>(eff) * sizeof(int64_t));
    if (out_score) std::memcpy(out_score, h->h_score, static_cast<size_t>(eff) * sizeof(float));
    for (int i = eff; i < topn; ++i) {
        out_idx[i] = -1;
        if (out_score) out_score[i] = 0.0f;
    }
    if (out_count) *out_count = c;
    return MI355REC_OK;
}
}  // namespace

int mi355rec_query_topn(mi355rec_t* h, const float* query12, int64_t exclude_global, int topn,
                        int64_t* out_idx, float* out_score, int* out_count) {
    if (!h || !query12 || !out_idx) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (h->n == 0) return mi355rec_query_batch_topn(h, query12, 1, &exclude_global, topn, out_idx, out_score, out_count);   // (an empty shard: all-empty lists)
    return sync_single_query(h, nullptr, query12, exclude_global, topn, out_idx, out_score, out_count);
}

int mi355rec_query_row_topn(mi355rec_t* h, int64_t local_row, int topn, int64_t* out_idx,
                            float* out_score, int* out_count) {
    if (!h || !out_idx) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    return sync_single_query(h, h->d_feats + local_row * kDim, nullptr, h->row_base + local_row, topn, out_idx, out_score, out_count);
}

// ---- key helpers -----------------------------------------------------------------

mi355rec_key_t mi355rec_pack_key(float score, int64_t global_row) {
    return pack_key(score, static_cast<uint32_t>(global_row));
}

float mi355rec_key_score(mi355rec_key_t key) {
    return key ? ordered_to_score(static_cast<uint32_t>(key >> 32)) : 0.0f;
}

int64_t mi355rec_key_row(mi355rec_key_t key) {
    return key ? static_cast<int64_t>(static_cast<uint32_t>(~static_cast<uint32_t>(key))) : -1;
}

}  // extern "C"
