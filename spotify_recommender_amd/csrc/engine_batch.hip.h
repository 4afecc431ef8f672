// engine_batch.hip.h — MANY queries per call: the exact 12-query pass over the fp32 rows (shards without a replica), one pass
// over the fp16 replica for 2 ... 32 queries and the stream of such batches (replica_multi.hip.h), the two-pass matrix-core
// path for up to 1024 queries per chunk (batched.hip.h: BASELINE configs[4]), and the dispatch between them.  The reference
// answers one query per process (main.cpp:46-131); per query the contract stays recommendByIndex's (Recommender.cu:275-318).
// (Part of mi355rec.hip's translation unit.)
#pragma once

#include "engine_single.hip.h"

namespace {

// Multi-query passes for up to kMultiChain queries: ONE cheap seed (approximate
// scores of a spread ~2.6 % sample -> a chip-wide starting threshold per query),
// then per group of kMultiQueries the full pass (the catalogue is streamed once per
// group), then ONE merge launch with a workgroup per query.  topn <= kMultiMaxTopK.
int enqueue_multi(mi355rec* h, const float* queries, const int64_t* exclude, int count, int topn,
                  uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    MultiQueryArg qa[kMultiChain / kMultiQueries];
    const int groups = (count + kMultiQueries - 1) / kMultiQueries;
    for (int g = 0; g < groups; ++g) {
        std::memset(&qa[g], 0, sizeof qa[g]);
        for (int q = 0; q < kMultiQueries; ++q) {
            const int src = g * kMultiQueries + q;
            qa[g].exclude[q] = -1;
            if (src < count) {
                std::memcpy(qa[g].q[q], queries + static_cast<size_t>(src) * kDim, sizeof(float) * kDim);
                if (exclude) qa[g].exclude[q] = exclude[src];
            }
        }
    }
    const int64_t list_stride = static_cast<int64_t>(h->mgrid) * topn;
    const int seed_count = h->mgrid * kSeedWaves;
    const bool seeded = h->miters >= 3 && seed_count >= topn && seed_count <= kMergeBlock * kSeedSelectPerThread;
    if (seeded) {
        // one cheap launch for the whole chain: approximate scores of a spread
        // 2.6 % sample, then the per-query bound (kernels.hip.h, "seed")
        SeedQueryArg sq;
        std::memset(&sq, 0, sizeof sq);
        for (int q = 0; q < kMultiChain; ++q) sq.exclude[q] = -1;
        for (int q = 0; q < count; ++q) {
            std::memcpy(sq.q[q], queries + static_cast<size_t>(q) * kDim, sizeof(float) * kDim);
            if (exclude) sq.exclude[q] = exclude[q];
        }
        hipLaunchKernelGGL(seed_multi_kernel, dim3(h->mgrid), dim3(kSeedBlock), 0, s, h->d_feats, h->n,
                           h->mrows_per_block, h->row_base, sq, count, h->d_seed_vals);
        hipLaunchKernelGGL(seed_select_kernel, dim3(count), dim3(kMergeBlock), 0, s, h->d_seed_vals, seed_count,
                           topn, h->d_seed_keys);
    }
    for (int g = 0; g < groups; ++g) {
        const int nq = count - g * kMultiQueries < kMultiQueries ? count - g * kMultiQueries : kMultiQueries;
        ++h->routes.multi_fp32;
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_multi_kernel<MultiConfig>),
                     dim3(h->mgrid), dim3(MultiConfig::kBlock), s,
                     h->d_feats, h->n, static_cast<int64_t>(0), static_cast<int64_t>(0), h->miters, h->row_base,
                     qa[g], nq, g * kMultiQueries, topn, h->d_block_lists,
                     seeded ? h->d_seed_keys : static_cast<const uint64_t*>(nullptr));
    }
    HIP_TRY(h, hipGetLastError());
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(kMergeBlock), 0, s, h->d_block_lists, h->mgrid, topn,
                       static_cast<int64_t>(topn), list_stride, topn, out_keys, out_idx, out_score, static_cast<int64_t>(topn));
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

// Multi-query passes over the fp16 replica (replica_multi.hip.h): per group of up to kHmQueries
// queries ONE sample launch + ONE pass over the 24 B/row replica, then one merge launch with a
// workgroup per query for the whole chain.  queries[i] by value, or qptrs[i] != null: where its 12
// floats live in device-readable memory.  topn <= kMultiMaxTopK, count <= kMultiChain.
// Which replica a multi-query pass streams.  The 8-bit front end (12 B/row, integer matrix core, fp16 re-check of
// its candidates) moves half the bytes but its bound is 25x the fp16 one: ~1 % of the (row, query) pairs come
// back as candidates, 3.7 us per query of a pass against 0.85 us (measured, 10 M rows: 1 query 36.9 vs 44.1 us,
// 2: 41.9 vs 44.7, 12: 82 vs 53, 32: 152 vs 71).  So: passes of one or two queries, or when forced.
// Round 5: the front end is an A/B route of experiment builds (its one AUTO cell, passes of two queries, was worth
// 2.6 us per call and a second instantiation of the pass kernel to keep bit-identical).
bool multi_front_q8(const mi355rec* h, int nq) {
#ifdef MI355REC_EXPERIMENTS
    if (!use_q8(h) || h->batch_path == MI355REC_BATCH_HALF) return false;
    return h->batch_path == MI355REC_BATCH_Q8 || nq <= 2;
#else
    (void)h;
    (void)nq;
    return false;
#endif
}

bool half_multi_ok(const mi355rec* h, int topn) {
    return h->d_half && h->replica_mode != MI355REC_REPLICA_OFF && topn <= kMultiMaxTopK && h->hg.seed_grid > 0 &&
           h->hg.seed_grid * kHalfSeedWaves >= topn;
}

void fill_half_multi_arg(HalfMultiArg& arg, const mi355rec* h, const float* queries, const float* const* qptrs, const int64_t* exclude,
                         int g0, int nq) {
    std::memset(&arg, 0, sizeof arg);
    arg.margin = h->margin_mfma;
    arg.anchors = h->d_anchor;
    for (int q = 0; q < kHmQueries; ++q) {
        arg.exclude[q] = -1;
        if (q >= nq) continue;
        if (qptrs && qptrs[g0 + q]) {
            hm_set_pointer(arg, q, qptrs[g0 + q]);
        } else if (queries) {
            std::memcpy(arg.q[q], queries + static_cast<size_t>(g0 + q) * kDim, sizeof(float) * kDim);
        }
        if (exclude) arg.exclude[q] = exclude[g0 + q];
    }
}

// How much of the shard a batch of nq queries samples for its cutoffs (replica_multi.hip.h, hm_sample_regions):
// regions of 1024 << l rows.  The sample is paid once per batch, the candidates its cutoff lets through once per
// query: 2.6 % of 10 M rows leave ~5 100 candidates per query, 5 % ~2 700, 10 % ~1 400.  A sample launch of its own
// is over in a few us whatever it reads; seed riders (`riding`) share the memory system with the pass they ride in,
// row for row, so a streamed batch samples at most 5 % (measured, tools/hm_riders.sh, 10 M rows x 12 queries: launches
// of 45.1 / 43.6 / 45.1 us at 2.6 / 5 / 10 %; x 32 queries: 52.3 / 49.7 / 50.2).  Regions must not overlap.
int hm_sample_log2(const mi355rec* h, int nq, bool riding) {
    int l = nq >= 12 ? 2 : nq >= 5 ? 1 : 0;
    if (riding && l > 1) l = 1;
    MI355REC_EXP_INT(l, "MI355REC_EXP_SAMPLE_LOG2", 0, 3);
    while (l > 0 && (static_cast<int64_t>(1024) << l) > h->hg.seed_stride) --l;
    return l;
}

int enqueue_half_multi(mi355rec* h, const float* queries, const float* const* qptrs, const int64_t* exclude, int count,
                       int topn, uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    const int n_seed = h->hg.seed_grid * kHalfSeedWaves;
    HmRide no_ride;
    std::memset(&no_ride, 0, sizeof no_ride);
    HalfMultiArg arg;
    for (int g0 = 0; g0 < count; g0 += kHmQueries) {
        const int nq = count - g0 < kHmQueries ? count - g0 : kHmQueries;
        fill_half_multi_arg(arg, h, queries, qptrs, exclude, g0, nq);
        const uint32_t epoch = next_epoch(h);
        // the sample launch's last workgroup selects the cutoffs; the pass reads them (stream order)
        const unsigned long long* const cuts = h->d_half_mcuts;
        // (+ one workgroup per query for its neighbourhood's bound: handoff.hip.h)
        hipLaunchKernelGGL(seed_half_multi_kernel, dim3(h->hg.seed_grid + nq), dim3(kHmBlock), 0, s, h->d_feats, h->d_half, h->n, h->row_base,
                           h->hg.seed_stride, arg, nq, h->hg.seed_grid, h->d_half_mseed, epoch, hm_sample_log2(h, nq, false), h->d_half_mctl,
                           h->half_mctl_done + (h->dbg_no_last ? 0x40000000u : 0u), h->d_half_mcuts, topn, h->dbg_skip_regions);
        HIP_TRY(h, hipGetLastError());
        h->half_mctl_done += static_cast<unsigned>(h->hg.seed_grid);
        h->dbg_no_last = false;   // (test hooks of mi355rec_debug_handoff: they apply to ONE sampling launch)
        h->dbg_skip_regions = 0;
        ++h->half_scans;
#ifdef MI355REC_EXPERIMENTS
        if (multi_front_q8(h, nq)) {   // rows from the 8-bit replica through the integer matrix core (replica_multi.hip.h)
            ++h->q8_scans;
            ++h->routes.multi_q8;
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_multi_kernel<false, true>), dim3(h->hg.grid),
                         dim3(kHmBlock), s, h->d_feats, h->d_half, reinterpret_cast<const uint32_t*>(h->d_q8), h->n, h->row_base, arg, nq,
                         g0, topn, h->d_block_lists, h->d_half_mseed, n_seed, h->d_half_rescored, no_ride, arg, cuts, epoch);
        } else
#endif
        {
            ++h->routes.multi_fp16;
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_multi_kernel<false, false>), dim3(h->hg.grid),
                         dim3(kHmBlock), s, h->d_feats, h->d_half, static_cast<const uint32_t*>(nullptr), h->n, h->row_base, arg, nq,
                         g0, topn, h->d_block_lists, h->d_half_mseed, n_seed, h->d_half_rescored, no_ride, arg, cuts, epoch);
        }
    }
    HIP_TRY(h, hipGetLastError());
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(kMergeBlock), 0, s, h->d_block_lists, h->hg.grid, topn,
                       static_cast<int64_t>(topn), static_cast<int64_t>(h->hg.grid) * topn, topn, out_keys, out_idx, out_score,
                       static_cast<int64_t>(topn));
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

// ---- a STREAM of batches over the replica (mi355rec_enqueue_batch_keys_streamed) ---------------------
// The single-query stream's scheme (enqueue_streamed), one level up: the stream runs one call behind —
// call k + 1 LAUNCHES batch k — and that launch carries, beside its scanners, a merging workgroup per three
// queries of batch k - 1 and a few seed riders that take the sample of batch k + 1.  A stream of K batches
// costs K launches + one sample launch at its head + one merge launch at its tail (the flush).
// Workgroups of a streamed launch that do not scan take a scanner's place among the resident ones (measured at 10 M rows:
// 32 mergers + 64 riders of 512 made a 41 us pass 57 us), so they are as few as can still finish inside the pass:
constexpr int kHmRiders = 16;       // seed riders per 1024 rows of a sampled region: a rider's wave gets through a 128-row
                                    // chunk every ~2 us beside a pass (as a scanner's does), 16 (or 32) of them take 33 us
constexpr int kHmMergesPerWg = 3;   // queries of the previous batch one merging workgroup takes, one after the other (~10 us each)
constexpr int kHmNbhdPerWg = 4;     // queries of the next batch one neighbourhood workgroup takes, one after the other (~7 us each beside a
                                    // pass; with 8 the workgroup outlasted it: a 32-query launch 51.4 -> 58.7 us)

int ensure_mstream(mi355rec* h) {
    if (h->mstream_ready) return MI355REC_OK;
    const size_t list_bytes = sizeof(uint64_t) * static_cast<size_t>(kHmQueries) * h->hg.grid * kMultiMaxTopK;
    const size_t seed_bytes = sizeof(unsigned long long) * static_cast<size_t>(kHmSampleSlots);
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc(&h->d_mstream_lists[i], list_bytes);
        if (e == hipSuccess) e = hipMalloc(&h->d_mstream_seed[i], seed_bytes);
    }
    if (e == hipSuccess) e = hipMalloc(&h->d_mstream_cuts, sizeof(unsigned long long) * 2 * kHmQueries);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_mstream_cuts, 0, sizeof(unsigned long long) * 2 * kHmQueries, h->stream);
    if (e == hipSuccess) e = hipMalloc(&h->d_mstream_ctl, sizeof(SeedCtl) * 2);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_mstream_ctl, 0, sizeof(SeedCtl) * 2, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {   // all or nothing
        if (h->d_mstream_cuts) (void)hipFree(h->d_mstream_cuts);
        if (h->d_mstream_ctl) (void)hipFree(h->d_mstream_ctl);
        h->d_mstream_cuts = nullptr;
        h->d_mstream_ctl = nullptr;
        for (int i = 0; i < 2; ++i) {
            if (h->d_mstream_lists[i]) (void)hipFree(h->d_mstream_lists[i]);
            if (h->d_mstream_seed[i]) (void)hipFree(h->d_mstream_seed[i]);
            h->d_mstream_lists[i] = nullptr;
            h->d_mstream_seed[i] = nullptr;
        }
        return fail(h, e == hipErrorOutOfMemory ? MI355REC_ERR_OUT_OF_MEMORY : MI355REC_ERR_HIP, "hipMalloc(batch stream): %s",
                    hipGetErrorString(e));
    }
    h->mctl_done[0] = h->mctl_done[1] = 0u;
    h->mstream_ready = true;
    return MI355REC_OK;
}

// Launches the stashed batch: scanners + the mergers of the batch before it + (next != null) the seed
// riders of the batch after it.
int launch_mstash(mi355rec* h, hipStream_t s, const HalfMultiArg* next, int next_nq, int next_topn, int next_buf, uint32_t next_epoch_tag) {
    auto& st = h->mstash;
    const int buf = h->mpending.has ? 1 - h->mpending.buf : 0;
    HmRide ride;
    std::memset(&ride, 0, sizeof ride);
    if (h->mpending.has) {
        ride.prev_lists = h->d_mstream_lists[h->mpending.buf];
        ride.prev_out = h->mpending.out;
        ride.prev_queries = h->mpending.nq;
        ride.merge_wgs = (h->mpending.nq + kHmMergesPerWg - 1) / kHmMergesPerWg;
        ride.prev_n_lists = h->mpending.n_lists;
        ride.prev_topk = h->mpending.topn;
    }
    if (next) {
        ride.sample_log2 = hm_sample_log2(h, next_nq, true);
        ride.seed_wgs = kHmRiders << ride.sample_log2;
        MI355REC_EXP_INT(ride.seed_wgs, "MI355REC_EXP_RIDERS", 1, 512);
        if (ride.seed_wgs > h->hg.seed_grid) ride.seed_wgs = h->hg.seed_grid;
        // (a query that excludes no row of this shard looks for its anchor first — handoff.hip.h, nbhd_anchor: two more round
        // trips per query — so a workgroup takes half as many of a batch that holds such queries)
        bool anchors = false;
        for (int q = 0; q < next_nq; ++q)
            anchors = anchors || next->exclude[q] < h->row_base || next->exclude[q] >= h->row_base + h->n;
        const int per_wg = anchors ? (kHmNbhdPerWg + 1) / 2 : kHmNbhdPerWg;
        ride.nb_wgs = (next_nq + per_wg - 1) / per_wg;
        ride.next_queries = next_nq;
        ride.regions = h->hg.seed_grid;
        ride.stride_rows = h->hg.seed_stride;
        ride.next_seed_vals = h->d_mstream_seed[next_buf];
        ride.next_ctl = h->d_mstream_ctl + next_buf;
        ride.next_cuts = h->d_mstream_cuts + next_buf * kHmQueries;
        ride.next_topk = next_topn;
        ride.next_epoch = next_epoch_tag;
        // the riders' arrival counter counts up and is never reset: this launch's riders start from ...
        ride.done_base = h->mctl_done[next_buf] + (h->dbg_no_last ? 0x40000000u : 0u);
        ride.debug_skip = h->dbg_skip_regions;
        h->dbg_no_last = false;
        h->dbg_skip_regions = 0;
    }
    const unsigned long long* cuts_ready = st.cuts_ready ? h->d_mstream_cuts + st.seed_buf * kHmQueries : nullptr;
    // the launch stays within one resident wave of workgroups: the riders and mergers take scanner slots
    const int others = ride.merge_wgs + ride.seed_wgs + ride.nb_wgs;
    int scanners = h->hg.grid - others;
    if (scanners < 1) scanners = 1;
    ++h->half_scans;
#ifdef MI355REC_EXPERIMENTS
    if (multi_front_q8(h, st.nq)) {
        ++h->q8_scans;
        ++h->routes.multi_q8;
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_multi_kernel<true, true>),
                     dim3(scanners + others), dim3(kHmBlock), s, h->d_feats, h->d_half,
                     reinterpret_cast<const uint32_t*>(h->d_q8), h->n, h->row_base,
                     st.arg, st.nq, 0, st.topn, h->d_mstream_lists[buf], h->d_mstream_seed[st.seed_buf],
                     h->hg.seed_grid * kHalfSeedWaves, h->d_half_rescored, ride, next ? *next : st.arg, cuts_ready, st.epoch);
    } else
#endif
    {
        ++h->routes.multi_fp16;
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_multi_kernel<true, false>),
                     dim3(scanners + others), dim3(kHmBlock), s, h->d_feats, h->d_half,
                     static_cast<const uint32_t*>(nullptr), h->n, h->row_base,
                     st.arg, st.nq, 0, st.topn, h->d_mstream_lists[buf], h->d_mstream_seed[st.seed_buf],
                     h->hg.seed_grid * kHalfSeedWaves, h->d_half_rescored, ride, next ? *next : st.arg, cuts_ready, st.epoch);
    }
    HIP_TRY(h, hipGetLastError());
    if (next) h->mctl_done[next_buf] += static_cast<unsigned>(ride.seed_wgs);   // (the books move once the launch has been accepted)
    h->mpending.has = true;
    h->mpending.buf = buf;
    h->mpending.nq = st.nq;
    h->mpending.topn = st.topn;
    h->mpending.n_lists = scanners;
    h->mpending.out = st.out;
    st.has = false;
    return MI355REC_OK;
}

int flush_mstream(mi355rec* h, hipStream_t s) {
    if (h->mstash.has) {
        const int rc = launch_mstash(h, s, nullptr, 0, 0, 0, 0u);
        if (rc) return rc;
    }
    if (!h->mpending.has) return MI355REC_OK;
    const auto& p = h->mpending;
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    hipLaunchKernelGGL(merge_kernel, dim3(p.nq), dim3(kMergeBlock), 0, s, h->d_mstream_lists[p.buf], p.n_lists, p.topn,
                       static_cast<int64_t>(p.topn), static_cast<int64_t>(p.n_lists) * p.topn, p.topn, p.out,
                       static_cast<int64_t*>(nullptr), static_cast<float*>(nullptr), static_cast<int64_t>(p.topn));
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    h->mpending.has = false;
    return MI355REC_OK;
}

// One batch of <= kHmQueries queries joins the stream.
int enqueue_mstream(mi355rec* h, const float* queries, const float* const* qptrs, const int64_t* exclude, int g0, int nq, int topn,
                    uint64_t* out_keys, hipStream_t s) {
    int rc = ensure_mstream(h);
    if (rc) return rc;
    HalfMultiArg arg;
    fill_half_multi_arg(arg, h, queries, qptrs, exclude, g0, nq);
    int seed_buf = 0;
    bool cuts_ready = false;
    const uint32_t epoch = next_epoch(h);   // the tag of this batch's sample values and cutoffs
    if (h->mstash.has) {
        seed_buf = 1 - h->mstash.seed_buf;
        rc = launch_mstash(h, s, &arg, nq, topn, seed_buf, epoch);   // its riders take THIS batch's sample (and select its cutoffs)
        if (rc) return rc;
        cuts_ready = h->hg.seed_grid > 0;   // (launch_mstash gave the launch seed riders)
    } else {   // the head of a stream: a sample launch of its own
        hipLaunchKernelGGL(seed_half_multi_kernel, dim3(h->hg.seed_grid + nq), dim3(kHmBlock), 0, s, h->d_feats, h->d_half, h->n, h->row_base,
                           h->hg.seed_stride, arg, nq, h->hg.seed_grid, h->d_mstream_seed[seed_buf], epoch, hm_sample_log2(h, nq, false),
                           h->d_mstream_ctl + seed_buf, h->mctl_done[seed_buf] + (h->dbg_no_last ? 0x40000000u : 0u),
                           h->d_mstream_cuts + seed_buf * kHmQueries, topn, h->dbg_skip_regions);
        HIP_TRY(h, hipGetLastError());
        h->mctl_done[seed_buf] += static_cast<unsigned>(h->hg.seed_grid);
        h->dbg_no_last = false;
        h->dbg_skip_regions = 0;
        cuts_ready = h->hg.seed_grid > 0;
    }
    auto& st = h->mstash;
    st.has = true;
    st.arg = arg;
    st.nq = nq;
    st.topn = topn;
    st.out = out_keys;
    st.seed_buf = seed_buf;
    st.epoch = epoch;
    st.cuts_ready = cuts_ready;
    return MI355REC_OK;
}

// ---- batched path (batched.hip.h) -----------------------------------------------

constexpr int64_t kBqMinRows = 65536;   // below this the launch count, not the arithmetic, decides
constexpr int kBqMinBatch = 13;          // without a replica: up to 12 queries are ONE exact multi-query pass (141 us at 10 M rows)
constexpr int kHmAutoMax = 32;           // up to here a batch goes in ONE multi-query pass over the replica (measured at 10 M
                                         // rows x top-100, round 4: 70 / 76 / 78 / 81 us per call for 2 / 12 / 16 / 32 queries, the
                                         // matrix-core path 88-93 for any chunk of <= 32)
constexpr int kBqMinBatchReplica = 3;    // with one, the passes cost ~92 us for any chunk of <= 32 queries (two single
                                         // replica scans cost 88): measured at 10 M rows, tools/run_batched.py

void free_bq(mi355rec* h);

int ensure_bq_alloc(mi355rec* h);

// Pass 1 looks at every step-th 64-row tile (a threshold from ANY subset of the rows is valid): 4 once each wave still
// gets a couple of dozen tiles, less on small shards.  A power of two.
int bq_step1(const mi355rec* h, int64_t n_tiles) {
    const auto& b = h->bq;
    int step1 = n_tiles >= static_cast<int64_t>(b.grid) * (kBqPassBlock / 64) * 16 ? b.step1 : 1;
    while (step1 > 1 && n_tiles < static_cast<int64_t>(b.grid) * (kBqPassBlock / 64) * 8 * step1) step1 /= 2;
    return step1;
}

// First batched call on a handle: allocate the path's scratch (all or nothing).
int ensure_bq(mi355rec* h) {
    if (h->bq.ready) return MI355REC_OK;
    const int rc = ensure_bq_alloc(h);
    if (rc != MI355REC_OK) free_bq(h);   // no half-allocated state survives a failure
    return rc;
}

int ensure_bq_alloc(mi355rec* h) {
    auto& b = h->bq;
    // workgroups of a pass: what the 1024-query kernels can keep resident (LDS: 32 KiB of B
    // fragments per workgroup; registers: 4 resp. 5 waves per SIMD), the same for both passes
    int occ1 = 0, occ2 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, bq_pass_kernel<kBqMaxBlocks, false>, kBqPassBlock, 0) != hipSuccess || occ1 < 1) occ1 = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, bq_pass_kernel<kBqMaxBlocks, true>, kBqPassBlock, 0) != hipSuccess || occ2 < 1) occ2 = 1;
    b.occ1 = occ1 < 5 ? occ1 : 5;
    b.occ2 = occ2 < 5 ? occ2 : 5;
    int grid = h->cus * b.occ1;
    if (grid > kBqMaxPassGrid) grid = kBqMaxPassGrid;
    b.grid = grid;
    b.grid2 = h->cus * b.occ2;
    {
        int v = b.step1;
        MI355REC_EXP_INT(v, "MI355REC_BQ_STEP1", 1, 8);
        if (v == 1 || v == 2 || v == 4 || v == 8) b.step1 = v;
    }
    b.qgrid = h->cus < 1024 ? h->cus : 1024;   // (the queued scan's last workgroup merges up to 1024 lists per query)
    const int64_t tiles = (h->n + MultiConfig::kTileRows - 1) / MultiConfig::kTileRows;
    if (tiles < b.qgrid) b.qgrid = static_cast<int>(tiles);
    b.qiters = static_cast<int>((tiles + b.qgrid - 1) / b.qgrid);
    HIP_TRY(h, hipMalloc(&b.bfrag, sizeof(uint32_t) * kBqMaxBlocks * 64 * 4));
    HIP_TRY(h, hipMalloc(&b.qnorm, sizeof(float) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.qthr, sizeof(float) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.qflags, sizeof(uint32_t) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.cand_count, sizeof(int) * kBqMaxQueries * kBqCountStride));
    HIP_TRY(h, hipMemsetAsync(b.cand_count, 0, sizeof(int) * kBqMaxQueries * kBqCountStride, h->stream));
    // A query keeps about rows / 64 candidates at most (a power of two in [2048, 65536]): uniform rows need ~650 at 10 M,
    // rows that cluster a whole cluster's worth (profiles/r04_clustered.jsonl); past it the query goes to the exact queue.
    b.cand_cap = kBqCapMin;
    while (b.cand_cap < kBqCapMax && static_cast<int64_t>(b.cand_cap) * 64 < h->n) b.cand_cap *= 2;
    HIP_TRY(h, hipMalloc(&b.cand_rows, sizeof(uint64_t) * static_cast<size_t>(kBqMaxQueries) * b.cand_cap));
    HIP_TRY(h, hipMalloc(&b.cand_examined, sizeof(int) * kBqMaxQueries));
    HIP_TRY(h, hipMemsetAsync(b.cand_examined, 0, sizeof(int) * kBqMaxQueries, h->stream));
    // [0..3]: batched.hip.h; [4]: chunks that computed neighbourhood bounds, [7]: queries they won for (Batched::h_nb_report); [5]: queries ever queued; [6]: cand_cap; [8 .. 8 + 128): the queued scan's arrival counters, one per group of queries
    HIP_TRY(h, hipMalloc(&b.counters, sizeof(int) * (8 + 128)));
    HIP_TRY(h, hipMemsetAsync(b.counters, 0, sizeof(int) * (8 + 128), h->stream));
    HIP_TRY(h, hipMemcpyAsync(b.counters + 6, &b.cand_cap, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMalloc(&b.special_rows, sizeof(uint32_t) * kBqSpecialCap));
    HIP_TRY(h, hipMalloc(&b.nb_vals, sizeof(uint32_t) * kBqMaxQueries));
    {   // (optional: without it the bound is simply computed for every chunk)
        void* host = nullptr;
        void* dev = nullptr;
        if (hipHostMalloc(&host, sizeof(int) * 2, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&dev, host, 0) == hipSuccess) {
            b.h_nb_report = static_cast<volatile int*>(host);
            b.h_nb_report[0] = 0;
            b.h_nb_report[1] = 0;
            b.d_nb_report = static_cast<int*>(dev);
        } else {
            if (host) (void)hipHostFree(host);
            (void)hipGetLastError();
        }
    }
    HIP_TRY(h, hipMalloc(&b.gmax, sizeof(float) * static_cast<size_t>(grid) * kBqMaxBlocks * 64));
    // Room for the tile maxima of pass 1 (rows from the replica only).  Optional: without it pass 2 looks at every
    // (tile, query block) pair, as before.
    if (h->d_half) {
        const int64_t n_tiles = (h->n + 63) / 64;
        const int64_t visited = (n_tiles + bq_step1(h, n_tiles) - 1) / bq_step1(h, n_tiles);
        if (hipMalloc(&b.tile_max, sizeof(uint4) * static_cast<size_t>(visited) * (kBqMaxBlocks / 8) * 64) == hipSuccess) {
            b.tile_max_tiles = visited;
        } else {
            (void)hipGetLastError();
            b.tile_max = nullptr;
        }
    }
    HIP_TRY(h, hipMalloc(&b.queue, sizeof(int) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.qlists, sizeof(uint64_t) * static_cast<size_t>(kBqMaxQueries) * b.qgrid * kMultiMaxTopK));
    HIP_TRY(h, hipMalloc(&b.d_queries, sizeof(float) * kBqMaxQueries * kDim));
    HIP_TRY(h, hipMalloc(&b.d_exclude, sizeof(long long) * kBqMaxQueries));
    for (int i = 0; i < mi355rec::Batched::kSlots; ++i) {
        HIP_TRY(h, hipHostMalloc(&b.h_queries[i], sizeof(float) * kBqMaxQueries * kDim, hipHostMallocDefault));
        HIP_TRY(h, hipHostMalloc(&b.h_exclude[i], sizeof(long long) * kBqMaxQueries, hipHostMallocDefault));
        HIP_TRY(h, hipEventCreateWithFlags(&b.slot_ev[i], hipEventDisableTiming));
    }
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(bq_select_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(sizeof(float) * grid * 2 * 5 + sizeof(int) * (kBqSelectBlock / 64) * 256)));
    // the tighter bound is only claimed where fp16 subnormals are demonstrably kept
    hipLaunchKernelGGL(bq_selfcheck_kernel, dim3(1), dim3(64), 0, h->stream, b.qnorm);
    float chk[2] = {0.0f, 0.0f};
    HIP_TRY(h, hipMemcpyAsync(chk, b.qnorm, sizeof chk, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const bool kept = chk[0] == 9.5367431640625e-07f && chk[1] > 2.9e-6f && chk[1] < 3.1e-6f;
    b.margin = kept ? kBqMargin : kBqMarginFlush;
    b.ready = true;
    return MI355REC_OK;
}

void free_bq(mi355rec* h) {
    auto& b = h->bq;
    void* dev[] = {b.bfrag, b.qnorm, b.qthr, b.qflags, b.cand_count, b.cand_rows, b.cand_examined, b.counters, b.special_rows, b.nb_vals,
                   b.gmax, b.queue, b.qlists, b.d_queries, b.d_exclude, b.tile_max};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    if (b.h_nb_report) (void)hipHostFree(const_cast<int*>(b.h_nb_report));
    for (int i = 0; i < mi355rec::Batched::kSlots; ++i) {
        if (b.h_queries[i]) (void)hipHostFree(b.h_queries[i]);
        if (b.h_exclude[i]) (void)hipHostFree(b.h_exclude[i]);
        if (b.slot_ev[i]) (void)hipEventDestroy(b.slot_ev[i]);
    }
    b = mi355rec::Batched();
}

template <int NB, bool kFromReplica, bool kTileMax>
void launch_bq_passes(mi355rec* h, const float* d_queries, const long long* d_exclude, int count, int topn, hipStream_t s) {
    auto& b = h->bq;
    const int64_t n_tiles = (h->n + 63) / 64;   // a wave handles 64 rows (two 32-row MFMA tiles) at a time
    const int step1 = bq_step1(h, n_tiles);
    const size_t smem = sizeof(float) * b.grid * 2 * 5 + sizeof(int) * (kBqSelectBlock / 64) * 256;
    const uint2* half = reinterpret_cast<const uint2*>(h->d_half);
    // The queries are prepared by a launch of their own.  Folding it into pass 1's prologue (every workgroup builds
    // the fragments from the raw queries itself; bq_pass_kernel still can: prep_queries) was built and measured: the
    // launch it saves takes 4.4 us, the prologue it adds to each of pass 1's 1024 workgroups made pass 1 13 us slower
    // (10 M rows x 1024 queries: 102.5 instead of 89.7 us).
    // The same launch takes every query's NEIGHBOURHOOD bound (one workgroup each: batched.hip.h) when the queries exclude
    // rows — their own, for recommendByIndex — so that bq_select has it beside pass 1's group maxima.
    const int prep_blocks = (NB * 32 + 255) / 256;
    bool nbhd = h->n >= kNbhdRows;   // (queries without an excluded row here take the bound around their anchor: handoff.hip.h)
    if (nbhd && b.h_nb_report) {   // is it worth its thousand workgroups on this catalogue?  (Batched::h_nb_report)
        const int wins = b.h_nb_report[0], seen = b.h_nb_report[1];
        if (wins > 0) b.nb_sparse = false;
        else if (seen >= kBqNbProbeChunks) b.nb_sparse = true;
        if (b.nb_sparse && ++b.nb_skipped < kBqNbProbeEvery) nbhd = false;
        else b.nb_skipped = 0;
    }
    hipLaunchKernelGGL(bq_prepare_kernel, dim3(prep_blocks + (nbhd ? count : 0)), dim3(256), 0, s, d_queries, count, NB, b.bfrag,
                       b.qnorm, b.qflags, b.cand_count, b.counters, prep_blocks, h->d_feats, h->n, h->row_base, d_exclude, topn, b.nb_vals,
                       static_cast<const float*>(h->d_anchor));
    d_queries = nullptr;
    int slot = timing_begin(h, h->ev_pass, h->n_pass_pairs, h->pass_launches, s);
    hipLaunchKernelGGL((bq_pass_kernel<NB, false, 0, kFromReplica, kTileMax>), dim3(b.grid), dim3(kBqPassBlock), 0, s, h->d_feats, h->n,
                       n_tiles, step1, b.bfrag, b.gmax, b.cand_count, b.cand_rows, b.counters, b.special_rows, half,
                       b.tile_max, step1, static_cast<const float*>(b.qthr), static_cast<const uint32_t*>(b.qflags),
                       d_queries, count, b.qnorm, b.qflags);
    timing_end(h, h->ev_pass, h->n_pass_pairs, slot, s);
    hipLaunchKernelGGL(bq_select_kernel, dim3(NB * 8), dim3(kBqSelectBlock), smem, s, b.gmax, b.grid, NB, topn, b.margin, b.bfrag,
                       b.qflags, b.qthr, nbhd ? static_cast<const uint32_t*>(b.nb_vals) : static_cast<const uint32_t*>(nullptr), count, b.counters);
    b.nbhd_this_chunk = nbhd;
    int skip_step = step1;
    {   // experiment builds only: where does pass 2's time go (tools/bq_ab.sh)
        int v = 0;
        MI355REC_EXP_INT(v, "MI355REC_BQ_EXP", 1, 2);
        if (v == 1) skip_step = 1 << 30;   // no tile counts as visited: the new loop over ALL blocks of every tile
        if (v == 2 && kTileMax) {          // every visited tile skips ALL its blocks: what a tile costs without any
            static std::vector<float> inf(kBqMaxQueries, __builtin_inff());
            (void)hipMemcpyAsync(b.qthr, inf.data(), sizeof(float) * kBqMaxQueries, hipMemcpyHostToDevice, s);
        }
    }
    slot = timing_begin(h, h->ev_pass, h->n_pass_pairs, h->pass_launches, s);
    hipLaunchKernelGGL((bq_pass_kernel<NB, true, 0, kFromReplica, kTileMax>), dim3(b.grid2), dim3(kBqPassBlock), 0, s, h->d_feats, h->n,
                       n_tiles, 1, b.bfrag, b.gmax, b.cand_count, b.cand_rows, b.counters, b.special_rows, half,
                       b.tile_max, skip_step, static_cast<const float*>(b.qthr), static_cast<const uint32_t*>(b.qflags),
                       static_cast<const float*>(nullptr), 0, static_cast<float*>(nullptr), static_cast<uint32_t*>(nullptr));
    timing_end(h, h->ev_pass, h->n_pass_pairs, slot, s);
}

template <int NB>
void launch_bq_passes(mi355rec* h, const float* d_queries, const long long* d_exclude, int count, int topn, hipStream_t s) {
    // the passes read the fp16 replica when the handle has one (it holds their A operand ready-made)
    if (h->d_half && h->replica_mode != MI355REC_REPLICA_OFF) {
        // 512 queries and more: pass 1 also leaves the maxima of the tiles it looked at, pass 2 skips what they rule out
        if constexpr (NB >= 16) {
            const int64_t n_tiles = (h->n + 63) / 64;
            const int step1 = bq_step1(h, n_tiles);
            if (h->batch_path != MI355REC_BATCH_MFMA_NOSKIP && h->bq.tile_max && (n_tiles + step1 - 1) / step1 <= h->bq.tile_max_tiles) {
                launch_bq_passes<NB, true, true>(h, d_queries, d_exclude, count, topn, s);
                return;
            }
        }
        launch_bq_passes<NB, true, false>(h, d_queries, d_exclude, count, topn, s);
    } else {
        launch_bq_passes<NB, false, false>(h, d_queries, d_exclude, count, topn, s);
    }
}

// One chunk of up to kBqMaxQueries queries that are already in device memory.
int enqueue_bq_chunk(mi355rec* h, const float* d_queries, const long long* d_exclude, int count, int topn,
                     uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    auto& b = h->bq;
    const int blocks = (count + 31) / 32;
    int nb = 1;
    while (nb < blocks) nb *= 2;
    switch (nb) {
        case 1: launch_bq_passes<1>(h, d_queries, d_exclude, count, topn, s); break;
        case 2: launch_bq_passes<2>(h, d_queries, d_exclude, count, topn, s); break;
        case 4: launch_bq_passes<4>(h, d_queries, d_exclude, count, topn, s); break;
        case 8: launch_bq_passes<8>(h, d_queries, d_exclude, count, topn, s); break;
        case 16: launch_bq_passes<16>(h, d_queries, d_exclude, count, topn, s); break;
        default: launch_bq_passes<32>(h, d_queries, d_exclude, count, topn, s); break;
    }
    hipLaunchKernelGGL(bq_finalize_kernel, dim3(count), dim3(kBqFinalBlock), 0, s, h->d_feats, h->row_base, d_queries,
                       d_exclude, count, topn, b.qflags, b.cand_count, b.cand_rows, b.cand_cap, b.counters, b.special_rows, b.queue,
                       out_keys, out_idx, out_score,
                       b.nbhd_this_chunk ? static_cast<const uint32_t*>(b.nb_vals) : static_cast<const uint32_t*>(nullptr),
                       b.cand_examined, b.nbhd_this_chunk ? b.d_nb_report : static_cast<int*>(nullptr));
    // The exact multi-query scan for whatever the bound could not be claimed for, its merge included (usually
    // nothing: the launch exits at once on an empty queue).
    hipLaunchKernelGGL((scan_multi_queued_kernel<MultiConfig>), dim3(b.qgrid), dim3(MultiConfig::kBlock), 0, s,
                       h->d_feats, h->n, b.qiters, h->row_base, d_queries, d_exclude, b.queue, b.counters + 1, topn,
                       b.qlists, reinterpret_cast<unsigned*>(b.counters + 8), out_keys, out_idx, out_score,
                       static_cast<int64_t>(topn));
    HIP_TRY(h, hipGetLastError());
    ++b.launches;
    ++h->routes.mfma_two_pass;
    b.last_count = count;
    return MI355REC_OK;
}

// Host queries: through a pinned staging slot into the handle's device buffers.
int stage_queries(mi355rec* h, const float* queries, const int64_t* exclude, int count, hipStream_t s) {
    auto& b = h->bq;
    const int slot = b.next_slot;
    b.next_slot = (slot + 1) % mi355rec::Batched::kSlots;
    if (b.slot_used[slot]) HIP_TRY(h, hipEventSynchronize(b.slot_ev[slot]));  // its previous copy has long finished
    std::memcpy(b.h_queries[slot], queries, sizeof(float) * static_cast<size_t>(count) * kDim);
    for (int i = 0; i < count; ++i) b.h_exclude[slot][i] = exclude ? static_cast<long long>(exclude[i]) : -1ll;
    HIP_TRY(h, hipMemcpyAsync(b.d_queries, b.h_queries[slot], sizeof(float) * static_cast<size_t>(count) * kDim,
                              hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(b.d_exclude, b.h_exclude[slot], sizeof(long long) * static_cast<size_t>(count),
                              hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipEventRecord(b.slot_ev[slot], s));
    b.slot_used[slot] = true;
    return MI355REC_OK;
}

bool use_bq(const mi355rec* h, int batch, int topn) {
    if (topn > kMultiMaxTopK || h->n < 1) return false;
    if (h->batch_path == MI355REC_BATCH_MULTI || h->batch_path == MI355REC_BATCH_HALF || h->batch_path == MI355REC_BATCH_Q8) return false;
    if (h->batch_path == MI355REC_BATCH_MFMA || h->batch_path == MI355REC_BATCH_MFMA_NOSKIP) return true;
    const bool replica = h->d_half && h->replica_mode != MI355REC_REPLICA_OFF;
    return batch >= (replica ? kBqMinBatchReplica : kBqMinBatch) && h->n >= kBqMinRows;
}

int enqueue_bq_host(mi355rec* h, const float* queries, const int64_t* exclude, int batch, int topn,
                    uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    int rc = ensure_bq(h);
    if (rc) return rc;
    for (int b0 = 0; b0 < batch; b0 += kBqMaxQueries) {
        const int count = batch - b0 < kBqMaxQueries ? batch - b0 : kBqMaxQueries;
        rc = stage_queries(h, queries + static_cast<size_t>(b0) * kDim, exclude ? exclude + b0 : nullptr, count, s);
        if (rc) return rc;
        const size_t off = static_cast<size_t>(b0) * topn;
        rc = enqueue_bq_chunk(h, h->bq.d_queries, h->bq.d_exclude, count, topn, out_keys + off,
                              out_idx ? out_idx + off : nullptr, out_score ? out_score + off : nullptr, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// `batch` queries on stream `s`: multi-query passes where they apply (topn <=
// kMultiMaxTopK), otherwise one scan per query.  Outputs are batch x topn.
int enqueue_batch(mi355rec* h, const float* queries, const int64_t* exclude_global, int batch, int topn,
                  uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    // 2 ... kHmAutoMax queries on a shard with a replica: multi-query passes over the replica (24 B/row,
    // one pass per 12 queries); more: the matrix-core path (two passes whatever the count up to 1024)
    const bool half_multi = half_multi_ok(h, topn) && h->n >= kBqMinRows &&
                            (h->batch_path == MI355REC_BATCH_HALF || h->batch_path == MI355REC_BATCH_Q8 ||
                             (h->batch_path == MI355REC_BATCH_AUTO && batch >= 2 && batch <= kHmAutoMax));
    if (half_multi) {
        for (int b = 0; b < batch; b += kMultiChain) {
            const int count = batch - b < kMultiChain ? batch - b : kMultiChain;
            const size_t off = static_cast<size_t>(b) * topn;
            const int rc = enqueue_half_multi(h, queries + static_cast<size_t>(b) * kDim, nullptr,
                                              exclude_global ? exclude_global + b : nullptr, count, topn, out_keys + off,
                                              out_idx ? out_idx + off : nullptr, out_score ? out_score + off : nullptr, s);
            if (rc) return rc;
        }
        return MI355REC_OK;
    }
    if (use_bq(h, batch, topn))
        return enqueue_bq_host(h, queries, exclude_global, batch, topn, out_keys, out_idx, out_score, s);
    if (batch > 1 && topn <= kMultiMaxTopK && h->n > 0) {
        for (int b = 0; b < batch; b += kMultiChain) {
            const int count = batch - b < kMultiChain ? batch - b : kMultiChain;
            const size_t off = static_cast<size_t>(b) * topn;
            const int rc = enqueue_multi(h, queries + static_cast<size_t>(b) * kDim,
                                         exclude_global ? exclude_global + b : nullptr, count, topn, out_keys + off,
                                         out_idx ? out_idx + off : nullptr, out_score ? out_score + off : nullptr, s);
            if (rc) return rc;
        }
        return MI355REC_OK;
    }
    for (int b = 0; b < batch; ++b) {
        const size_t off = static_cast<size_t>(b) * topn;
        const int rc = enqueue_query(h, nullptr, queries + static_cast<size_t>(b) * kDim,
                                     exclude_global ? exclude_global[b] : -1, topn, out_keys + off,
                                     out_idx ? out_idx + off : nullptr, out_score ? out_score + off : nullptr, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

}  // namespace
