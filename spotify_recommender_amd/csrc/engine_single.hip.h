// engine_single.hip.h — ONE query at a time: which rows its scan streams (fp32, 8-bit replica), the sample / neighbourhood
// launch in front of a query alone, the scan + merge (replaces calculateSimilarities + the host heap, Recommender.cu:184-254,
// 293-315), rounds for topn > 1024, the completion word of a synchronous query, and the STREAM of single queries that runs
// one call behind so that every launch carries the next query's seed riders.  (Part of mi355rec.hip's translation unit.)
#pragma once

#include "engine_state.hip.h"

namespace {

int flush_mstream(mi355rec* h, hipStream_t s);   // (engine_batch.hip.h: a stream of BATCHES on the handle is closed before a single query joins)

bool use_half(const mi355rec* h, const uint64_t* upper_dev) {
    if (!h->d_half || upper_dev || h->replica_mode == MI355REC_REPLICA_OFF) return false;
    return h->replica_mode == MI355REC_REPLICA_ON || h->replica_mode == MI355REC_REPLICA_FP16 || h->n >= kHalfAutoMinRows;
}

// Single queries stream the 8-bit replica (half the fp16 one's bytes per row); experiment builds can keep them on the
// fp16 one (MI355REC_REPLICA_FP16: A/B).
bool use_q8(const mi355rec* h) { return h->d_q8 && h->replica_mode != MI355REC_REPLICA_FP16; }

// Which rows the next single query on this handle streams.
int single_kind(const mi355rec* h, const uint64_t* upper_dev) {
    if (!use_half(h, upper_dev)) return kFp32;
    return use_q8(h) ? kQ8 : kFp16;
}

// Streamed launches over the 8-bit replica: the last seed rider out turns the sample into the next launch's
// cutoff (saves a ~4 us select in every workgroup of that launch).  The riders then take sample + select
// (~10 us) in all, so only where the scanners run longer than that.
bool q8_hoists(const mi355rec* h) { return h->qg.riders > 0 && h->qg.r_iters >= 5; }
// The sample holds EXACT scores of its rows (one margin in the cutoff instead of two: a third of the candidates)
// where the extra fetch per sampled wave is not on the launch's critical path.
bool q8_exact_sample(const mi355rec* h) { return h->qg.iters >= 3; }

// Does a neighbourhood give the scan a bound (handoff.hip.h)?  Around the row the query excludes when that is a row of THIS
// shard, else (a query by value, a row of another shard) around the query's anchor — on every shard large enough to have one.
bool nbhd_applies(const mi355rec* h, int64_t exclude_global) {
    (void)exclude_global;
    return h->n >= kNbhdRows;
}

// The sample launch of a query ALONE over a replica (the first query of a stream as well): the sampled regions and,
// when the excluded row is a row of this shard, one more workgroup for its neighbourhood.  The values are tagged with
// `epoch`, which the scan that reads them is given as well.
void enqueue_half_seed(mi355rec* h, int kind, const float* qptr, const QueryArg& qa, int64_t exclude_global, int topn,
                       unsigned long long* seed_buf, uint32_t epoch, hipStream_t s) {
    if (kind == kQ8) {
        const int extra = nbhd_applies(h, exclude_global) ? 1 : 0;
        if (h->qg.seed_grid + extra <= 0) return;
#define SEED_Q8(EXACT)                                                                                                     \
    hipLaunchKernelGGL((seed_q8_kernel<EXACT>), dim3(h->qg.seed_grid + extra), dim3(kHalfSeedBlock), 0, s, h->d_feats, h->d_q8, \
                       h->n, h->qg.seed_stride, h->row_base, qa, qptr, exclude_global, seed_buf, epoch, h->qg.seed_grid, topn, \
                       static_cast<const float*>(h->d_anchor))
        if (q8_exact_sample(h)) SEED_Q8(true);
        else SEED_Q8(false);
#undef SEED_Q8
        return;
    }
#ifdef MI355REC_EXPERIMENTS
    if (h->hg.seed_grid <= 0) return;
    uint32_t* const seed_out = reinterpret_cast<uint32_t*>(seed_buf);   // the fp16 scan's plain values
    if (qptr) {
        hipLaunchKernelGGL((seed_half_kernel<true>), dim3(h->hg.seed_grid), dim3(kHalfSeedBlock), 0, s, h->d_feats, h->d_half,
                           h->n, h->hg.seed_stride, h->row_base, qa, qptr, exclude_global, seed_out);
    } else {
        hipLaunchKernelGGL((seed_half_kernel<false>), dim3(h->hg.seed_grid), dim3(kHalfSeedBlock), 0, s, h->d_feats, h->d_half,
                           h->n, h->hg.seed_stride, h->row_base, qa, kNoQueryPtr, exclude_global, seed_out);
    }
#endif
}

// The same for a query alone over the fp32 rows (kernels.hip.h, seed_f32_kernel): the regions' last workgroup leaves the
// bound in `ctl`, the neighbourhood workgroup its own in seed_buf[kNbhdSlot].  `*ctl_done` is what ctl->done holds (the
// counter is never reset).  Returns whether a sample (hence a bound in `ctl`) was enqueued.
bool enqueue_f32_seed(mi355rec* h, const float* qptr, const float* query12, int64_t exclude_global, int topn,
                      unsigned long long* seed_buf, SeedCtl* ctl, unsigned* ctl_done, uint32_t epoch, hipStream_t s) {
    NextSeed sd;
    std::memset(&sd, 0, sizeof sd);
    sd.anchors = h->d_anchor;
    sd.query_ptr = qptr;
    if (!qptr) std::memcpy(sd.q, query12, sizeof sd.q);
    sd.exclude_global = exclude_global;
    sd.out = seed_buf;
    sd.regions = h->fg.seed_grid;
    sd.n_wgs = h->fg.seed_grid;
    sd.stride_rows = h->fg.seed_stride;
    sd.ctl = sd.regions > 0 ? ctl : nullptr;
    sd.topk = topn;
    sd.epoch = epoch;
    sd.done_base = *ctl_done + (h->dbg_no_last ? 0x40000000u : 0u);
    sd.debug_skip = h->dbg_skip_regions;
    sd.nbhd = nbhd_applies(h, exclude_global) ? 1 : 0;
    if (sd.regions + sd.nbhd <= 0) return false;
    hipLaunchKernelGGL(seed_f32_kernel, dim3(sd.regions + sd.nbhd), dim3(kHalfSeedBlock), 0, s, h->d_feats, h->n, h->row_base, sd);
    if (sd.regions > 0) {
        *ctl_done += static_cast<unsigned>(sd.regions);
        h->dbg_no_last = false;   // (test hooks of mi355rec_debug_handoff: they apply to ONE sampling launch)
        h->dbg_skip_regions = 0;
    }
    return sd.regions > 0;
}

// Enqueue the scan for one query.  qptr != null: the kernel reads the query's 12 floats from there
// (a resident row, or any other device-readable address).
// *n_lists = per-workgroup lists it leaves in d_block_lists.
// lone != null (a lone query whose caller waits on the host): over the 8-bit replica of a large shard the launch
// also merges its own lists into lone's buffers (merge.hip.h, lone_tail) and *fused is set.
constexpr int64_t kLoneFusedMinRows = 4000000;
// A query alone over the fp32 rows gets a sample launch of its own (~5 us) from here up: below, the scan is a dozen
// microseconds and launch-bound.
constexpr int64_t kF32LoneSeedMinRows = 4000000;
// (Round 4 had a LONE synchronous query below 1.5 M rows read the fp32 rows — two launches against the replica's three
// were worth more than the bytes: 27.3 against 29.7 us at 1 M rows.  Once the 8-bit scan's prologue had been fixed —
// sample requested before the first tile, one LDS atomic per wave in its selection — the replica won from 1 M rows up
// again (tools/route_thresholds.sh: 25.7 against 27.4 us at 1 M, 27.1 against 30.4 at 1.4 M, 28.3 against 39.9 at 3 M;
// 28.0 against 24.3 at 0.7 M), which is where single queries take it anyway: the rule is gone.)
int enqueue_scan(mi355rec* h, const float* qptr, const float* query12,
                 int64_t exclude_global, int topn, const uint64_t* upper_dev, hipStream_t s, int* n_lists,
                 const LoneTail* lone = nullptr, bool* fused = nullptr) {
    QueryArg qa;
    std::memset(&qa, 0, sizeof qa);
    qa.margin = h->margin_mix;
    if (!qptr) std::memcpy(qa.q, query12, sizeof qa.q);
    const PrevMerge none{nullptr, 0, 0, nullptr};
    const LoneTail no_tail{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};
    NextSeed no_next;
    std::memset(&no_next, 0, sizeof no_next);
    if (fused) *fused = false;
    const int kind = single_kind(h, upper_dev);
    if (kind == kQ8) {
        ++h->half_scans;
        *n_lists = h->qg.grid;
        ++h->q8_scans;
        const uint32_t epoch = next_epoch(h);
        enqueue_half_seed(h, kQ8, qptr, qa, exclude_global, topn, h->d_half_seed, epoch, s);
        const int q8_seeds = (q8_exact_sample(h) ? -1 : 1) * h->qg.seed_grid * kHalfSeedWaves;   // (negative: exact values)
        const unsigned long long* const no_cutoff = nullptr;
        if (lone && h->n >= kLoneFusedMinRows) {
            ++h->routes.q8_lone;
            // the arrival counters of the launch's tail count up and are never reset: this launch starts from ...
            LoneTail tail = *lone;
            const unsigned grid = static_cast<unsigned>(h->qg.grid);
            for (unsigned g = 0; g < 8u; ++g) tail.base[g] = h->lone_base[g];
            tail.base[8] = h->lone_base[8];
            if (qptr) {
                LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, true, false, true>),
                             dim3(h->qg.grid), dim3(Q8Config::kBlock), s,
                             h->d_feats, h->d_q8, h->n, h->qg.iters, h->row_base, qa, qptr, exclude_global, topn,
                             h->d_block_lists, h->d_half_seed, q8_seeds, h->d_half_rescored, none, no_next,
                             no_cutoff, tail, epoch);
            } else {
                LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, false, false, true>),
                             dim3(h->qg.grid), dim3(Q8Config::kBlock), s,
                             h->d_feats, h->d_q8, h->n, h->qg.iters, h->row_base, qa, kNoQueryPtr, exclude_global, topn,
                             h->d_block_lists, h->d_half_seed, q8_seeds, h->d_half_rescored, none, no_next,
                             no_cutoff, tail, epoch);
            }
            HIP_TRY(h, hipGetLastError());
            // (the books move only once the launch is known to have been accepted: a refused launch leaves host and
            // device counters in step)
            for (unsigned g = 0; g < 8u; ++g) h->lone_base[g] += lone_tail_members(grid, g);
            h->lone_base[8] += lone_tail_groups(grid);
            *fused = true;
            return MI355REC_OK;
        }
        ++h->routes.q8;
        if (qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, true, false>),
                         dim3(h->qg.grid), dim3(Q8Config::kBlock), s,
                         h->d_feats, h->d_q8, h->n, h->qg.iters, h->row_base, qa, qptr, exclude_global, topn,
                         h->d_block_lists, h->d_half_seed, q8_seeds, h->d_half_rescored, none, no_next,
                         no_cutoff, no_tail, epoch);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, false, false>),
                         dim3(h->qg.grid), dim3(Q8Config::kBlock), s,
                         h->d_feats, h->d_q8, h->n, h->qg.iters, h->row_base, qa, kNoQueryPtr, exclude_global, topn,
                         h->d_block_lists, h->d_half_seed, q8_seeds, h->d_half_rescored, none, no_next,
                         no_cutoff, no_tail, epoch);
        }
        HIP_TRY(h, hipGetLastError());
        return MI355REC_OK;
    }
#ifdef MI355REC_EXPERIMENTS
    if (kind == kFp16) {
        ++h->half_scans;
        *n_lists = h->hg.grid;
        ++h->routes.fp16;
        uint32_t* const half_seed = reinterpret_cast<uint32_t*>(h->d_half_seed);   // (the fp16 scan's plain sample values)
        enqueue_half_seed(h, kFp16, qptr, qa, exclude_global, topn, h->d_half_seed, 0u, s);
        if (qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_kernel<HalfConfig, true, false>),
                         dim3(h->hg.grid), dim3(HalfConfig::kBlock), s,
                         h->d_feats, h->d_half, h->n, h->hg.iters, h->row_base, qa, qptr, exclude_global, topn,
                         h->d_block_lists, half_seed, h->hg.seed_grid * kHalfSeedWaves, h->d_half_rescored, none, no_next);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_kernel<HalfConfig, false, false>),
                         dim3(h->hg.grid), dim3(HalfConfig::kBlock), s,
                         h->d_feats, h->d_half, h->n, h->hg.iters, h->row_base, qa, kNoQueryPtr, exclude_global,
                         topn, h->d_block_lists, half_seed, h->hg.seed_grid * kHalfSeedWaves, h->d_half_rescored, none, no_next);
        }
        HIP_TRY(h, hipGetLastError());
        return MI355REC_OK;
    }
#endif
    *n_lists = h->grid;
    ++h->routes.fp32;
    // The launch-wide bound (kernels.hip.h): on shards where ~5 us are worth it, and never for the later rounds of
    // topn > 1024 (they look for keys BELOW the round before: a lower bound on the best keys says nothing there).
    const unsigned long long* bound = nullptr;
    const unsigned long long* sample = nullptr;
    uint32_t epoch = 0u;
    if (!upper_dev && h->n >= kF32LoneSeedMinRows) {
        epoch = next_epoch(h);
        if (enqueue_f32_seed(h, qptr, query12, exclude_global, topn, h->d_half_seed, h->d_lone_ctl, &h->lone_ctl_done, epoch, s))
            bound = &h->d_lone_ctl->cutoff;
        sample = h->d_half_seed;
    }
    if (qptr) {
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_kernel<ScanConfig, true, false>),
                     dim3(h->grid), dim3(kScanBlock), s,
                     h->d_feats, h->n, h->rows_per_block, h->iters, h->row_base, qa,
                     qptr, exclude_global, topn, h->d_block_lists,
                     static_cast<float*>(nullptr), upper_dev, none, bound, sample, epoch, no_next);
    } else {
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_kernel<ScanConfig, false, false>),
                     dim3(h->grid), dim3(kScanBlock), s,
                     h->d_feats, h->n, h->rows_per_block, h->iters, h->row_base, qa,
                     kNoQueryPtr, exclude_global, topn, h->d_block_lists,
                     static_cast<float*>(nullptr), upper_dev, none, bound, sample, epoch, no_next);
    }
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

int enqueue_merge(mi355rec* h, const uint64_t* lists, int n_lists, int list_len, int topn,
                  uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s, uint32_t notify = 0) {
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    if (notify) {   // the host polls h->h_done for this value (mi355rec_query_row_topn)
        hipLaunchKernelGGL(merge_notify_kernel, dim3(1), dim3(kMergeBlock), 0, s, lists, n_lists, list_len,
                           static_cast<int64_t>(list_len), topn, out_keys, out_idx, out_score, h->hd_done, notify);
    } else {
        hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(kMergeBlock), 0, s, lists, n_lists, list_len,
                           static_cast<int64_t>(list_len), static_cast<int64_t>(0), topn, out_keys, out_idx, out_score,
                           static_cast<int64_t>(0));
    }
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

int check_topn(mi355rec* h, int topn, bool allow_rounds) {
    if (topn <= 0)
        return fail(h, MI355REC_ERR_INVALID_ARG, "topn must be positive, got %d", topn);
    if (!allow_rounds && topn > kMaxTopK)
        return fail(h, MI355REC_ERR_INVALID_ARG, "topn %d > %d is not supported by this call", topn, kMaxTopK);
    return MI355REC_OK;
}

// One query end to end on stream `s`: scan + merge, in rounds of kMaxTopK when
// topn is larger (round r only sees keys below the last key of round r-1, read
// from device memory, so the rounds are enqueued back to back without a sync).
int enqueue_query(mi355rec* h, const float* qptr, const float* query12, int64_t exclude_global,
                  int topn_asked, uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s, uint32_t notify = 0) {
    // A shard of n rows has at most n results (the reference's heap never grows
    // past N-1, Recommender.cu:300): run only the rounds that can produce keys and
    // pad the rest, so an absurd topn costs a memset, not topn/1024 catalogue scans.
    const int topn = static_cast<int64_t>(topn_asked) < h->n ? topn_asked : static_cast<int>(h->n);
    if (topn < topn_asked) {
        const size_t pad = static_cast<size_t>(topn_asked - topn);
        HIP_TRY(h, hipMemsetAsync(out_keys + topn, 0, pad * sizeof(uint64_t), s));
        if (out_idx) HIP_TRY(h, hipMemsetAsync(out_idx + topn, 0xff, pad * sizeof(int64_t), s));
        if (out_score) HIP_TRY(h, hipMemsetAsync(out_score + topn, 0, pad * sizeof(float), s));
    }
    for (int done = 0; done < topn; done += kMaxTopK) {
        const int k = topn - done < kMaxTopK ? topn - done : kMaxTopK;
        const uint64_t* upper = done ? out_keys + done - 1 : nullptr;
        int lists = 0;
        // a notifying query (single round, its caller polls the completion word): scan, merge and the word in ONE launch
        LoneTail lone{h->d_lone_ctr, out_keys, out_idx, out_score, h->hd_done, notify, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};   // (bases: enqueue_scan)
        bool fused = false;
        int rc = enqueue_scan(h, qptr, query12, exclude_global, k, upper, s, &lists, (notify && h->d_lone_ctr) ? &lone : nullptr, &fused);
        if (rc) return rc;
        if (fused) {
            ++h->lone_fused;
            continue;
        }
        // (a notifying merge is only asked for single-round queries: it is the last launch of the call)
        rc = enqueue_merge(h, h->d_block_lists, lists, k, k, out_keys + done,
                           out_idx ? out_idx + done : nullptr, out_score ? out_score + done : nullptr, s, notify);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// Waits for the completion word of a notifying merge (a relaxed spin on pinned host memory); the stream
// is asked now and then so that a failed launch cannot hang the caller.
int wait_done(mi355rec* h, uint32_t want) {
    for (uint64_t spins = 1;; ++spins) {
        if (__atomic_load_n(h->h_done, __ATOMIC_ACQUIRE) == want) return MI355REC_OK;
        __builtin_ia32_pause();
        if ((spins & 0x3ffff) == 0) {   // every ~1 ms
            const hipError_t e = hipStreamQuery(h->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n(h->h_done, __ATOMIC_ACQUIRE) == want) return MI355REC_OK;
                return fail(h, MI355REC_ERR_HIP, "the query's stream drained without its completion word");
            }
            if (e != hipErrorNotReady) return fail(h, MI355REC_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
        }
    }
}

// ---- streamed single queries -------------------------------------------------------
// A stream of single queries runs ONE CALL BEHIND: query k is launched by call k + 1 (or by the flush), and its launch
// carries, beside the scanners, the merger of query k - 1's lists (one workgroup) and — where the launch can spare
// them — the seed riders and the neighbourhood workgroup of query k + 1 (handoff.hip.h), so that every launch starts
// from a launch-wide bound without a sample launch of its own.  That holds for all three kinds of rows a scan can
// stream (fp32, 8-bit replica; fp16 replica in experiment builds).  One scanning workgroup fewer than the plain scan
// uses per non-scanning one, so the launch still fits the chip in one wave of workgroups.
int ensure_streamed_alloc(mi355rec* h);
int ensure_streamed(mi355rec* h) {
    if (h->streamed_ready) return MI355REC_OK;
    const int rc = ensure_streamed_alloc(h);
    if (rc != MI355REC_OK) {   // all or nothing: no half-allocated state survives a failure
        for (int i = 0; i < 2; ++i) {
            if (h->d_stream_lists[i]) (void)hipFree(h->d_stream_lists[i]);
            h->d_stream_lists[i] = nullptr;
        }
        return rc;
    }
    h->streamed_ready = true;
    return MI355REC_OK;
}

int ensure_streamed_alloc(mi355rec* h) {
    int most = h->sgrid > h->hg.sgrid ? h->sgrid : h->hg.sgrid;
    if (h->qg.sgrid > most) most = h->qg.sgrid;
    for (int i = 0; i < 2; ++i)
        HIP_TRY(h, hipMalloc(&h->d_stream_lists[i], sizeof(uint64_t) * static_cast<size_t>(most) * kMaxTopK));
    return MI355REC_OK;
}

int launch_stashed(mi355rec* h, hipStream_t s, bool with_next, const float* next_ptr, const float* next_q,
                   int64_t next_exclude, int next_topn, int next_buf, uint32_t next_epoch_tag);

int flush_streamed(mi355rec* h, hipStream_t s) {
    if (h->stashed.has) {
        const int rc = launch_stashed(h, s, false, nullptr, nullptr, -1, 0, 0, 0u);
        if (rc) return rc;
    }
    if (!h->pending) return MI355REC_OK;
    h->pending = false;
    return enqueue_merge(h, h->d_stream_lists[h->pending_buf], h->pending_lists, h->pending_topn, h->pending_topn,
                         h->pending_out, nullptr, nullptr, s);
}

// How many seed riders a streamed launch over `kind` rows carries for the NEXT query, and whether their last one
// leaves that query's bound (cutoff) in d_stream_ctl.
int stream_riders(const mi355rec* h, int kind) { return kind == kFp32 ? h->fg.riders : (kind == kQ8 ? h->qg.riders : h->hg.riders); }
bool stream_hoists(const mi355rec* h, int kind) {
    return kind == kFp32 ? h->fg.riders > 0 : (kind == kQ8 ? q8_hoists(h) : false);
}
// ... and whether the launch has a workgroup for the next query's neighbourhood at all.
bool stream_nbhd(const mi355rec* h, int kind) { return kind == kFp32 ? h->fg.nbhd != 0 : (kind == kQ8 ? h->qg.riders > 0 : false); }

// Launches the stashed streamed query: scanners + the riding merger of the query before it + (with_next) the seed
// riders and the neighbourhood workgroup of the query after it.
int launch_stashed(mi355rec* h, hipStream_t s, bool with_next, const float* next_ptr, const float* next_q,
                   int64_t next_exclude, int next_topn, int next_buf, uint32_t next_epoch_tag) {
    auto& st = h->stashed;
    // The fp32 scan's riding merger keeps 2048 survivors; with ~770 lists and topN near 1000 about
    // 2.2 topN keys survive its first cut, and an overflow drops into the exact radix select over all
    // keys in global memory (correct, ~1 ms).  Such a query's merge gets its own launch instead.
    if (st.kind == kFp32 && h->pending && h->pending_topn > kRideTopnMax) {
        h->pending = false;
        const int rc = enqueue_merge(h, h->d_stream_lists[h->pending_buf], h->pending_lists, h->pending_topn, h->pending_topn,
                                     h->pending_out, nullptr, nullptr, s);
        if (rc) return rc;
    }
    const int buf = h->pending ? 1 - h->pending_buf : 0;
    PrevMerge prev{nullptr, 0, 0, nullptr};
    if (h->pending) prev = PrevMerge{h->d_stream_lists[h->pending_buf], h->pending_lists, h->pending_topn, h->pending_out};
    NextSeed next;
    std::memset(&next, 0, sizeof next);
    next.anchors = h->d_anchor;
    next.query_ptr = nullptr;
    next.exclude_global = -1;
    int scanners, iters;
    if (st.kind == kFp32) {
        scanners = h->sgrid;
        iters = h->siters;
    } else {
        const ReplicaGeom& g = st.kind == kQ8 ? h->qg : h->hg;
        scanners = g.sgrid;
        iters = g.siters;
    }
    unsigned riders_arriving = 0u;
    if (with_next && (stream_riders(h, st.kind) > 0 || stream_nbhd(h, st.kind))) {
        next.query_ptr = next_ptr;
        if (!next_ptr) std::memcpy(next.q, next_q, sizeof next.q);
        next.exclude_global = next_exclude;
        next.out = h->d_stream_seed[next_buf];
        next.n_wgs = stream_riders(h, st.kind);
        next.nbhd = stream_nbhd(h, st.kind) ? 1 : 0;   // (it stores its slot even when the excluded row is not of this shard)
        if (st.kind == kFp32) {
            next.regions = h->fg.seed_grid;
            next.stride_rows = h->fg.seed_stride;
            scanners = h->fg.r_scan;
            iters = h->fg.r_iters;
        } else {
            const ReplicaGeom& g = st.kind == kQ8 ? h->qg : h->hg;
            next.regions = g.seed_grid;
            next.stride_rows = g.seed_stride;
            scanners = g.r_scan;
            iters = g.r_iters;
        }
        next.ctl = (next.n_wgs > 0 && stream_hoists(h, st.kind)) ? h->d_stream_ctl + next_buf : nullptr;
        next.topk = next_topn;
        next.exact = st.kind == kQ8 && q8_exact_sample(h);
        next.epoch = next_epoch_tag;
        if (next.ctl) {   // the riders' arrival counter counts up and is never reset: this launch's riders start from ...
            next.done_base = h->ctl_done[next_buf] + (h->dbg_no_last ? 0x40000000u : 0u);
            riders_arriving = static_cast<unsigned>(next.n_wgs);
        }
        next.debug_skip = h->dbg_skip_regions;
        h->dbg_no_last = false;
        h->dbg_skip_regions = 0;
    }
    QueryArg qa;
    std::memset(&qa, 0, sizeof qa);
    qa.margin = h->margin_mix;
    if (!st.qptr) std::memcpy(qa.q, st.q, sizeof qa.q);
    const dim3 grid(static_cast<unsigned>(scanners + 1 + next.n_wgs + next.nbhd));
    unsigned long long* const my_seed = h->d_stream_seed[st.seed_buf];
    const unsigned long long* ready = st.cutoff_ready ? &h->d_stream_ctl[st.seed_buf].cutoff : nullptr;
    if (st.kind == kQ8) {
        ++h->half_scans;
        ++h->q8_scans;
        ++h->routes.q8;
        const int n_seed = h->qg.seed_grid * kHalfSeedWaves;
        const int q8_seeds = q8_exact_sample(h) ? -n_seed : n_seed;   // (negative: exact values)
        const LoneTail no_tail{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};
        if (st.qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, true, true>),
                         grid, dim3(Q8Config::kBlock), s,
                         h->d_feats, h->d_q8, h->n, iters, h->row_base, qa, st.qptr, st.exclude, st.topn,
                         h->d_stream_lists[buf], my_seed, q8_seeds, h->d_half_rescored, prev, next, ready, no_tail, st.epoch);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, false, true>),
                         grid, dim3(Q8Config::kBlock), s,
                         h->d_feats, h->d_q8, h->n, iters, h->row_base, qa, kNoQueryPtr, st.exclude, st.topn,
                         h->d_stream_lists[buf], my_seed, q8_seeds, h->d_half_rescored, prev, next, ready, no_tail, st.epoch);
        }
#ifdef MI355REC_EXPERIMENTS
    } else if (st.kind == kFp16) {
        ++h->half_scans;
        ++h->routes.fp16;
        const int n_seed = h->hg.seed_grid * kHalfSeedWaves;
        if (st.qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_kernel<HalfConfig, true, true>),
                         grid, dim3(HalfConfig::kBlock), s,
                         h->d_feats, h->d_half, h->n, iters, h->row_base, qa, st.qptr, st.exclude, st.topn,
                         h->d_stream_lists[buf], reinterpret_cast<uint32_t*>(my_seed), n_seed, h->d_half_rescored, prev, next);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_kernel<HalfConfig, false, true>),
                         grid, dim3(HalfConfig::kBlock), s,
                         h->d_feats, h->d_half, h->n, iters, h->row_base, qa, kNoQueryPtr, st.exclude, st.topn,
                         h->d_stream_lists[buf], reinterpret_cast<uint32_t*>(my_seed), n_seed, h->d_half_rescored, prev, next);
        }
#endif
    } else {
        ++h->routes.fp32;
        if (st.qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_kernel<ScanConfig, true, false, 0, true>),
                         grid, dim3(kScanBlock), s,
                         h->d_feats, h->n, static_cast<int64_t>(0), iters, h->row_base, qa, st.qptr,
                         st.exclude, st.topn, h->d_stream_lists[buf], static_cast<float*>(nullptr),
                         static_cast<const uint64_t*>(nullptr), prev, ready, my_seed, st.epoch, next);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_kernel<ScanConfig, false, false, 0, true>),
                         grid, dim3(kScanBlock), s,
                         h->d_feats, h->n, static_cast<int64_t>(0), iters, h->row_base, qa, kNoQueryPtr,
                         st.exclude, st.topn, h->d_stream_lists[buf], static_cast<float*>(nullptr),
                         static_cast<const uint64_t*>(nullptr), prev, ready, my_seed, st.epoch, next);
        }
    }
    HIP_TRY(h, hipGetLastError());
    // (the books move only once the launch is known to have been accepted)
    if (riders_arriving) h->ctl_done[next_buf] += riders_arriving;
    h->pending = true;
    h->pending_buf = buf;
    h->pending_topn = st.topn;
    h->pending_out = st.out;
    h->pending_lists = scanners;
    st.has = false;
    return MI355REC_OK;
}

int enqueue_streamed(mi355rec* h, const float* qptr, const float* query12, int64_t exclude_global, int topn,
                     uint64_t* out_keys, hipStream_t s) {
    int rc = ensure_streamed(h);
    if (rc) return rc;
    rc = flush_mstream(h, s);   // a stream of BATCHES on this handle is closed first
    if (rc) return rc;
    if (MI355REC_EXP_FLAG("MI355REC_EXP_RIDE_NOMERGE") && h->pending) {
        rc = flush_streamed(h, s);
        if (rc) return rc;
    }
    // One call behind: the query of the PREVIOUS call is launched now, and its launch takes the sample and the
    // neighbourhood of this one.  The first query of a stream needs a sample launch of its own.
    const int kind = single_kind(h, nullptr);
    int seed_buf = 0;
    bool sampled = false, nbhd_taken = false;
    const uint32_t epoch = next_epoch(h);   // the tag of this query's sample values and bound
    if (h->stashed.has) {
        seed_buf = 1 - h->stashed.seed_buf;
        // the riders of a launch sample the rows that launch scans: a change of rows (mi355rec_set_replica) between two
        // calls costs the next query a sample launch of its own
        const bool same = h->stashed.kind == kind;
        sampled = same && stream_riders(h, kind) > 0;
        nbhd_taken = same && stream_nbhd(h, kind);
        rc = launch_stashed(h, s, same, qptr, query12, exclude_global, topn, seed_buf, epoch);
        if (rc) return rc;
    }
    bool bound_ready = sampled && stream_hoists(h, kind);
    if (!sampled) {   // first query of a stream, or a shard too small to spare riders
        if (kind == kFp32) {
            if (h->n >= kF32LoneSeedMinRows)
                bound_ready = enqueue_f32_seed(h, qptr, query12, exclude_global, topn, h->d_stream_seed[seed_buf], h->d_stream_ctl + seed_buf,
                                               &h->ctl_done[seed_buf], epoch, s);
        } else if (!nbhd_taken) {
            QueryArg qa;
            std::memset(&qa, 0, sizeof qa);
            qa.margin = h->margin_mix;
            if (!qptr) std::memcpy(qa.q, query12, sizeof qa.q);
            enqueue_half_seed(h, kind, qptr, qa, exclude_global, topn, h->d_stream_seed[seed_buf], epoch, s);
        }
        HIP_TRY(h, hipGetLastError());
    }
    auto& st = h->stashed;
    st.has = true;
    st.qptr = qptr;
    if (!qptr) std::memcpy(st.q, query12, sizeof st.q);
    st.exclude = exclude_global;
    st.topn = topn;
    st.out = out_keys;
    st.seed_buf = seed_buf;
    st.epoch = epoch;
    st.kind = kind;
    st.cutoff_ready = bound_ready;
    return MI355REC_OK;
}

}  // namespace
