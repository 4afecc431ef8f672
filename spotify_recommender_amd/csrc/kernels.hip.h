// kernels.hip.h — the streaming scans over the fp32 rows (gfx950 only).
//
// One streaming pass over the row-major N x 12 fp32 catalogue replaces the
// reference's three device passes + host heap:
//   cublasSgemv            Recommender.cu:217-223   (dot products)
//   computeNormsKernel     Recommender.cu:48-59     (row norms, 2nd matrix read)
//   normalizeSimilarities  Recommender.cu:62-77     (divide / threshold / clamp)
//   host heap top-N        Recommender.cu:293-315
// Arithmetic: core.hip.h (the reference's CPU path, Recommender.cu:256-273, bit for bit).
#pragma once

#include "core.hip.h"
#include "merge.hip.h"
#include "handoff.hip.h"

#pragma clang fp contract(off)

namespace mi355 {

// ---- streaming scan ----------------------------------------------------------
// Tiles of kTileRows rows are dealt round-robin over the workgroups (or, with
// rows_per_block > 0, workgroup b owns a contiguous 64-row-aligned block); every
// wave-level load instruction covers one contiguous 3 KiB span.  One lane = one row: the 12-term sums are
// sequential in-lane, which is what makes the result bit-identical to the
// reference CPU loop.  The next tile is loaded before the current one is
// scored (24 waves x 64 lanes x 2 tiles x 48 B = ~150 KiB per CU in flight).
//
// kScoresOnly: write the n scores (mirrors calculateSimilarities' output).
// else: filter keys against the workgroup's running topk-th key, append the
// survivors to an LDS buffer, and leave the workgroup's sorted top-k list in
// block_lists[b][0..topk).

// kDebug (development A/B only; 0 in the product): 1 = no end-of-tile barrier pair,
// 2 = ballot only, no LDS append, 4 = skip the seed compaction (threshold preset).
// kWithMerge (streamed queries, mi355rec_enqueue_*_streamed): the LAST workgroup of the launch
// does not scan — it merges the per-workgroup lists of the PREVIOUS query (`prev`, a second
// list buffer) while workgroups 0 .. gridDim.x-2 scan this one, so the ~8 us one-workgroup merge
// kernel and its boundary disappear from the critical path of a stream of single queries.
// The host launches one scan workgroup fewer than are resident (767 + the merger on a
// 256-CU part), so nothing waits for a slot.
// LAUNCH-WIDE BOUND (handoff.hip.h).  Workgroup-local thresholds alone let every workgroup find its own way up: on a
// catalogue whose similar rows lie next to each other (sorted by genre) a workgroup meets a better cluster a few times
// per launch and each time a whole tile passes its stale threshold into LDS appends and a radix compaction (measured
// at 10 M rows, 300 contiguous clusters: 204 us per query instead of 81).  So the scan starts from the largest of two
// exact lower bounds of the shard's topk-th score, when the host has them: `bound_ctl` (what the last seed rider of
// the launch before, or the sample launch, made of a spread sample of the fp32 rows) and slot kNbhdSlot of
// `sample_buf` (the neighbourhood of the excluded row), both tagged with this query's `epoch`.  Exact scores of real
// rows: the key threshold starts AT the bound, no margin.  In a streamed launch (kWithMerge) the workgroups behind
// the merger are the seed riders of the NEXT query (`next`: f32_sample_regions, then the last one selects) and, last
// of the grid, its neighbourhood.
struct PrevMerge {
    const uint64_t* lists;   // [n_lists][topk] of the previous query, nullptr = nothing pending
    int n_lists;
    int topk;
    uint64_t* out_keys;
};
constexpr int kRideMaxLists = 1024;   // the riding merger's bounds (scan grids are <= 1023 workgroups)
constexpr int kRideSurvCap = 2048;

template <typename Cfg, bool kWithMerge>
struct ScanSmemT {
    uint64_t cand[Cfg::kCandCap];
    SelectSmem sel;
    int count;
    int seeds;   // (seed riders: the count of present sample values)
};
template <typename Cfg, bool kWithMerge>
union ScanOrMergeSmem {
    ScanSmemT<Cfg, kWithMerge> scan;
    MergeSmemT<Cfg::kBlock, kRideMaxLists, kRideSurvCap> merge;
};

template <typename Cfg, bool kQueryFromRow, bool kScoresOnly, int kDebug = 0, bool kWithMerge = false>
__global__ __launch_bounds__(Cfg::kBlock, Cfg::kMinWaves) void scan_kernel(
    const float* __restrict__ feats, int64_t n, int64_t rows_per_block, int iters,
    int64_t row_base, QueryArg qarg, const float* __restrict__ query_ptr, int64_t exclude_global,
    int topk, uint64_t* __restrict__ block_lists, float* __restrict__ scores_out,
    const uint64_t* __restrict__ upper_ptr, PrevMerge prev,
    const unsigned long long* __restrict__ bound_ctl /* tagged float: >= topk rows score at least this (null: none) */,
    const unsigned long long* __restrict__ sample_buf /* its slot kNbhdSlot: the neighbourhood's bound (null: none) */,
    uint32_t epoch /* of this query */, NextSeed next) {
    constexpr int kBlock = Cfg::kBlock;
    static_assert(!kWithMerge || kBlock == kHalfSeedBlock, "the seed riders are sampling workgroups");
    constexpr int kRowsPerThread = Cfg::kRowsPerThread;
    constexpr int kTileRows = Cfg::kTileRows;
    static_assert(!(kWithMerge && kScoresOnly), "the riding merger belongs to top-N scans");
    MI355REC_KPHASE(0);
    // Plain statics for the ordinary scan (exactly the round-1 layout: the kernel sits at
    // 79 of its 80 VGPRs and one more costs a spill); the union only in the riding variant.
    __shared__ uint64_t s_cand_plain[(kScoresOnly || kWithMerge) ? 1 : Cfg::kCandCap];
    __shared__ SelectSmem s_sel_plain;
    __shared__ int s_count_plain;
    __shared__ typename std::conditional<kWithMerge, ScanOrMergeSmem<Cfg, true>, int>::type s_ride;
    // scanning workgroups [0, nblocks), then the merger, then the seed riders, then (next.nbhd) the neighbourhood
    const unsigned nblocks = kWithMerge ? gridDim.x - 1u - static_cast<unsigned>(next.n_wgs) - static_cast<unsigned>(next.nbhd) : gridDim.x;
    if constexpr (kWithMerge) {
        if (blockIdx.x >= nblocks) {
            if (blockIdx.x == nblocks) {      // the merger (behind the scanners: they keep blockIdx = tile slot)
                if (prev.lists)
                    merge_body(s_ride.merge, prev.lists, prev.n_lists, prev.topk, static_cast<int64_t>(prev.topk),
                               static_cast<int64_t>(0), prev.topk, prev.out_keys, static_cast<int64_t*>(nullptr),
                               static_cast<float*>(nullptr), static_cast<int64_t>(0), static_cast<int64_t>(0),
                               static_cast<int64_t>(0));
                MI355REC_KPHASE(5);
            } else if (next.nbhd && blockIdx.x == gridDim.x - 1u) {   // the next query's neighbourhood
                // (2048 rows, two per thread in flight at a time: the scanners of this kernel live in 80 VGPRs and four rows
                // in flight beside their keys did not fit; this workgroup has the whole launch to itself)
                float nq[kDim];
                if (next.query_ptr) {
#pragma unroll
                    for (int j = 0; j < kDim; ++j) nq[j] = next.query_ptr[j];
                } else {
#pragma unroll
                    for (int j = 0; j < kDim; ++j) nq[j] = next.q[j];
                }
                const uint32_t v = nbhd_bound_rounds<kBlock, 2, 2>(feats, n, row_base, next.exclude_global, nq, query_norm(nq), next.topk,
                                                                   reinterpret_cast<int*>(s_ride.scan.cand), next.anchors);
                if (threadIdx.x == 0) static_cast<unsigned long long*>(next.out)[kNbhdSlot] = tag_value(next.epoch, v);
                MI355REC_KPHASE(5);
            } else {                              // a seed rider: its share of the next query's sample
                f32_sample_regions<2>(feats, n, row_base, next, static_cast<int>(blockIdx.x - nblocks - 1u));
                if (next.ctl) {   // uniform: the last rider out turns the sample into the next launch's bound
                    float v;
                    if (sample_arrive_and_select<kBlock>(next.ctl, next.done_base, static_cast<unsigned>(next.n_wgs),
                                                         static_cast<const unsigned long long*>(next.out), next.regions * kHalfSeedWaves,
                                                         next.topk, next.epoch, &s_ride.scan.count, &s_ride.scan.seeds, s_ride.scan.sel,
                                                         reinterpret_cast<int*>(s_ride.scan.cand), v)) {
                        if (threadIdx.x == 0) next.ctl->cutoff = tag_value(next.epoch, __float_as_uint(v));
                    }
                }
                MI355REC_KPHASE(5);
            }
            return;
        }
    } else {
        (void)next;
    }
    // this workgroup's index among the scanning workgroups
    const unsigned bid = blockIdx.x;
    uint64_t* s_cand;
    SelectSmem* s_sel_p;
    int* s_count_p;
    if constexpr (kWithMerge) {
        s_cand = s_ride.scan.cand;
        s_sel_p = &s_ride.scan.sel;
        s_count_p = &s_ride.scan.count;
    } else {
        (void)s_ride;
        s_cand = s_cand_plain;
        s_sel_p = &s_sel_plain;
        s_count_p = &s_count_plain;
    }
    SelectSmem& s_sel = *s_sel_p;
    int& s_count = *s_count_p;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // rows_per_block > 0: workgroup b owns the contiguous rows [b*rpb, (b+1)*rpb).
    // rows_per_block == 0: tiles are dealt round-robin (tile t -> workgroup t % grid),
    // so at any moment the whole chip reads one moving ~20 MB window of the matrix.
    const bool interleaved = rows_per_block == 0;
    const int64_t blk_begin = static_cast<int64_t>(bid) * (interleaved ? kTileRows : rows_per_block);
    const int64_t tile_stride = interleaved ? static_cast<int64_t>(nblocks) * kTileRows : kTileRows;
    int64_t blk_end = interleaved ? n : blk_begin + rows_per_block;
    if (blk_end > n) blk_end = n;
    // rows past the block's end re-read its last row (one cached line) so the
    // prefetch can be unconditional: a conditional load would make the compiler
    // wait vmcnt(0) at the join and serialise the pipeline
    const int64_t last_row = blk_end - 1;  // the host launches only non-empty blocks

    auto load_tile = [&](Row (&dst)[kRowsPerThread], int it) {
        const int64_t tile_begin = blk_begin + static_cast<int64_t>(it) * tile_stride;
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            const int64_t r = tile_begin + u * kBlock + tid;
            dst[u] = load_row(feats, r < blk_end ? r : last_row);
        }
    };

    // The first tile(s) are requested BEFORE the query is: the query's 12 floats sit behind two dependent scalar loads
    // (the pointer, then the row) and a norm, the tile behind one vector load, and neither needs the other — issued in
    // source order the tile waited ~1.5-2 us for the query on every launch (a sixth of a 1 M-row scan).
    constexpr int kDepth = Cfg::kDepth;
    Row ring[kDepth][kRowsPerThread];
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d) load_tile(ring[d], d);

    float q[kDim];
    if constexpr (kQueryFromRow) {
        // 12 floats anywhere this device can read: a resident row of this shard, a row of
        // ANOTHER shard through the peer mapping (sharded.hip), a staged vector.  Wave-uniform: scalar loads.
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = query_ptr[j];
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = qarg.q[j];
    }
    // (wave-uniform: scalar loads, requested with the query)
    const unsigned long long bound_raw = bound_ctl ? *bound_ctl : 0ull;
    const unsigned long long nbhd_raw = sample_buf ? sample_buf[kNbhdSlot] : 0ull;
    const float qn = query_norm(q);
    if (qn >= 0.0f || n >= 0) MI355REC_KPHASE(1);   // (depends on the query)
    // Only keys strictly below *upper_ptr take part (nullptr: no bound).  This is
    // how topn > kMaxTopK is served: round r asks for the best kMaxTopK keys
    // below the last key of round r-1 (0 there = catalogue exhausted).
    // (streamed queries are single-round: no bound, and two VGPRs the riding variant needs)
    const uint64_t upper = kWithMerge ? ~0ull : (upper_ptr ? *upper_ptr : ~0ull);

    if constexpr (!kScoresOnly) {
        if (tid == 0) s_count = 0;
        __syncthreads();
    }
    uint64_t thr = 0;
    float cutoff = 0.0f;  // approx pre-filter active only while > 0
    const float inv_qn = 1.0f / qn;
    const bool prefilter_ok = qn < kApproxMaxQueryNorm;  // false for inf / NaN norms too
    if constexpr ((kDebug & 4) != 0) {
        thr = pack_key(0.985f, 0u);
        cutoff = 0.985f - kApproxMargin;
    }
    if constexpr (!kScoresOnly) {
        // the launch-wide bound: the larger of the sample's and the neighbourhood's (0 / -inf: absent, or another query's)
        uint32_t vb = untag_value(nbhd_raw, epoch);
        const float sample_v = untag_cutoff(bound_raw, epoch);
        if (sample_v > -3.0e38f) {
            const uint32_t o = score_to_ordered(sample_v);
            vb = o > vb ? o : vb;
        }
        if (vb) {   // at least topk rows score >= the bound: keys below it are dropped (a key AT it passes: key > thr)
            thr = (static_cast<uint64_t>(vb) << 32) - 1ull;
            const float vs = ordered_to_score(vb);
            if (prefilter_ok && vs > 0.0f) cutoff = vs - kApproxMargin;
        }
    }
    // Re-tighten the threshold once ~topk NEW candidates have piled up (the
    // select is O(c), but every candidate that slips past a stale threshold
    // costs an LDS atomic round trip in the streaming loop).
    int compact_at = 2 * topk > 256 ? 2 * topk : 256;
    if (compact_at > kCandLimit) compact_at = kCandLimit;

    auto process_tile = [&](const Row (&rows)[kRowsPerThread], int it) {
        const int64_t tile_begin = blk_begin + static_cast<int64_t>(it) * tile_stride;
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            const int64_t r = tile_begin + u * kBlock + tid;
            const bool in_range = r < blk_end;
            if constexpr (kScoresOnly) {
                const float s = cosine_score(q, qn, rows[u]);
                if (in_range) scores_out[r] = s;
            } else {
                // While the threshold score is positive, rows whose cheap upper
                // bound cannot reach it are dropped without the exact chain.
                bool maybe = true;
                if (cutoff > 0.0f) maybe = !(approx_cosine(q, inv_qn, rows[u]) < cutoff);
                if (__ballot(maybe)) {
                    const float s = cosine_score(q, qn, rows[u]);
                    const int64_t g = row_base + r;
                    uint64_t key = pack_key(s, static_cast<uint32_t>(g));
                    if (!in_range || g == exclude_global || key >= upper) key = 0;
                    const bool pass = maybe && key > thr;
                    const uint64_t ballot = __ballot(pass);
                    if (ballot && (kDebug & 2) == 0) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_count, __popcll(ballot));
                        base = __builtin_amdgcn_readfirstlane(base);
                        const int pos = base + lanes_below(ballot);
                        if (pass) s_cand[pos] = key;
                    }
                }
            }
        }
        if constexpr (!kScoresOnly && (kDebug & 5) == 0) {
            // Two barriers: every wave must have read the count before any wave
            // appends again, or the waves could disagree about compacting.
            __syncthreads();
            const int c = s_count;
            __syncthreads();
            if (c >= compact_at) {
                const uint64_t local_thr = compact_candidates<kBlock, Cfg::kCandPerThread>(s_cand, &s_count, topk, false, s_sel);
                if (local_thr > thr) {
                    thr = local_thr;
                    if (prefilter_ok) cutoff = ordered_to_score(static_cast<uint32_t>(thr >> 32)) - kApproxMargin;
                }
            }
        }
    };

    // kDepth tiles in flight per lane: the loads of tile it + kDepth - 1 are issued
    // before tile it is scored (register ring, statically indexed).
    for (int it = 0; it < iters; it += kDepth) {
#pragma unroll
        for (int sidx = 0; sidx < kDepth; ++sidx) {
            load_tile(ring[(sidx + kDepth - 1) % kDepth], it + sidx + kDepth - 1);
            if (it + sidx < iters) process_tile(ring[sidx], it + sidx);  // uniform
        }
    }

    MI355REC_KPHASE(3);   // tiles done
    if constexpr (!kScoresOnly) {
        __syncthreads();
        // (from kRankCountMax keys up the ranking is a bitonic sort of the next power of two — 45 barrier stages, ~2.7 us
        // for the ~250 keys a scan without a launch-wide cutoff ends with; an INEXACT cut to topk + topk / 4 first costs a
        // radix pass or two and leaves a set the counting rank handles)
        if (s_count > kRankCountMax && s_count > topk)  // uniform
            compact_candidates<kBlock, Cfg::kCandPerThread>(s_cand, &s_count, topk, false, s_sel);
        __syncthreads();
        block_rank_and_store<kBlock>(s_cand, s_count, block_lists + static_cast<int64_t>(bid) * topk, topk);
        MI355REC_KPHASE(4);   // list stored
    }
}

// ---- the sample launch of a query ALONE over the fp32 rows -----------------------------------------------------
// A streamed query's sample rides in the launch before it (scan_kernel<.., kWithMerge>); a query on its own gets this
// launch first where the shard is large enough for ~5 us to be worth a launch-wide bound: workgroup g < next.regions
// takes region g (next.n_wgs == next.regions), the last to arrive selects the bound into next.ctl->cutoff, and
// (next.nbhd) one more workgroup takes the neighbourhood of the excluded row into next.out[kNbhdSlot].
__global__ __launch_bounds__(kHalfSeedBlock) void seed_f32_kernel(const float* __restrict__ feats, int64_t n, int64_t row_base, NextSeed next) {
    __shared__ SelectSmem s_sel;
    __shared__ int s_flag, s_seeds;
    __shared__ int s_bins[Nbhd<kHalfSeedBlock>::kScratch > kSelScratch ? Nbhd<kHalfSeedBlock>::kScratch : kSelScratch];
    if (static_cast<int>(blockIdx.x) >= next.regions) {   // uniform: the neighbourhood workgroup
        nbhd_to_slot<kHalfSeedBlock>(feats, n, row_base, next.query_ptr, next.q, next.exclude_global, next.topk, next.epoch,
                                     static_cast<unsigned long long*>(next.out), s_bins, next.anchors);
        return;
    }
    f32_sample_regions(feats, n, row_base, next, static_cast<int>(blockIdx.x));
    if (next.ctl) {   // uniform
        float v;
        if (sample_arrive_and_select<kHalfSeedBlock>(next.ctl, next.done_base, static_cast<unsigned>(next.n_wgs),
                                                     static_cast<const unsigned long long*>(next.out), next.regions * kHalfSeedWaves, next.topk,
                                                     next.epoch, &s_flag, &s_seeds, s_sel, s_bins, v)) {
            if (threadIdx.x == 0) next.ctl->cutoff = tag_value(next.epoch, __float_as_uint(v));
        }
    }
}

// ---- multi-query streaming scan ------------------------------------------------
// One pass over the catalogue scores kQ queries at once: the 48 B of a row are
// fetched once and reused from registers, so the pass costs about the same HBM
// time as a single query while answering kQ of them (~8 + 16*kQ VALU instructions
// per 64 rows; 135 us for 12 queries vs 84 us for one at 10 M rows).  Per query the logic is the
// single-query kernel's: packed-FMA upper bound against that query's running
// threshold, exact in-order re-score of the rare rows that may beat it, LDS
// candidate buffer, O(c) radix select when candidates pile up.  Differences:
//  * each query has its own candidate buffer, count and threshold in LDS;
//  * compaction is WAVE-level (wave w serves queries w, w + waves, ...), so the
//    kQ compactions of a tile boundary run in parallel between two barriers;
//  * the row norm (6 packed FMAs + rsq) is shared by all queries.
// Output: block_lists[q][workgroup][topk], each list sorted descending.

constexpr int kMultiQueries = 12;     // queries per pass (12 x 5.5 KB of candidates: 2 workgroups per CU)
constexpr int kMultiChain = 36;       // queries whose seed / final merges share one launch
constexpr int kMultiMaxTopK = 128;    // larger topn goes through the single-query kernel
constexpr int kMultiCompactAt = 192;  // > kMultiMaxTopK + the select's slack (32): a compaction always makes room

template <int kBlockT, int kRowsT, int kMinWavesT>
struct MultiCfg {
    static constexpr int kBlock = kBlockT;
    static constexpr int kRowsPerThread = kRowsT;
    static constexpr int kMinWaves = kMinWavesT;
    static constexpr int kTileRows = kBlockT * kRowsT;
    static constexpr int kWaves = kBlockT / 64;
    static constexpr int kCap = kMultiCompactAt + kTileRows;   // per-query candidate slots
    static constexpr int kKeysPerLane = (kCap + 63) / 64;
};
using DefaultMultiCfg = MultiCfg<512, 1, 4>;

struct MultiQueryArg {
    float q[kMultiQueries][kDim];
    long long exclude[kMultiQueries];   // global row to skip per query, -1 = none
};

// One group of up to kMultiQueries queries against the whole shard.  `load_query(t,
// q12, excl)` hands thread t < kQ its query (called by the first kQ threads only).
// Lists go to block_lists[list_slot0 + qi][workgroup][topk].  Called by every thread
// of the workgroup; may be called again after it returns (it ends on a barrier-free
// tail, so the caller puts a __syncthreads() between two groups).
template <typename Cfg, typename LoadQuery>
__device__ __forceinline__ void multi_scan_group(
    const float* __restrict__ feats, int64_t n, int64_t rows_per_block, int64_t block_stride, int iters,
    int64_t row_base, LoadQuery load_query, int n_queries, int list_slot0, int topk,
    uint64_t* __restrict__ block_lists, const uint64_t* __restrict__ seed_keys, int seed_slot0) {
    constexpr int kBlock = Cfg::kBlock;
    constexpr int kRowsPerThread = Cfg::kRowsPerThread;
    constexpr int kTileRows = Cfg::kTileRows;
    constexpr int kQ = kMultiQueries;
    __shared__ uint64_t s_cand[kQ][Cfg::kCap];
    __shared__ int s_hist[Cfg::kWaves][256];
    __shared__ int s_count[kQ];
    __shared__ uint64_t s_thr[kQ];   // running filter threshold per query (key > thr passes)
    __shared__ float4 s_qc[kQ];      // {cutoff of the approx pre-filter, 1/|q|, |q|, unused}
    __shared__ v4f s_q[kQ][3];       // the query vectors (broadcast ds_read_b128; 96 SGPRs would spill)
    __shared__ long long s_excl[kQ]; // global row to skip per query, -1 = none (read on the rare exact path only)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    // rows_per_block > 0 (development harness): workgroup b scans rows_per_block
    // rows starting at b * block_stride.
    // rows_per_block == 0: tiles dealt round-robin over the workgroups (full pass).
    const bool interleaved = rows_per_block == 0;
    const int64_t blk_begin = static_cast<int64_t>(blockIdx.x) * (interleaved ? kTileRows : block_stride);
    const int64_t tile_stride = interleaved ? static_cast<int64_t>(gridDim.x) * kTileRows : kTileRows;
    int64_t blk_end = interleaved ? n : blk_begin + rows_per_block;
    if (blk_end > n) blk_end = n;
    const int64_t last_row = blk_end - 1;

    // Per-query state lives in LDS (broadcast reads), not in ~50 scalar registers.
    // cutoff = -inf disables the pre-filter until the threshold score is > 0.
    // seed_keys (optional): slot [query][topk - 1] holds a key that bounds the
    // catalogue's topk-th best from below (seed_multi_kernel + seed_select_kernel
    // over a sample, or 0 = none), so every workgroup starts with a chip-wide
    // threshold instead of re-deriving a weak local one — the exact re-score path
    // then runs for ~1e-4 of the rows.
    if (tid < kQ) {
        float qv[kDim];
        long long excl = -1;
        if (tid < n_queries) {
            load_query(tid, qv, excl);
        } else {
#pragma unroll
            for (int j = 0; j < kDim; ++j) qv[j] = 0.0f;
        }
        const float norm = query_norm(qv);
        uint64_t t = 0ull;
        if (seed_keys && tid < n_queries) t = seed_keys[static_cast<int64_t>(seed_slot0 + tid) * topk + (topk - 1)];
        float cut = -__builtin_inff();
        if (t) {
            const float score = ordered_to_score(static_cast<uint32_t>(t >> 32));
            if (score > 0.0f && norm < kApproxMaxQueryNorm) cut = score - kApproxMargin;
            t -= 1ull;
        }
        s_count[tid] = 0;
        s_thr[tid] = t;
        s_excl[tid] = excl;
        s_qc[tid] = make_float4(cut, 1.0f / norm, norm, 0.0f);
        s_q[tid][0] = v4f{qv[0], qv[1], qv[2], qv[3]};
        s_q[tid][1] = v4f{qv[4], qv[5], qv[6], qv[7]};
        s_q[tid][2] = v4f{qv[8], qv[9], qv[10], qv[11]};
    }
    __syncthreads();

    auto load_tile = [&](Row (&dst)[kRowsPerThread], int it) {
        const int64_t tile_begin = blk_begin + static_cast<int64_t>(it) * tile_stride;
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            const int64_t r = tile_begin + u * kBlock + tid;
            dst[u] = load_row(feats, r < blk_end ? r : last_row);
        }
    };

    auto process_tile = [&](const Row (&rows)[kRowsPerThread], int it) {
        const int64_t tile_begin = blk_begin + static_cast<int64_t>(it) * tile_stride;
        // shared by all queries: approximate 1/|row| of this lane's rows
        float inv_norm[kRowsPerThread];
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            const Row& row = rows[u];
            v2f m = {row.a.x * row.a.x, row.a.y * row.a.y};
            m = __builtin_elementwise_fma(v2f{row.a.z, row.a.w}, v2f{row.a.z, row.a.w}, m);
            m = __builtin_elementwise_fma(v2f{row.b.x, row.b.y}, v2f{row.b.x, row.b.y}, m);
            m = __builtin_elementwise_fma(v2f{row.b.z, row.b.w}, v2f{row.b.z, row.b.w}, m);
            m = __builtin_elementwise_fma(v2f{row.c.x, row.c.y}, v2f{row.c.x, row.c.y}, m);
            m = __builtin_elementwise_fma(v2f{row.c.z, row.c.w}, v2f{row.c.z, row.c.w}, m);
            const float nrm2 = m.x + m.y;
            // NaN makes every "approx < cutoff" below false: rows whose sums may
            // overflow are always re-scored exactly (see kApproxMaxNorm2)
            inv_norm[u] = nrm2 < kApproxMaxNorm2 ? __builtin_amdgcn_rsqf(nrm2) : __builtin_nanf("");
        }
        // (unrolling this loop by 4 was measured: 156 us per 12-query pass instead of 130)
        for (int qi = 0; qi < n_queries; ++qi) {  // uniform trip count
            // three broadcast ds_read_b128; the .xy / .zw halves feed the packed FMAs directly
            const v4f qa = s_q[qi][0], qb = s_q[qi][1], qcv = s_q[qi][2];
            const float4 qc = s_qc[qi];
#pragma unroll
            for (int u = 0; u < kRowsPerThread; ++u) {
                const Row& row = rows[u];
                v2f d = v2f{row.a.x, row.a.y} * qa.xy;
                d = __builtin_elementwise_fma(v2f{row.a.z, row.a.w}, qa.zw, d);
                d = __builtin_elementwise_fma(v2f{row.b.x, row.b.y}, qb.xy, d);
                d = __builtin_elementwise_fma(v2f{row.b.z, row.b.w}, qb.zw, d);
                d = __builtin_elementwise_fma(v2f{row.c.x, row.c.y}, qcv.xy, d);
                d = __builtin_elementwise_fma(v2f{row.c.z, row.c.w}, qcv.zw, d);
                const bool maybe = !((d.x + d.y) * inv_norm[u] * qc.y < qc.x);
                if (__ballot(maybe)) {
                    const int64_t r = tile_begin + u * kBlock + tid;
                    const int64_t g = row_base + r;
                    const float q[kDim] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w, qcv.x, qcv.y, qcv.z, qcv.w};
                    const float s = cosine_score(q, qc.z, row);
                    uint64_t key = pack_key(s, static_cast<uint32_t>(g));
                    if (r >= blk_end || g == s_excl[qi]) key = 0;
                    const bool pass = maybe && key > s_thr[qi];
                    const uint64_t ballot = __ballot(pass);
                    if (ballot) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&s_count[qi], __popcll(ballot));
                        base = __builtin_amdgcn_readfirstlane(base);
                        const int pos = base + lanes_below(ballot);
                        if (pass) s_cand[qi][pos] = key;
                    }
                }
            }
        }
        // tile boundary: does any query need its threshold tightened?
        __syncthreads();
        bool any = false;
#pragma unroll
        for (int qi = 0; qi < kQ; ++qi) any = any || s_count[qi] >= kMultiCompactAt;
        __syncthreads();
        if (any) {  // uniform (every thread read the same counts between the barriers)
            for (int qi = wave; qi < kQ; qi += Cfg::kWaves) {
                if (s_count[qi] >= kMultiCompactAt) {
                    const uint64_t t = wave_compact<Cfg::kKeysPerLane>(s_cand[qi], &s_count[qi], topk, false, s_hist[wave]);
                    if (lane == 0 && t > s_thr[qi]) {
                        s_thr[qi] = t;
                        const float score = ordered_to_score(static_cast<uint32_t>(t >> 32));
                        if (score > 0.0f && s_qc[qi].z < kApproxMaxQueryNorm) s_qc[qi].x = score - kApproxMargin;
                    }
                }
            }
            __syncthreads();
        }
    };

    Row buf_a[kRowsPerThread];
    Row buf_b[kRowsPerThread];
    load_tile(buf_a, 0);
    for (int it = 0; it < iters; it += 2) {
        load_tile(buf_b, it + 1);   // clamped to the block's rows: harmless past the end
        process_tile(buf_a, it);
        if (it + 1 >= iters) break;  // uniform
        load_tile(buf_a, it + 2);
        process_tile(buf_b, it + 1);
    }

    // final: every query's best topk of this workgroup, sorted, one wave per query
    __syncthreads();
    for (int qi = wave; qi < n_queries; qi += Cfg::kWaves) {
        if (s_count[qi] > kRankDirectMax && s_count[qi] > topk)
            wave_compact<Cfg::kKeysPerLane>(s_cand[qi], &s_count[qi], topk, true, s_hist[wave]);
        const int c = __builtin_amdgcn_readfirstlane(s_count[qi]);
        uint64_t* dst = block_lists + (static_cast<int64_t>(list_slot0 + qi) * gridDim.x + blockIdx.x) * topk;
        wave_rank_and_store(s_cand[qi], c, dst, topk);
    }
}

template <typename Cfg>
__global__ __launch_bounds__(Cfg::kBlock, Cfg::kMinWaves) void scan_multi_kernel(
    const float* __restrict__ feats, int64_t n, int64_t rows_per_block, int64_t block_stride, int iters,
    int64_t row_base, MultiQueryArg qarg, int n_queries, int query_slot0, int topk,
    uint64_t* __restrict__ block_lists, const uint64_t* __restrict__ seed_keys) {
    auto load_query = [&](int t, float (&qv)[kDim], long long& excl) {
#pragma unroll
        for (int j = 0; j < kDim; ++j) qv[j] = qarg.q[t][j];
        excl = qarg.exclude[t];
    };
    multi_scan_group<Cfg>(feats, n, rows_per_block, block_stride, iters, row_base, load_query, n_queries,
                          query_slot0, topk, block_lists, seed_keys, query_slot0);
}

// The same pass for queries QUEUED on the device (batched.hip.h): queue[0..*count)
// are indices into queries_dev / exclude_dev; they are served in groups of
// kMultiQueries inside ONE launch (no seed: this is the rare, robust path), lists go
// to block_lists[position in the queue][workgroup][topk].  Exits at once when the
// queue is empty — the usual case, in which this launch costs its dispatch and nothing else.
// The merge of those lists is part of the launch as well (it used to be a second kernel that
// every batched call paid for): the workgroup that finishes a GROUP last merges that group's lists
// into its queries' output rows while the others scan the next group — in practice the same workgroup
// for every group, see below (`arrive`: one counter per group,
// kBqMaxQueries / kMultiQueries < 128 of them).  Rare path, so the hand-off is the plain one: a device-wide
// fence before each workgroup counts itself out and one behind the count of the last (the counters are zero
// between launches: the last workgroup of a group resets its own, and a launch with an empty queue touches none).
// (Launch bounds it can meet: one workgroup per CU is all the launch ever has — its grid is the CU count and the merge's
// shared memory, 120 KB, allows no second one — so two waves per SIMD, not the pass kernel's four: the compiler used to
// warn "desired occupancy was 4, final occupancy is 2".)
template <typename Cfg>
__global__ __launch_bounds__(Cfg::kBlock, 2) void scan_multi_queued_kernel(
    const float* __restrict__ feats, int64_t n, int iters, int64_t row_base,
    const float* __restrict__ queries_dev, const long long* __restrict__ exclude_dev,
    const int* __restrict__ queue, const int* __restrict__ queue_count, int topk,
    uint64_t* __restrict__ block_lists, unsigned* __restrict__ arrive,
    uint64_t* __restrict__ out_keys_base, int64_t* __restrict__ out_idx_base, float* __restrict__ out_score_base,
    int64_t out_query_stride) {
    const int count = *queue_count;
    if (count == 0) return;   // uniform over the whole grid
    // Every group of kMultiQueries queued queries is counted out on its OWN counter, and the workgroup that leaves a group
    // last merges that group's queries while the others go on scanning the next group.  What that buys is modest:
    // the workgroup that merged group g starts group g + 1 a merge late, so it is the last one out of g + 1 as well and
    // merges that too — ONE workgroup still does every merge, and the chain is groups x (scan + merge), only with the other
    // workgroups' scans no longer waiting for it (a whole 1024-query chunk in the queue, hostile data: 58 -> 55.8 ms).  Handing
    // the merges to whichever workgroups are idle would take a ticket counter and a wait on the groups' counters; on a path
    // that exists for inputs the bound cannot be claimed for, that was not built.
    // (`arrive[g]` is zero between launches: the last workgroup of a group resets it.)
    __shared__ int s_last;
    __shared__ MergeSmemT<Cfg::kBlock, 1024, kMergeSurvCap> s_merge;   // (one workgroup per CU fits with this: the launch has no more)
    for (int g0 = 0; g0 < count; g0 += kMultiQueries) {
        const int nq = count - g0 < kMultiQueries ? count - g0 : kMultiQueries;
        auto load_query = [&](int t, float (&qv)[kDim], long long& excl) {
            const int q = queue[g0 + t];
#pragma unroll
            for (int j = 0; j < kDim; ++j) qv[j] = queries_dev[static_cast<int64_t>(q) * kDim + j];
            excl = exclude_dev ? exclude_dev[q] : -1ll;
        };
        multi_scan_group<Cfg>(feats, n, static_cast<int64_t>(0), static_cast<int64_t>(0), iters, row_base, load_query,
                              nq, g0, topk, block_lists, static_cast<const uint64_t*>(nullptr), 0);
        __syncthreads();
        __threadfence();   // this workgroup's lists of the group are visible device-wide ...
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* const ctr = arrive + g0 / kMultiQueries;
            const bool last = atomicAdd(ctr, 1u) + 1u == gridDim.x;   // ... before it is counted
            if (last) *ctr = 0u;
            s_last = last ? 1 : 0;
        }
        __syncthreads();
        if (s_last) {          // uniform
            __threadfence();   // ... and the last one sees everybody's
            for (int b = g0; b < g0 + nq; ++b) {
                merge_body(s_merge, block_lists, static_cast<int>(gridDim.x), topk, static_cast<int64_t>(topk),
                           static_cast<int64_t>(gridDim.x) * topk, topk, out_keys_base, out_idx_base, out_score_base, out_query_stride,
                           static_cast<int64_t>(b), static_cast<int64_t>(queue[b]));
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// ---- seed for the multi-query pass ----------------------------------------------
// A chip-wide starting threshold per query from a small sample, with the cheap
// arithmetic only.  seed_multi_kernel: workgroup b looks at the first kSeedBlock
// rows of region b (regions are `block_stride` rows apart, so the sample is spread
// over the catalogue) and, per query, every wave writes the best APPROXIMATE
// cosine of its 64 rows (order-preserving u32 image).  seed_select_kernel: per
// query the topk-th largest of those grid*waves values, a_K.  Then topk distinct
// rows have approx >= a_K, hence exact score >= a_K - 2e-6, so
//     T = a_K - kApproxMargin
// bounds the catalogue's topk-th exact score from below.  Rows whose exact score
// is not the plain quotient (|row|*|q| <= 1e-8 -> 0 in the reference, NaN, the
// excluded row) never enter the sample, and the bound is only used when T > 0.
constexpr int kSeedBlock = 512;
constexpr int kSeedWaves = kSeedBlock / 64;

struct SeedQueryArg {
    float q[kMultiChain][kDim];
    long long exclude[kMultiChain];
};

__global__ __launch_bounds__(kSeedBlock) void seed_multi_kernel(
    const float* __restrict__ feats, int64_t n, int64_t block_stride, int64_t row_base,
    SeedQueryArg qarg, int n_queries, uint32_t* __restrict__ out /* [query][grid * kSeedWaves] */) {
    __shared__ v4f s_q[kMultiChain][3];
    __shared__ float4 s_qc[kMultiChain];  // {1/|q|, |q|^2, unused, unused}
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    if (tid < kMultiChain) {
        const float norm = query_norm(qarg.q[tid]);
        s_qc[tid] = make_float4(1.0f / norm, norm * norm, 0.0f, 0.0f);
        s_q[tid][0] = v4f{qarg.q[tid][0], qarg.q[tid][1], qarg.q[tid][2], qarg.q[tid][3]};
        s_q[tid][1] = v4f{qarg.q[tid][4], qarg.q[tid][5], qarg.q[tid][6], qarg.q[tid][7]};
        s_q[tid][2] = v4f{qarg.q[tid][8], qarg.q[tid][9], qarg.q[tid][10], qarg.q[tid][11]};
    }
    __syncthreads();
    const int64_t r = static_cast<int64_t>(blockIdx.x) * block_stride + tid;
    const bool in_range = r < n;
    const Row row = load_row(feats, in_range ? r : n - 1);
    const int64_t g = row_base + r;
    v2f m = {row.a.x * row.a.x, row.a.y * row.a.y};
    m = __builtin_elementwise_fma(v2f{row.a.z, row.a.w}, v2f{row.a.z, row.a.w}, m);
    m = __builtin_elementwise_fma(v2f{row.b.x, row.b.y}, v2f{row.b.x, row.b.y}, m);
    m = __builtin_elementwise_fma(v2f{row.b.z, row.b.w}, v2f{row.b.z, row.b.w}, m);
    m = __builtin_elementwise_fma(v2f{row.c.x, row.c.y}, v2f{row.c.x, row.c.y}, m);
    m = __builtin_elementwise_fma(v2f{row.c.z, row.c.w}, v2f{row.c.z, row.c.w}, m);
    const float nrm2 = m.x + m.y;
    const float inv_norm = __builtin_amdgcn_rsqf(nrm2);
    const int64_t per_query = static_cast<int64_t>(gridDim.x) * kSeedWaves;
    for (int qi = 0; qi < n_queries; ++qi) {
        const v4f qa = s_q[qi][0], qb = s_q[qi][1], qcv = s_q[qi][2];
        const float4 qc = s_qc[qi];
        v2f d = v2f{row.a.x, row.a.y} * qa.xy;
        d = __builtin_elementwise_fma(v2f{row.a.z, row.a.w}, qa.zw, d);
        d = __builtin_elementwise_fma(v2f{row.b.x, row.b.y}, qb.xy, d);
        d = __builtin_elementwise_fma(v2f{row.b.z, row.b.w}, qb.zw, d);
        d = __builtin_elementwise_fma(v2f{row.c.x, row.c.y}, qcv.xy, d);
        d = __builtin_elementwise_fma(v2f{row.c.z, row.c.w}, qcv.zw, d);
        float a = (d.x + d.y) * inv_norm * qc.x;
        // (|row| * |q|)^2 comfortably above (1e-8)^2, finite, not the excluded row
        const bool ok = in_range && g != qarg.exclude[qi] && (nrm2 * qc.y > 4e-16f) && (a - a == 0.0f) &&
                        nrm2 < kApproxMaxNorm2 && qc.y < kApproxMaxNorm2;
        if (!ok) a = -__builtin_inff();
        const uint32_t best = wave_max_u32(score_to_ordered(a));
        if (lane == 0) out[qi * per_query + static_cast<int64_t>(blockIdx.x) * kSeedWaves + wave] = best;
    }
}

constexpr int kSeedSelectPerThread = 8;  // up to kMergeBlock * 8 sample maxima per query

__global__ __launch_bounds__(kMergeBlock) void seed_select_kernel(
    const uint32_t* __restrict__ vals, int count, int topk, uint64_t* __restrict__ seed_keys) {
    __shared__ SelectSmem s_sel;
    const int tid = threadIdx.x;
    const uint32_t* mine_vals = vals + static_cast<int64_t>(blockIdx.x) * count;
    uint64_t mine[kSeedSelectPerThread];
#pragma unroll
    for (int r = 0; r < kSeedSelectPerThread; ++r) {
        const int i = tid + r * kMergeBlock;
        // unique keys: value in the high word, position in the low word
        mine[r] = i < count ? ((static_cast<uint64_t>(mine_vals[i]) << 32) | static_cast<uint32_t>(i + 1)) : 0ull;
    }
    uint64_t t = 0ull;
    if (count >= topk) {  // uniform
        const uint64_t kth = block_select_threshold<kMergeBlock, kSeedSelectPerThread>(mine, topk, true, 0, s_sel);
        const float a_k = ordered_to_score(static_cast<uint32_t>(kth >> 32));
        const float bound = a_k - kApproxMargin;  // -inf stays -inf
        if (bound > 0.0f) t = static_cast<uint64_t>(score_to_ordered(bound)) << 32;
    }
    // scan_multi_kernel reads slot [query][topk - 1] as "the sample's topk-th key"
    if (tid == 0) seed_keys[static_cast<int64_t>(blockIdx.x) * topk + (topk - 1)] = t;
}

// ---- read-only streaming probe (achievable-HBM ceiling) ---------------------

__global__ __launch_bounds__(kProbeBlock) void stream_probe_kernel(
    const float4* __restrict__ data, int64_t n_vec, uint32_t* __restrict__ sink) {
    // Tiles of kProbeBlock * kProbePerThread float4 are dealt round-robin over the
    // workgroups (one per CU): the fastest plain-read configuration found on the
    // box (tools/pattern_probe.hip: 76-77 us for 480 MB, 6.2-6.3 TB/s).
    constexpr int kProbePerThread = 3;
    uint32_t acc = 0;
    const int64_t tile = static_cast<int64_t>(kProbeBlock) * kProbePerThread;
    for (int64_t t0 = static_cast<int64_t>(blockIdx.x) * tile; t0 < n_vec; t0 += static_cast<int64_t>(gridDim.x) * tile) {
        float4 v[kProbePerThread];
#pragma unroll
        for (int u = 0; u < kProbePerThread; ++u) {
            const int64_t i = t0 + u * kProbeBlock + threadIdx.x;
            v[u] = data[i < n_vec ? i : n_vec - 1];
        }
#pragma unroll
        for (int u = 0; u < kProbePerThread; ++u)
            acc ^= __float_as_uint(v[u].x) ^ __float_as_uint(v[u].y) ^ __float_as_uint(v[u].z) ^ __float_as_uint(v[u].w);
    }
    // one word per workgroup keeps the loads alive without a measurable store stream
    __shared__ uint32_t s_acc;
    if (threadIdx.x == 0) s_acc = 0;
    __syncthreads();
    atomicXor(&s_acc, acc);
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = s_acc;
}

}  // namespace mi355
