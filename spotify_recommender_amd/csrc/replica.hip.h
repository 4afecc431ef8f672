// replica.hip.h — the fp16 REPLICA of the catalogue and the single-query scan over it.
//
// The reference streams the N x 12 fp32 matrix once per query (Recommender.cu:184-254,
// 48 B per row).  The contract per query is recommendByIndex's (Recommender.cu:275-318):
// the exact fp32 cosine of calculateSimilaritiesCPU (:256-273) and the best topN rows.
// A row can only be in that top-N if its score reaches the N-th best score T; almost no
// row does.  So the engine keeps, next to the fp32 rows, a second copy that is only good
// enough to RULE ROWS OUT: every row L2-normalised in fp32 and rounded to fp16 (24 B per
// row, built once when the handle is created).  A query reads the 24 B replica; the few
// rows per workgroup the replica cannot rule out are fetched from the fp32 matrix and
// scored with the exact chain (cosine_score(), kernels.hip.h).  Every key that leaves the
// kernel is therefore computed from the fp32 row by the reference's arithmetic: results
// are bit-identical to the fp32 scan (scan_kernel) and to the oracle, at half the bytes.
//
// Bound (the batched path's, batched.hip.h "Margin"): for a VALID row and a VALID query
//   |approx - exact| <= kHalfMargin,   approx = sum_j fp16(r^_j) * fp16(q^_j)
// (fp16 products are exact in fp32, accumulated by fp32 FMAs).  kHalfMargin is the
// flush-proof 1.5e-3, so nothing here depends on how a unit treats fp16 subnormals.
// valid row:   |row|^2 in [kBqMinNorm2, kBqMaxNorm2]  (then |row||q| > 1e-8: no zero branch,
//              and no fp32 sum overflows in any order)
// zero row:    fma-summed |row|^2 == 0 -> the reference's |row|^2 is 0 as well, its score is
//              exactly 0 against every valid query: the replica stores zeros (approx = 0).
// other rows:  (tiny, huge, inf, NaN) the replica stores fp16 NaN: approx = NaN, and
//              !(NaN < cutoff) sends the row to the exact chain every time.
// valid query: |q| in [kBqMinNorm, kBqMaxNorm]; otherwise the pre-filter stays off for the
//              whole launch (every row is fetched and scored exactly: slow, still right).
//
// Threshold.  Per-workgroup thresholds alone (a workgroup sees n / 512 rows) would send
// ~k ln(rows/k) rows per workgroup to the exact chain, each a random 48 B fetch.  So one
// small launch first looks at a spread sample of the replica (seed_half_kernel: one approx
// maximum per 128-row wave tile, excluded row masked).  With v the topN-th largest of those
// maxima, topN distinct rows have approx >= v, hence exact >= v - margin, hence the
// catalogue's topN-th best is >= v - margin and every row of the true top-N has
//   approx >= v - 2 margin  =: the launch-wide cutoff.
// Each scanning workgroup selects v itself from the <= 2048 sample maxima (one radix select
// while its first tiles are in flight).  Workgroup-local thresholds tighten the cutoff
// further whenever a local list fills.
#pragma once

#include "batched.hip.h"

namespace mi355 {

constexpr float kHalfMargin = kBqMarginFlush;   // what a launch claims unless the device check below allows the tighter bound
constexpr uint32_t kHalfNaN2 = kBqNaN2;        // two fp16 quiet NaNs: "always score this row exactly"
// (the sampling geometry — kHalfSeedBlock, kHalfSeedWaves, kHalfSeedMaxGrid — and the hand-off types live in handoff.hip.h)

// ---- building the replica ---------------------------------------------------------
// One thread per row; rows [n, n_padded) (n_padded even) are padding and hold NaN.
__global__ __launch_bounds__(256) void replica_build_kernel(const float* __restrict__ feats, int64_t n, int64_t n_padded,
                                                            uint2* __restrict__ half) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= n_padded) return;
    uint32_t p0 = kHalfNaN2, p1 = kHalfNaN2, p2 = kHalfNaN2, p3 = kHalfNaN2, p4 = kHalfNaN2, p5 = kHalfNaN2;
    if (row < n) {
        const float4* p = reinterpret_cast<const float4*>(feats) + row * 3;
        const float4 a = p[0], b = p[1], c = p[2];
        // the same normalisation as the batched passes (bq_pass_kernel), so one bound covers both
        float tot = a.x * a.x;
        tot = __builtin_fmaf(a.y, a.y, tot);
        tot = __builtin_fmaf(a.z, a.z, tot);
        tot = __builtin_fmaf(a.w, a.w, tot);
        tot = __builtin_fmaf(b.x, b.x, tot);
        tot = __builtin_fmaf(b.y, b.y, tot);
        tot = __builtin_fmaf(b.z, b.z, tot);
        tot = __builtin_fmaf(b.w, b.w, tot);
        tot = __builtin_fmaf(c.x, c.x, tot);
        tot = __builtin_fmaf(c.y, c.y, tot);
        tot = __builtin_fmaf(c.z, c.z, tot);
        tot = __builtin_fmaf(c.w, c.w, tot);
        const bool valid = tot >= kBqMinNorm2 && tot <= kBqMaxNorm2;
        if (valid) {
            const float inv = __builtin_amdgcn_rsqf(tot);
            p0 = bq_pack_h2(a.x * inv, a.y * inv); p1 = bq_pack_h2(a.z * inv, a.w * inv);
            p2 = bq_pack_h2(b.x * inv, b.y * inv); p3 = bq_pack_h2(b.z * inv, b.w * inv);
            p4 = bq_pack_h2(c.x * inv, c.y * inv); p5 = bq_pack_h2(c.z * inv, c.w * inv);
        } else if (tot == 0.0f) {
            p0 = p1 = p2 = p3 = p4 = p5 = 0u;
        }
    }
    uint2* dst = half + row * 3;
    dst[0] = make_uint2(p0, p1);
    dst[1] = make_uint2(p2, p3);
    dst[2] = make_uint2(p4, p5);
}

// ---- the query in fp16 --------------------------------------------------------------
struct HalfQuery {
    uint32_t h[6];   // fp16 pairs of q / |q| (wave-uniform: scalar registers)
    bool ok;         // the bound may be claimed for this query
};

__device__ __forceinline__ HalfQuery half_query(const float (&q)[kDim], float qn) {
    HalfQuery hq;
    hq.ok = qn >= kBqMinNorm && qn <= kBqMaxNorm;   // false for NaN
    const float inv = hq.ok ? 1.0f / qn : 0.0f;
#pragma unroll
    for (int p = 0; p < 6; ++p)
        hq.h[p] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(
            static_cast<int>(hq.ok ? bq_pack_h2(q[2 * p] * inv, q[2 * p + 1] * inv) : 0u)));
    return hq;
}

// sum_j fp16 * fp16 with fp32 FMAs (v_fma_mix_f32: the fp16 operands are converted exactly)
__device__ __forceinline__ float half_dot(const uint32_t (&qh)[6], uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3,
                                          uint32_t a4, uint32_t a5) {
    const uint32_t a[6] = {a0, a1, a2, a3, a4, a5};
    float acc = 0.0f;
#pragma unroll
    for (int p = 0; p < 6; ++p) {
        const bq_h2 x = __builtin_bit_cast(bq_h2, a[p]);
        const bq_h2 y = __builtin_bit_cast(bq_h2, qh[p]);
        acc = __builtin_fmaf(static_cast<float>(x[0]), static_cast<float>(y[0]), acc);
        acc = __builtin_fmaf(static_cast<float>(x[1]), static_cast<float>(y[1]), acc);
    }
    return acc;
}

// ---- does this device keep fp16 subnormals where the replica's arithmetic meets them? ---------------
// The bound is 1.0e-3 (kBqMargin) when they are kept and 1.5e-3 (kBqMarginFlush) when a unit flushes them
// (batched.hip.h, "Margin").  Three places matter: v_cvt_pk_f16_f32 when the replica is built (out[1]: fp16(3e-6)
// converted back), v_fma_mix_f32's fp16 operands in the single-query scan (out[2], out[3]: a subnormal on the row
// side, on the query side, times 1.0) and the matrix core in the multi-query pass (out[0], as bq_selfcheck_kernel).
// One wave; the host compares with the exact values.
__global__ void half_selfcheck_kernel(float* out) {
    const int lane = threadIdx.x;
    bq_h8 A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) {
        A[0] = __builtin_bit_cast(_Float16, static_cast<unsigned short>(0x0010));   // 2^-20, an fp16 subnormal
        B[0] = __builtin_bit_cast(_Float16, static_cast<unsigned short>(0x3c00));   // 1.0
    }
    const bq_f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const bq_f16v D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero, 0, 0, 0);
    const uint32_t packed = bq_pack_h2(3.0e-6f * (1.0f + static_cast<float>(lane)), 0.0f);   // lane 0: 3e-6
    // opaque operands: the compiler must not fold the products
    uint32_t sub = 0x00000010u, one = 0x00003c00u;
    asm volatile("" : "+v"(sub), "+v"(one));
    const uint32_t qs[6] = {one, 0u, 0u, 0u, 0u, 0u};
    const uint32_t qz[6] = {sub, 0u, 0u, 0u, 0u, 0u};
    const float row_side = half_dot(qs, sub, 0u, 0u, 0u, 0u, 0u);
    const float query_side = half_dot(qz, one, 0u, 0u, 0u, 0u, 0u);
    if (lane == 0) {
        out[0] = D[0];
        out[1] = static_cast<float>(__builtin_bit_cast(bq_h2, packed)[0]);
        out[2] = row_side;
        out[3] = query_side;
    }
}

#ifdef MI355REC_EXPERIMENTS   // the single-query scan over THIS replica is an A/B route of tools builds (single queries stream
                              // the 8-bit replica, replica_q8.hip.h); the product keeps the replica as the operand store of the
                              // matrix-core passes (batched.hip.h, replica_multi.hip.h)
// ---- the sample that seeds the launch-wide cutoff -------------------------------------
// A REGION is 1024 rows from row g * stride_rows on (stride_rows even and >= 1024, so no row is in
// two regions): one ordered-u32 approx maximum per 128-row wave tile (0 = nothing usable) goes to
// seed_vals[g * 8 + wave].  seed_region_load / seed_region_finish are the two halves of that, so a
// caller can have the loads of several regions in flight before it reduces the first.
struct SeedRegion {
    uint4 t0, t1, t2;
    int64_t pair;
    bool have;
};

__device__ __forceinline__ SeedRegion seed_region_load(const uint4* __restrict__ half, int64_t n_pairs, int64_t stride_rows,
                                                       int64_t g) {
    SeedRegion s;
    s.pair = ((g * stride_rows) >> 1) + threadIdx.x;
    s.have = s.pair < n_pairs;
    s.pair = s.have ? s.pair : n_pairs - 1;
    const uint4* p = half + s.pair * 3;
    s.t0 = p[0];
    s.t1 = p[1];
    s.t2 = p[2];
    return s;
}

__device__ __forceinline__ void seed_region_finish(const SeedRegion& s, const HalfQuery& hq, int64_t n, int64_t row_base,
                                                   int64_t exclude_global, uint32_t* __restrict__ seed_vals, int64_t g) {
    const float a0 = half_dot(hq.h, s.t0.x, s.t0.y, s.t0.z, s.t0.w, s.t1.x, s.t1.y);
    const float a1 = half_dot(hq.h, s.t1.z, s.t1.w, s.t2.x, s.t2.y, s.t2.z, s.t2.w);
    const int64_t r0 = s.pair * 2;
    // NaN compares false: special rows never seed
    const bool use0 = s.have && hq.ok && r0 < n && row_base + r0 != exclude_global && a0 >= -2.0f;
    const bool use1 = s.have && hq.ok && r0 + 1 < n && row_base + r0 + 1 != exclude_global && a1 >= -2.0f;
    uint32_t v = 0u;
    if (use0) v = score_to_ordered(a0);
    if (use1) {
        const uint32_t w = score_to_ordered(a1);
        v = w > v ? w : v;
    }
    v = wave_max_u32(v);
    if ((threadIdx.x & 63) == 0) seed_vals[g * kHalfSeedWaves + (threadIdx.x >> 6)] = v;
}

// One workgroup per region (single queries; the first query of a stream).
template <bool kQueryFromRow>
__global__ __launch_bounds__(kHalfSeedBlock) void seed_half_kernel(
    const float* __restrict__ feats, const uint4* __restrict__ half, int64_t n, int64_t stride_rows, int64_t row_base,
    QueryArg qarg, const float* __restrict__ query_ptr, int64_t exclude_global, uint32_t* __restrict__ seed_vals) {
    float q[kDim];
    if constexpr (kQueryFromRow) {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = query_ptr[j];
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = qarg.q[j];
    }
    const HalfQuery hq = half_query(q, query_norm(q));
    const SeedRegion s = seed_region_load(half, (n + 1) >> 1, stride_rows, blockIdx.x);
    seed_region_finish(s, hq, n, row_base, exclude_global, seed_vals, blockIdx.x);
}

// In a STREAM of queries the sample of query k + 1 is taken by a few workgroups of query k's scan
// launch instead of a launch of its own (scan_half_kernel<.., kWithMerge>: after the scanners and the
// merger come next.n_wgs "seed riders", handoff.hip.h).  Each walks its share of the regions, four loads in
// flight at a time.
__device__ __forceinline__ void seed_rider(const float* __restrict__ feats, const uint4* __restrict__ half, int64_t n,
                                           int64_t row_base, const NextSeed& next, int rider) {
    float q[kDim];
    if (next.query_ptr) {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = next.query_ptr[j];
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = next.q[j];
    }
    const HalfQuery hq = half_query(q, query_norm(q));
    const int64_t n_pairs = (n + 1) >> 1;
    constexpr int kAhead = 4;
    for (int g0 = rider; g0 < next.regions; g0 += kAhead * next.n_wgs) {
        SeedRegion s[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int g = g0 + u * next.n_wgs;
            s[u] = seed_region_load(half, n_pairs, next.stride_rows, g < next.regions ? g : rider);
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int g = g0 + u * next.n_wgs;
            if (g < next.regions)   // uniform
                seed_region_finish(s[u], hq, n, row_base, next.exclude_global, static_cast<uint32_t*>(next.out), g);
        }
    }
}

#endif   // MI355REC_EXPERIMENTS

// ---- the scan --------------------------------------------------------------------------
// One lane = one PAIR of rows = 48 B of replica (3 x dwordx4, the fp32 scan's own load
// pattern: every wave-level load covers a contiguous 3 KiB span).  Tiles of 2 * kBlock rows
// are dealt round-robin over the scanning workgroups.
template <int kBlockT, int kMinWavesT, int kDepthT>
struct HalfCfg {
    static constexpr int kBlock = kBlockT;
    static constexpr int kMinWaves = kMinWavesT;
    static constexpr int kDepth = kDepthT;
    static constexpr int kTileRows = 2 * kBlockT;
    static constexpr int kCandCap = kCandLimit + kTileRows;
    static constexpr int kCandPerThread = (kCandCap + kBlockT - 1) / kBlockT;
};
#ifndef MI355_HALF_DEPTH
#define MI355_HALF_DEPTH 2   // tiles in flight per lane (measured: 2, 3 and 4 are equally fast)
#endif
using DefaultHalfCfg = HalfCfg<512, 4, MI355_HALF_DEPTH>;

template <typename Cfg>
struct HalfScanSmemT {
    uint64_t cand[Cfg::kCandCap];
    SelectSmem sel;
    int count;
    int seeds;
    int rescored;
};
// The riding merger of the replica scan keeps 4096 survivors (the fp32 scan's keeps 2048: its kernel
// has neither the registers nor, at 3 workgroups per CU, the LDS for more).  With ~500 lists and topN
// near 1000 the merge probes every list 4-5 keys deep and ~2.2 topN keys survive the first cut: a
// 2048-slot buffer overflowed into the exact radix select over all keys in global memory (1 ms).
constexpr int kHalfRideSurvCap = 4096;
template <typename Cfg>
union HalfScanOrMergeSmem {
    HalfScanSmemT<Cfg> scan;
    MergeSmemT<Cfg::kBlock, kRideMaxLists, kHalfRideSurvCap> merge;
};

struct HalfTile {
    uint4 t0, t1, t2;
};

#ifdef MI355REC_EXPERIMENTS
// kWithMerge (streamed queries): workgroups [0, S) scan, workgroup S merges the PREVIOUS streamed
// query (as scan_kernel's riding merger), workgroups (S, gridDim) are seed riders for the NEXT one;
// S = gridDim.x - 1 - next.n_wgs.
template <typename Cfg, bool kQueryFromRow, bool kWithMerge>
__global__ __launch_bounds__(Cfg::kBlock, Cfg::kMinWaves) void scan_half_kernel(
    const float* __restrict__ feats, const uint4* __restrict__ half, int64_t n, int iters, int64_t row_base,
    QueryArg qarg, const float* __restrict__ query_ptr, int64_t exclude_global, int topk, uint64_t* __restrict__ block_lists,
    const uint32_t* __restrict__ seed_vals, int n_seed, unsigned long long* __restrict__ rescored /* [scanning workgroups] */,
    PrevMerge prev, NextSeed next) {
    constexpr int kBlock = Cfg::kBlock;
    __shared__ typename std::conditional<kWithMerge, HalfScanOrMergeSmem<Cfg>, HalfScanSmemT<Cfg>>::type s_mem;
    HalfScanSmemT<Cfg>* sm;
    unsigned nblocks = gridDim.x;   // scanning workgroups
    if constexpr (kWithMerge) {
        nblocks = gridDim.x - 1u - static_cast<unsigned>(next.n_wgs);
        if (blockIdx.x >= nblocks) {
            if (blockIdx.x == nblocks) {
                if (prev.lists)
                    merge_body(s_mem.merge, prev.lists, prev.n_lists, prev.topk, static_cast<int64_t>(prev.topk),
                               static_cast<int64_t>(0), prev.topk, prev.out_keys, static_cast<int64_t*>(nullptr),
                               static_cast<float*>(nullptr), static_cast<int64_t>(0), static_cast<int64_t>(0),
                               static_cast<int64_t>(0));
            } else {
                seed_rider(feats, half, n, row_base, next, static_cast<int>(blockIdx.x - nblocks - 1u));
            }
            return;
        }
        sm = &s_mem.scan;
    } else {
        (void)next;
        sm = &s_mem;
    }
    uint64_t* const s_cand = sm->cand;
    SelectSmem& s_sel = sm->sel;
    int& s_count = sm->count;

    const unsigned bid = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    float q[kDim];
    if constexpr (kQueryFromRow) {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = query_ptr[j];  // wave-uniform: scalar loads
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = qarg.q[j];
    }
    const float qn = query_norm(q);
    const HalfQuery hq = half_query(q, qn);

    const int64_t n_pairs = (n + 1) >> 1;
    const int64_t last_pair = n_pairs - 1;
    const int64_t pair_begin = static_cast<int64_t>(bid) * kBlock + tid;
    const int64_t pair_stride = static_cast<int64_t>(nblocks) * kBlock;

    auto load_tile = [&](HalfTile& dst, int it) {
        int64_t pair = pair_begin + static_cast<int64_t>(it) * pair_stride;
        pair = pair < n_pairs ? pair : last_pair;   // unconditional prefetch (see scan_kernel)
        const uint4* p = half + pair * 3;
        dst.t0 = p[0];
        dst.t1 = p[1];
        dst.t2 = p[2];
    };

    constexpr int kDepth = Cfg::kDepth;
    HalfTile ring[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d) load_tile(ring[d], d);

    // ---- launch-wide cutoff from the sample maxima (while the first tiles are in flight)
    if (tid == 0) {
        s_count = 0;
        sm->seeds = 0;
        sm->rescored = 0;
    }
    int n_rescored = 0;   // wave-uniform: rows this wave sent to the exact chain (diagnostics)
    __syncthreads();
    const float neg_inf = -__builtin_inff();
    float cutoff = neg_inf;   // -inf: everything is fetched and scored exactly
    if (hq.ok && n_seed > 0) {   // uniform
        uint64_t mine[kHalfSeedPerThread];
        int have = 0;
#pragma unroll
        for (int r = 0; r < kHalfSeedPerThread; ++r) {
            const int i = tid + r * kBlock;
            const uint32_t v = i < n_seed ? seed_vals[i] : 0u;
            mine[r] = v ? (static_cast<uint64_t>(v) << 32) | static_cast<uint32_t>(i + 1) : 0ull;
            have += v != 0u;
        }
        {   // one LDS atomic per wave, not per thread (replica_q8.hip.h: 2 us of serialised atomics otherwise)
            const int wave_have = __builtin_amdgcn_readlane(wave_inclusive_scan(have), 63);
            if ((tid & 63) == 0 && wave_have) atomicAdd(&sm->seeds, wave_have);
        }
        __syncthreads();
        if (sm->seeds >= topk) {   // uniform
            const uint64_t t = block_select_threshold<kBlock, kHalfSeedPerThread>(mine, topk, true, 0, s_sel);
            const float v = ordered_to_score(static_cast<uint32_t>(t >> 32));
            cutoff = v - 2.0f * qarg.margin - kBqSlack;
        }
    }
    uint64_t thr = 0;
    int compact_at = 2 * topk > 256 ? 2 * topk : 256;
    if (compact_at > kCandLimit) compact_at = kCandLimit;

    // A row the replica cannot rule out costs one random 48 B fetch from the fp32 matrix.  The
    // fetch is ISSUED when the row is found and CONSUMED one tile later (a one-entry pending slot
    // per lane, the same distance as the replica prefetch), so its latency overlaps the next
    // tile instead of stalling the wave.  A lane whose two rows both qualify scores the second
    // one on the spot (rare outside the first tiles of a launch without a seed).
    Row pend;
    pend.a = pend.b = pend.c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    int64_t pend_r = 0;
    bool pend_on = false;

    auto append = [&](bool have, const Row& row, int64_t r) {
        const float s = cosine_score(q, qn, row);
        const int64_t g = row_base + r;
        uint64_t key = pack_key(s, static_cast<uint32_t>(g));
        if (g == exclude_global) key = 0;
        const bool pass = have && key > thr;
        const uint64_t ballot = __ballot(pass);
        if (ballot) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_count, __popcll(ballot));
            base = __builtin_amdgcn_readfirstlane(base);
            const int pos = base + lanes_below(ballot);
            if (pass) s_cand[pos] = key;
        }
    };
    auto consume = [&]() {
        if (__ballot(pend_on)) append(pend_on, pend, pend_r);
        pend_on = false;
    };

    auto process_tile = [&](const HalfTile& t, int it) {
        const int64_t pair = pair_begin + static_cast<int64_t>(it) * pair_stride;
        const float a0 = half_dot(hq.h, t.t0.x, t.t0.y, t.t0.z, t.t0.w, t.t1.x, t.t1.y);
        const float a1 = half_dot(hq.h, t.t1.z, t.t1.w, t.t2.x, t.t2.y, t.t2.z, t.t2.w);
        const int64_t r0 = pair * 2;
        const bool maybe0 = pair < n_pairs && !(a0 < cutoff);              // r0 < n whenever the pair exists
        const bool maybe1 = pair < n_pairs && r0 + 1 < n && !(a1 < cutoff);
        consume();   // the rows fetched while the previous tile was scanned
        const bool some = maybe0 || maybe1;
        const uint64_t any = __ballot(some);
        if (any) {
            const bool both = maybe0 && maybe1;
            const uint64_t two = __ballot(both);
            n_rescored += __popcll(any) + __popcll(two);
            // lanes without a candidate re-read row 0: one cached line
            pend_r = maybe0 ? r0 : r0 + 1;
            pend = load_row(feats, some ? pend_r : static_cast<int64_t>(0));
            pend_on = some;
            if (two) {
                const Row second = load_row(feats, both ? r0 + 1 : static_cast<int64_t>(0));
                append(both, second, r0 + 1);
            }
        }
        __syncthreads();
        const int c = s_count;
        __syncthreads();
        if (c >= compact_at) {
            const uint64_t local_thr = compact_candidates<kBlock, Cfg::kCandPerThread>(s_cand, &s_count, topk, false, s_sel);
            if (local_thr > thr) {
                thr = local_thr;
                if (hq.ok) {
                    const float local_cut = ordered_to_score(static_cast<uint32_t>(thr >> 32)) - qarg.margin - kBqSlack;
                    cutoff = local_cut > cutoff ? local_cut : cutoff;
                }
            }
        }
    };

    for (int it = 0; it < iters; it += kDepth) {
#pragma unroll
        for (int sidx = 0; sidx < kDepth; ++sidx) {
            load_tile(ring[(sidx + kDepth - 1) % kDepth], it + sidx + kDepth - 1);
            if (it + sidx < iters) process_tile(ring[sidx], it + sidx);  // uniform
        }
    }
    consume();   // the last tile's fetches

    if (lane == 0 && n_rescored) atomicAdd(&sm->rescored, n_rescored);
    __syncthreads();
    if (tid == 0) rescored[bid] += static_cast<unsigned long long>(sm->rescored);   // launches of a handle are stream-ordered
    // (from kRankCountMax keys up the ranking is a bitonic sort of the next power of two — 45 barrier stages, ~2.7 us
    // for the ~250 keys a scan without a launch-wide cutoff ends with; an INEXACT cut to topk + topk / 4 first costs a
    // radix pass or two and leaves a set the counting rank handles)
    if (s_count > kRankCountMax && s_count > topk)  // uniform
        compact_candidates<kBlock, Cfg::kCandPerThread>(s_cand, &s_count, topk, false, s_sel);
    __syncthreads();
    block_rank_and_store<kBlock>(s_cand, s_count, block_lists + static_cast<int64_t>(bid) * topk, topk);
}

#endif   // MI355REC_EXPERIMENTS

}  // namespace mi355
