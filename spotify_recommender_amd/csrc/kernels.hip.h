// kernels.hip.h — device code of the MI355X cosine top-N engine (gfx950 only).
//
// One streaming pass over the row-major N x 12 fp32 catalogue replaces the
// reference's three device passes + host heap:
//   cublasSgemv            Recommender.cu:217-223   (dot products)
//   computeNormsKernel     Recommender.cu:48-59     (row norms, 2nd matrix read)
//   normalizeSimilarities  Recommender.cu:62-77     (divide / threshold / clamp)
//   host heap top-N        Recommender.cu:293-315
// Arithmetic follows the reference's CPU path bit for bit
// (calculateSimilaritiesCPU, Recommender.cu:256-273): sequential j = 0..11,
// multiply and add rounded separately (contraction is OFF for this file),
// correctly rounded sqrtf and '/', threshold 1e-8f, std::min/std::max clamp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#pragma clang fp contract(off)

namespace mi355 {

#ifdef MI355REC_PHASE_CLOCK   // tools/ builds only (tools/phase_clock.sh): 100 MHz wall-clock stamps, [workgroup][phase]
__device__ unsigned long long g_phase_clock[1024 * 8];
// the phases of ONE merge (merge_body in the workgroup with blockIdx.x == 0: merge_notify_kernel), kept in row 1023
#define MI355REC_MPHASE(i)                                                                       \
    do {                                                                                         \
        if (threadIdx.x == 0 && blockIdx.x == 0) g_phase_clock[1023 * 8 + (i)] = wall_clock64(); \
    } while (0)
// the phases of every workgroup of the fp32 scan (row = blockIdx.x, as the 8-bit scan's MI355REC_PHASE)
#define MI355REC_KPHASE(i)                                                                  \
    do {                                                                                    \
        if (threadIdx.x == 0 && blockIdx.x < 1023) g_phase_clock[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MI355REC_MPHASE(i) \
    do {                   \
    } while (0)
#define MI355REC_KPHASE(i) \
    do {                   \
    } while (0)
#endif

constexpr int kDim = 12;              // Song.h:12
constexpr int kMaxTopK = 1024;        // MI355REC_MAX_TOPN_FAST
constexpr int kCandLimit = 2 * kMaxTopK;  // a tile is never entered with more candidates

// Geometry of the streaming scan: threads per workgroup, rows in flight per lane
// per tile, and the minimum waves per SIMD the register allocator must leave
// room for (__launch_bounds__'s second argument).
template <int kBlockT, int kRowsT, int kMinWavesT, int kDepthT = 2>
struct ScanCfg {
    static constexpr int kBlock = kBlockT;
    static constexpr int kRowsPerThread = kRowsT;
    static constexpr int kMinWaves = kMinWavesT;
    static constexpr int kDepth = kDepthT;   // tiles in flight per lane (register ring)
    static constexpr int kTileRows = kBlockT * kRowsT;
    static constexpr int kCandCap = kCandLimit + kTileRows;  // LDS candidate slots
    static constexpr int kCandPerThread = (kCandCap + kBlockT - 1) / kBlockT;
};
using DefaultScanCfg = ScanCfg<512, 1, 6>;
constexpr int kProbeBlock = 512;      // stream_probe_kernel
constexpr int kMergeBlock = 1024;
constexpr int kMergeMaxLists = 2048;
constexpr int kMergeSurvCap = 4096;
constexpr int kMergeChunk = 16;       // keys probed per list per deeper merge round
constexpr int kMergeFirst = 4;        // keys of every list loaded up front (many-lists case)
constexpr int kMergeFirstPerThread = 8;  // covers kMergeMaxLists * kMergeFirst keys
constexpr int kMergeSurvPerThread = kMergeSurvCap / kMergeBlock;
constexpr int kMergeHeadsPerThread = kMergeMaxLists / kMergeBlock;


struct QueryArg {
    float q[kDim];
    float margin;   // error bound of the fp16 pre-filter the launch may claim (replica scans only; set by the host)
};

// ---- packed keys -----------------------------------------------------------

__host__ __device__ inline uint32_t score_to_ordered(float s) {
    s = s + 0.0f;  // -0.0f -> +0.0f: float-equal scores get equal images
    union { float f; uint32_t u; } c;
    c.f = s;
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}

__host__ __device__ inline float ordered_to_score(uint32_t o) {
    union { float f; uint32_t u; } c;
    c.u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return c.f;
}

__host__ __device__ inline uint64_t pack_key(float s, uint32_t global_row) {
    return (static_cast<uint64_t>(score_to_ordered(s)) << 32) |
           static_cast<uint64_t>(~global_row);
}

// ---- the score of one row ----------------------------------------------------

struct Row {
    float4 a, b, c;
};

__device__ __forceinline__ float query_norm(const float (&q)[kDim]) {
    float qn = 0.0f;  // Recommender.cu:259-261
#pragma unroll
    for (int j = 0; j < kDim; ++j) qn = qn + q[j] * q[j];
    return sqrtf(qn);
}

__device__ __forceinline__ float cosine_score(const float (&q)[kDim], float qn,
                                              const Row& r) {
    const float f[kDim] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y,
                           r.b.z, r.b.w, r.c.x, r.c.y, r.c.z, r.c.w};
    float dot = 0.0f;  // Recommender.cu:264-269
    float nrm = 0.0f;
#pragma unroll
    for (int j = 0; j < kDim; ++j) {
        dot = dot + q[j] * f[j];
        nrm = nrm + f[j] * f[j];
    }
    const float den = sqrtf(nrm) * qn;  // :270
    float s = 0.0f;
    if (den > 1e-8f) {                  // :271
        const float t = dot / den;
        const float m = (t < 1.0f) ? t : 1.0f;   // std::min(1.0f, t)
        s = (-1.0f < m) ? m : -1.0f;             // std::max(-1.0f, m)
    }
    return s;
}

// Cheap UPPER-BOUND test used only to skip rows that cannot beat the running
// threshold: the cosine evaluated with packed FMAs (even/odd partial sums) and
// v_rsq_f32.  Against the exactly rounded reference chain its error is
// <= ~2e-6 (12-term fp32 accumulation in a different order + 1-ulp rsq + two
// multiplies), so a row is skipped only when approx < threshold_score -
// kApproxMargin; everything else (NaN included) is re-scored with
// cosine_score().  Only used while the threshold score is > 0, where the
// reference's "den <= 1e-8 -> 0" rows can never qualify.
// The bound only holds while no fp32 sum overflows: in a different summation
// order an overflow can appear in one chain and not in the other (e.g. q =
// (1e19,1e19,1e19,0..), f = (3e19,3e19,-3e19,0..): the reference's sequential
// chain gives inf/inf = NaN -> clamped to 1.0, the paired chain a finite 0).
// So the pre-filter is trusted only for |row|^2 < kApproxMaxNorm2 (checked per
// row: anything else is re-scored exactly) and |q| < kApproxMaxQueryNorm
// (checked once per query: otherwise the pre-filter stays off); then
// |partial sums| <= |row||q| < 1e37 in every order.
constexpr float kApproxMargin = 8e-6f;
constexpr float kApproxMaxNorm2 = 1e37f;
constexpr float kApproxMaxQueryNorm = 3e18f;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float approx_cosine(const float (&q)[kDim], float inv_qn, const Row& r) {
    const float f[kDim] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y,
                           r.b.z, r.b.w, r.c.x, r.c.y, r.c.z, r.c.w};
    v2f d = {0.0f, 0.0f};
    v2f m = {0.0f, 0.0f};
#pragma unroll
    for (int p = 0; p < kDim / 2; ++p) {
        const v2f ff = {f[2 * p], f[2 * p + 1]};
        const v2f qq = {q[2 * p], q[2 * p + 1]};
        d = __builtin_elementwise_fma(ff, qq, d);
        m = __builtin_elementwise_fma(ff, ff, m);
    }
    const float nrm2 = m.x + m.y;
    const float a = (d.x + d.y) * __builtin_amdgcn_rsqf(nrm2) * inv_qn;
    return nrm2 < kApproxMaxNorm2 ? a : __builtin_nanf("");  // NaN = "cannot tell": the caller re-scores exactly
}

__device__ __forceinline__ Row load_row(const float* __restrict__ feats, int64_t row) {
    const float4* p = reinterpret_cast<const float4*>(feats + row * kDim);
    Row r;
    r.a = p[0];
    r.b = p[1];
    r.c = p[2];
    return r;
}

// Number of set bits of `mask` below this lane: v_mbcnt_lo + v_mbcnt_hi, no 64-bit
// (1 << lane) - 1 mask to keep in two VGPRs across the streaming loop.
__device__ __forceinline__ int lanes_below(uint64_t mask) {
    return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
                                                      __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u)));
}

// ---- workgroup-level selection ------------------------------------------------
// Ranking c candidates by counting costs ~c*c/32 LDS cycles (85 us at c = 2560),
// so thresholds come from an O(c) MSB-first radix select on LDS histograms and
// only the final <= topk survivors are ever ranked.

struct SelectSmem {
    int hist[256];
    unsigned int hi_max;
    unsigned int hi_min;
    int digit;
    int above;
    int in_bin;
    int pad;
};

// Inclusive prefix sum across the 64 lanes of a wave in DPP (no LDS traffic):
// four row_shr steps scan each row of 16 lanes, row_bcast:15 / row_bcast:31
// carry the row totals forward.
__device__ __forceinline__ int wave_inclusive_scan(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2,3
    return x;
}

// Wave-wide max / min of a 32-bit value in DPP; the result is returned in every lane.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    x = mx(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xf, 0xf, true)));
    x = mx(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xf, 0xf, true)));
    x = mx(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xf, 0xf, true)));
    x = mx(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xf, 0xf, true)));
    x = mx(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xa, 0xf, false)));
    x = mx(x, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xc, 0xf, false)));
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(x), 63));
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) { return ~wave_max_u32(~x); }

// Every thread of the workgroup calls this with its share of the keys in
// registers (0 = empty slot; keys are unique).  Precondition: at least `need`
// non-empty keys in total, need >= 1.  Returns T with |{key >= T}| >= need;
// with `exact` the count is exactly `need`, otherwise up to `slack` extra keys
// may remain (fewer passes).  Digits are taken relative to the smallest key so
// the first pass already separates the candidates.  Two barriers per pass:
// wave 0 scans the 256 bins (lane l owns bins 255-4l .. 252-4l, so a prefix
// scan over lanes is a suffix scan over bins) and clears them for the next pass.
template <int kThreads, int kPerThread>
__device__ inline uint64_t block_select_threshold(const uint64_t (&mine)[kPerThread], int need,
                                                  bool exact, int slack, SelectSmem& sm) {
    static_assert(kThreads >= 256, "histogram is cleared by the first 256 threads");
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    // range of the score halves (32-bit, one LDS atomic per wave); the digits
    // below are relative to base = smallest score << 32
    uint32_t mx = 0, mn = ~0u;
#pragma unroll
    for (int r = 0; r < kPerThread; ++r) {
        if (mine[r]) {
            const uint32_t hi = static_cast<uint32_t>(mine[r] >> 32);
            mx = hi > mx ? hi : mx;
            mn = hi < mn ? hi : mn;
        }
    }
    mx = wave_max_u32(mx);
    mn = wave_min_u32(mn);
    if (tid < 256) sm.hist[tid] = 0;
    if (tid == 0) {
        sm.hi_max = 0u;
        sm.hi_min = ~0u;
    }
    __syncthreads();
    if (lane == 0) {
        atomicMax(&sm.hi_max, mx);
        atomicMin(&sm.hi_min, mn);
    }
    __syncthreads();
    const uint64_t base = static_cast<uint64_t>(sm.hi_min) << 32;
    const uint64_t span = (static_cast<uint64_t>(sm.hi_max - sm.hi_min) << 32) | 0xffffffffull;
    int shift = span ? (64 - __clzll(static_cast<long long>(span))) - 8 : 0;
    if (shift < 0) shift = 0;
    uint64_t prefix = 0, mask = 0;

    for (;;) {
#pragma unroll
        for (int r = 0; r < kPerThread; ++r) {
            const uint64_t k = mine[r];
            if (k) {
                const uint64_t v = k - base;
                if ((v & mask) == prefix) atomicAdd(&sm.hist[static_cast<int>((v >> shift) & 255u)], 1);
            }
        }
        __syncthreads();
        if (tid < 64) {
            const int top = 255 - 4 * lane;  // this lane's highest bin
            const int h0 = sm.hist[top], h1 = sm.hist[top - 1], h2 = sm.hist[top - 2], h3 = sm.hist[top - 3];
            sm.hist[top] = 0; sm.hist[top - 1] = 0; sm.hist[top - 2] = 0; sm.hist[top - 3] = 0;
            const int lane_sum = h0 + h1 + h2 + h3;
            int cum = wave_inclusive_scan(lane_sum) - lane_sum;  // keys in bins above this lane's
            const int hs[4] = {h0, h1, h2, h3};
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (cum < need && cum + hs[b] >= need) {
                    sm.digit = top - b;
                    sm.above = cum;
                    sm.in_bin = hs[b];
                }
                cum += hs[b];
            }
        }
        __syncthreads();
        const int digit = sm.digit, above = sm.above, in_bin = sm.in_bin;
        prefix |= static_cast<uint64_t>(digit) << shift;
        mask |= 255ull << shift;
        need -= above;
        if (shift == 0 || in_bin == need || (!exact && in_bin - need <= slack)) break;
        shift = shift > 8 ? shift - 8 : 0;
    }
    return base + prefix;
}

// Writes the best min(c, topk) of the `c` unique keys in s_keys to dst in descending
// order, zero-filling up to dst[topk).  Small sets are ranked by counting (~c*c/32 LDS
// cycles); from kRankCountMax keys up the keys are sorted IN PLACE by a bitonic
// network in LDS (log2(P)*(log2(P)+1)/2 stages of one compare-exchange per thread
// pair: 55 stages for 1024 keys, ~1 us, where counting cost 12 us per workgroup at
// topN = 1000).  Needs room for the next power of two >= c in s_keys; every thread of
// the workgroup must call it (barriers inside).
constexpr int kRankDirectMax = 384;   // callers cut larger survivor sets to exactly topk first (O(c) select)
constexpr int kRankCountMax = 160;
// kCoherent: the list is stored THROUGH to device scope (another workgroup of the same launch will read it:
// lone_tail).
template <bool kCoherent>
__device__ __forceinline__ void st_key(uint64_t* p, uint64_t v) {
    if constexpr (kCoherent) {
        __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *p = v;
    }
}

template <int kThreads, bool kCoherent = false>
__device__ inline void block_rank_and_store(uint64_t* s_keys, int c, uint64_t* dst, int topk) {
    for (int i = threadIdx.x; i < topk; i += kThreads) {
        if (i >= c) st_key<kCoherent>(&dst[i], 0ull);
    }
    if (c > kRankCountMax) {   // uniform
        int p2 = 256;
        while (p2 < c) p2 <<= 1;
        for (int i = c + threadIdx.x; i < p2; i += kThreads) s_keys[i] = 0ull;   // empty keys sort last
        __syncthreads();
        for (int k = 2; k <= p2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < (p2 >> 1); t += kThreads) {
                    // pair (lo, lo + j) with bit j clear in lo; descending overall
                    const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int hi = lo | j;
                    const uint64_t a = s_keys[lo], b = s_keys[hi];
                    const bool down = (lo & k) == 0;   // this run sorts descending
                    if ((a < b) == down) {
                        s_keys[lo] = b;
                        s_keys[hi] = a;
                    }
                }
                __syncthreads();
            }
        }
        const int n_out = c < topk ? c : topk;
        for (int i = threadIdx.x; i < n_out; i += kThreads) st_key<kCoherent>(&dst[i], s_keys[i]);
        return;
    }
    for (int i = threadIdx.x; i < c; i += kThreads) {
        const uint64_t mine = s_keys[i];
        int rank = 0;
        int j = 0;
        for (; j + 8 <= c; j += 8) {
            const uint64_t k0 = s_keys[j], k1 = s_keys[j + 1], k2 = s_keys[j + 2], k3 = s_keys[j + 3];
            const uint64_t k4 = s_keys[j + 4], k5 = s_keys[j + 5], k6 = s_keys[j + 6], k7 = s_keys[j + 7];
            rank += (k0 > mine) + (k1 > mine) + (k2 > mine) + (k3 > mine) +
                    (k4 > mine) + (k5 > mine) + (k6 > mine) + (k7 > mine);
        }
        for (; j < c; ++j) rank += (s_keys[j] > mine);
        if (rank < topk) st_key<kCoherent>(&dst[rank], mine);
    }
}

// Shrinks s_cand[0..*s_count) to the keys >= T where T bounds the topk-th best
// (exactly topk keys remain with `exact`), and returns the filter threshold for
// the streaming loop: later keys pass iff key > return value.
template <int kThreads, int kCandPerThread>
__device__ inline uint64_t compact_candidates(uint64_t* s_cand, int* s_count, int topk,
                                              bool exact, SelectSmem& sm) {
    __syncthreads();
    const int c = *s_count;
    if (c <= topk) return 0ull;  // uniform: nothing to drop yet
    uint64_t mine[kCandPerThread];
#pragma unroll
    for (int r = 0; r < kCandPerThread; ++r) {
        const int i = threadIdx.x + r * kThreads;
        mine[r] = i < c ? s_cand[i] : 0ull;
    }
    int slack = topk / 4;
    if (slack < 16) slack = 16;
    const uint64_t t = block_select_threshold<kThreads, kCandPerThread>(mine, topk, exact, slack, sm);
    if (threadIdx.x == 0) *s_count = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kCandPerThread; ++r) {   // (uniform loop) one LDS atomic per wave: per-thread adds to the one word serialise
        const bool keep = mine[r] >= t;
        const uint64_t who = __ballot(keep);
        int base = 0;
        if ((threadIdx.x & 63) == 0 && who) base = atomicAdd(s_count, __popcll(who));
        base = __builtin_amdgcn_readfirstlane(base);
        if (keep) s_cand[base + lanes_below(who)] = mine[r];
    }
    __syncthreads();
    return t - 1ull;
}

// ---- merge of sorted candidate lists -----------------------------------------
// n_lists lists of list_len keys (each sorted descending, 0-padded) -> the best
// topk keys, sorted descending, 0-padded; optional unpack to (row, score).
// One workgroup per query (blockIdx.x = query in a batch): list l of query b starts
// at lists_base + b*lists_query_stride + l*list_stride, so both layouts work:
// [query][list][key] (per-workgroup lists of a scan) and [list][query][key]
// (per-rank results of a batch after the all-gather).
//
// A key can only be in the global top-k if it is >= T whenever m lists each
// hold >= j keys that are >= T with m*j >= topk.  With plenty of lists j = 1:
// T = the topk-th largest list HEAD; with few (8 per-rank lists) deeper probes.
// Lists are sorted, so each list's survivors are a prefix; for statistically
// similar shards ~1.2*topk keys survive in total.  They are cut to exactly
// topk by the radix select and ranked.  If more than kMergeSurvCap survive
// (adversarial input), an exact radix select over all keys in global memory
// finds the topk-th key instead.

// Lists written by OTHER workgroups of the SAME launch (lone_tail below) are read past this XCD's L2, with
// device-scope atomic loads; lists of an earlier launch with plain loads.
template <bool kCoherent>
__device__ __forceinline__ uint64_t ld_key(const uint64_t* p) {
    if constexpr (kCoherent) {
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        return *p;
    }
}

template <int kThreads, bool kCoherent = false>
__device__ inline uint64_t merge_global_radix_select(const uint64_t* lists,
                                                     int64_t total, int list_len, int64_t list_stride,
                                                     int topk, int* s_hist, int* s_pair) {
    // returns the topk-th largest key (0 if fewer than topk non-zero keys)
    uint64_t prefix = 0, mask = 0;
    int remaining = topk;
    for (int pass = 7; pass >= 0; --pass) {
        for (int i = threadIdx.x; i < 256; i += kThreads) s_hist[i] = 0;
        __syncthreads();
        const int shift = pass * 8;
        for (int64_t i = threadIdx.x; i < total; i += kThreads) {
            const uint64_t k = ld_key<kCoherent>(&lists[(i / list_len) * list_stride + (i % list_len)]);
            if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0, d = 255;
            for (; d > 0; --d) {
                if (acc + s_hist[d] >= remaining) break;
                acc += s_hist[d];
            }
            s_pair[0] = d;
            s_pair[1] = remaining - acc;
        }
        __syncthreads();
        prefix |= static_cast<uint64_t>(s_pair[0]) << shift;
        mask |= 255ull << shift;
        remaining = s_pair[1];
        __syncthreads();
    }
    return prefix;
}

// `slot` = which list set (lists_base + slot * lists_query_stride), `out_slot` = which
// output row (out_*_base + out_slot * out_query_stride).
// Shared memory of one merge: the product's merge kernels use <1024 threads, 2048 lists,
// 4096 survivors>; the merger that rides along in a scan launch (scan_kernel<.., kWithMerge>)
// uses the scan's block size and smaller bounds so that it fits the scan's register and LDS
// budget (an overflowing survivor set falls back to the exact radix select either way).
template <int kThreads, int kMaxLists, int kSurvCap>
struct MergeSmemT {
    uint64_t surv[kSurvCap];
    uint64_t top[kMaxTopK];
    SelectSmem sel;
    int pair[2];
    int count;
    int overflow;
    int more;
    unsigned short active[kMaxLists];
};

template <bool kCoherent = false, int kThreads, int kMaxLists, int kSurvCap>
__device__ __forceinline__ void merge_body(
    MergeSmemT<kThreads, kMaxLists, kSurvCap>& sm, const uint64_t* lists_base, int n_lists, int list_len, int64_t list_stride,
    int64_t lists_query_stride, int topk, uint64_t* __restrict__ out_keys_base,
    int64_t* __restrict__ out_idx_base, float* __restrict__ out_score_base,
    int64_t out_query_stride, int64_t slot, int64_t out_slot, int tid_in = -1 /* threadIdx.x, if the caller has a reason to
    pass it (a caller that merges in a LOOP passes an opaque copy, or what depends on it alone is hoisted out and spilled) */) {
    constexpr int kFirstPer = kMaxLists * kMergeFirst / kThreads;   // first-chunk keys per thread
    constexpr int kSurvPer = kSurvCap / kThreads;
    constexpr int kHeadsPer = kMaxLists / kThreads;
    static_assert(kThreads % kMergeFirst == 0 && kSurvCap % kThreads == 0 && kMaxLists % kThreads == 0, "even shares");
    uint64_t* const s_surv = sm.surv;
    uint64_t* const s_top = sm.top;
    SelectSmem& s_sel = sm.sel;
    int* const s_pair = sm.pair;
    int& s_count = sm.count;
    int& s_overflow = sm.overflow;
    int& s_more = sm.more;
    unsigned short* const s_active = sm.active;

    const int tid = tid_in >= 0 ? tid_in : static_cast<int>(threadIdx.x);
    const uint64_t* lists = lists_base + slot * lists_query_stride;
    uint64_t* out_keys = out_keys_base + out_slot * out_query_stride;

    MI355REC_MPHASE(0);
    if (tid == 0) {
        s_count = 0;
        s_overflow = 0;
        s_pair[0] = 0;
        s_more = 0;
    }
    __syncthreads();
    int probe = 1;
    if (n_lists < 2 * topk) probe = (2 * topk + n_lists - 1) / n_lists;
    if (probe > list_len) probe = list_len;
    const int need_lists = (topk + probe - 1) / probe;
    int slack = need_lists / 8;
    uint64_t thr = 1;  // accept every non-empty key
    int first = 0;     // keys [0, first) of every list are already dealt with

    const int64_t total_keys = static_cast<int64_t>(n_lists) * list_len;
    if (total_keys <= kSurvCap) {
        // Small input (e.g. one list of topn keys per rank after the all-gather):
        // take every key in one load phase; the select / rank below does the rest.
        // (one LDS atomic per wave instead of one per key; measured on the 326 top-10 lists of a 1 M-row scan: no
        // difference, 9.1 vs 9.3 us for the whole merge_notify_kernel — the merge is latency, not atomics)
        first = list_len;
        for (int64_t i0 = 0; i0 < total_keys; i0 += kThreads) {   // uniform trip count: every lane takes part in the ballot
            const int64_t i = i0 + tid;
            const uint64_t k = i < total_keys ? ld_key<kCoherent>(&lists[(i / list_len) * list_stride + (i % list_len)]) : 0ull;
            const uint64_t have = __ballot(k != 0ull);
            int base = 0;
            if ((tid & 63) == 0 && have) base = atomicAdd(&s_count, __popcll(have));
            base = __builtin_amdgcn_readfirstlane(base);
            if (k) s_surv[base + lanes_below(have)] = k;
        }
    } else if (probe == 1 && n_lists * kMergeFirst <= kThreads * kFirstPer) {
        // Many lists (the per-workgroup lists of one scan).  ONE load phase brings
        // in the first kMergeFirst keys of every list; the heads among them give
        // the threshold and the rest is filtered from registers, so the usual
        // case costs a single global-memory round trip.
        first = kMergeFirst < list_len ? kMergeFirst : list_len;
        uint64_t k[kFirstPer];
        uint64_t hk[kFirstPer];
        const int j = tid % kMergeFirst;  // kThreads % kMergeFirst == 0
        int local_nonzero = 0;
#pragma unroll
        for (int u = 0; u < kFirstPer; ++u) {
            const int l = (u * kThreads + tid) / kMergeFirst;
            k[u] = (l < n_lists && j < list_len) ? ld_key<kCoherent>(&lists[static_cast<int64_t>(l) * list_stride + j]) : 0ull;
        }
#pragma unroll
        for (int u = 0; u < kFirstPer; ++u) {
            hk[u] = j == 0 ? k[u] : 0ull;
            local_nonzero += hk[u] != 0ull;
        }
        for (int l = tid; l < n_lists; l += kThreads) s_active[l] = 0xffff;
        {   // one LDS atomic per wave (hundreds of threads adding to the one word serialise: measured 2 us in the 8-bit
            // scan's sample selection, the same pattern)
            const int wave_nonzero = __builtin_amdgcn_readlane(wave_inclusive_scan(local_nonzero), 63);
            if ((tid & 63) == 0 && wave_nonzero) atomicAdd(&s_pair[0], wave_nonzero);
        }
        __syncthreads();
        if (s_pair[0] >= need_lists)  // uniform
            thr = block_select_threshold<kThreads, kFirstPer>(hk, need_lists, false, slack, s_sel);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kFirstPer; ++u) {
            const bool pass = k[u] >= thr;
            const uint64_t who = __ballot(pass);   // (uniform loop: every lane takes part)
            int base = 0;
            if ((tid & 63) == 0 && who) base = atomicAdd(&s_count, __popcll(who));
            base = __builtin_amdgcn_readfirstlane(base);
            if (pass) {
                const int slot = base + lanes_below(who);
                if (slot < kSurvCap) s_surv[slot] = k[u];
                else s_overflow = 1;
                if (j == first - 1) {  // the whole first chunk passed: look deeper
                    s_active[(u * kThreads + tid) / kMergeFirst] = 0;
                    s_more = 1;
                }
            }
        }
    } else {
        // Few lists (e.g. one per rank) or very many: probe each list at depth
        // `probe` and start the rounds from the top of every list.
        uint64_t heads[kHeadsPer];
        int local_nonzero = 0;
#pragma unroll
        for (int r = 0; r < kHeadsPer; ++r) {
            const int l = tid + r * kThreads;
            heads[r] = l < n_lists ? ld_key<kCoherent>(&lists[static_cast<int64_t>(l) * list_stride + (probe - 1)]) : 0ull;
            local_nonzero += heads[r] != 0ull;
            if (l < n_lists) s_active[l] = 0;
        }
        if (tid == 0) s_more = 1;  // every list starts active
        if (local_nonzero) atomicAdd(&s_pair[0], local_nonzero);
        __syncthreads();
        if (s_pair[0] >= need_lists)  // uniform; the select needs >= need_lists non-empty probes
            thr = block_select_threshold<kThreads, kHeadsPer>(heads, need_lists, false, slack, s_sel);
    }
    __syncthreads();
    MI355REC_MPHASE(1);   // first chunk loaded, threshold selected, survivors appended

    // Deeper rounds: round d looks at keys [first + d*C, first + (d+1)*C) of every
    // list that is still active (its previous chunk passed entirely); the loads
    // of a round are independent and issued before any of them is consumed.
    // With the first-chunk phase above this loop usually does not run at all.
    for (int round = 0; first + round * kMergeChunk < list_len && !s_overflow; ++round) {
        // s_more was raised by whoever marked a list active for this round
        if (!s_more) break;  // uniform: read after a barrier, rewritten only after the next one
        __syncthreads();
        if (tid == 0) s_more = 0;
        __syncthreads();
        const int total = n_lists * kMergeChunk;
        for (int t0 = 0; t0 < total; t0 += kThreads * 8) {
            uint64_t k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * kThreads + tid;
                const int l = t / kMergeChunk;
                const int pos = first + round * kMergeChunk + (t % kMergeChunk);
                const bool live = t < total && pos < list_len && s_active[l] == round;
                k[u] = live ? ld_key<kCoherent>(&lists[static_cast<int64_t>(l) * list_stride + pos]) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k[u] >= thr) {
                    const int slot = atomicAdd(&s_count, 1);
                    if (slot < kSurvCap) s_surv[slot] = k[u];
                    else s_overflow = 1;
                }
            }
        }
        __syncthreads();
        // a list stays active iff the LAST key of this chunk passed
        for (int l = tid; l < n_lists; l += kThreads) {
            if (s_active[l] == round) {
                const int last = first + (round + 1) * kMergeChunk - 1;
                if (last < list_len && ld_key<kCoherent>(&lists[static_cast<int64_t>(l) * list_stride + last]) >= thr) {
                    s_active[l] = static_cast<unsigned short>(round + 1);
                    s_more = 1;
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (s_overflow) {
        // exact fallback: radix-select the topk-th key over everything
        const int64_t total = static_cast<int64_t>(n_lists) * list_len;
        uint64_t kth = merge_global_radix_select<kThreads, kCoherent>(lists, total, list_len, list_stride, topk, s_sel.hist, s_pair);
        if (kth == 0) kth = 1;
        if (tid == 0) s_count = 0;
        __syncthreads();
        for (int64_t i0 = 0; i0 < total; i0 += kThreads) {
            const int64_t i = i0 + tid;
            const uint64_t k = (i < total) ? ld_key<kCoherent>(&lists[(i / list_len) * list_stride + (i % list_len)]) : 0ull;
            if (k >= kth) {
                const int pos = atomicAdd(&s_count, 1);  // exactly topk keys when unique
                if (pos < kSurvCap) s_surv[pos] = k;
            }
        }
        __syncthreads();
    }

    MI355REC_MPHASE(2);   // deeper rounds done
    int c = s_count < kSurvCap ? s_count : kSurvCap;
    __syncthreads();
    if (c > topk && c > kRankDirectMax) {  // uniform: too many to rank, cut to exactly topk in O(c)
        uint64_t mine[kSurvPer];
#pragma unroll
        for (int r = 0; r < kSurvPer; ++r) {
            const int i = tid + r * kThreads;
            mine[r] = i < c ? s_surv[i] : 0ull;
        }
        // (not to EXACTLY topk: that takes the radix select through all its byte passes; a cut that may leave up to
        // kRankCountMax - topk keys more stops after one or two, and the ranking below keeps the best topk of what is left)
        const int cut_slack = kRankCountMax > topk ? kRankCountMax - topk : 0;
        const uint64_t t = block_select_threshold<kThreads, kSurvPer>(mine, topk, cut_slack == 0, cut_slack, s_sel);
        if (tid == 0) s_count = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kSurvPer; ++r) {   // (uniform loop) one LDS atomic per wave
            const bool keep = mine[r] >= t;
            const uint64_t who = __ballot(keep);
            int base = 0;
            if ((tid & 63) == 0 && who) base = atomicAdd(&s_count, __popcll(who));
            base = __builtin_amdgcn_readfirstlane(base);
            if (keep && base + lanes_below(who) < kSurvCap) s_surv[base + lanes_below(who)] = mine[r];
        }
        __syncthreads();
        c = s_count < kSurvCap ? s_count : kSurvCap;
    }
    MI355REC_MPHASE(3);   // final cut done
    block_rank_and_store<kThreads>(s_surv, c, s_top, topk);
    __syncthreads();
    MI355REC_MPHASE(4);   // ranked
    for (int i = tid; i < topk; i += kThreads) {
        const uint64_t k = s_top[i];
        out_keys[i] = k;
        if (out_idx_base) {
            out_idx_base[out_slot * out_query_stride + i] =
                k ? static_cast<int64_t>(static_cast<uint32_t>(~static_cast<uint32_t>(k))) : -1;
        }
        if (out_score_base) {
            out_score_base[out_slot * out_query_stride + i] =
                k ? ordered_to_score(static_cast<uint32_t>(k >> 32)) : 0.0f;
        }
    }
}

// ---- one launch per lone query --------------------------------------------------
// A caller that waits for ONE query on the host (mi355rec_query_row_topn, what Recommender::recommend sits on)
// pays three launches: sample, scan, merge.  With a LoneTail (the scan over the 8-bit replica on shards of
// >= 4 M rows; measured from C++: 46 us instead of 49 at 10 M rows, but 31 instead of 28 at 1 M, where the
// separate 1024-thread merge kernel beats the last workgroup of the scan) the scan is the last one: every workgroup
// stores its list through to device scope and counts itself out (two levels: eight group counters, then one, so
// that no counter sees more than ~100 arrivals); the workgroup that finds itself last merges all lists — read
// past its L2 — into the caller's buffers and, like merge_notify_kernel, raises the completion word the host
// polls.  No fences under the scanners (see scan_q8_kernel's seed riders for what those cost) and no spinning:
// every workgroup leaves after one atomic or two.
struct LoneTail {
    unsigned* counters;       // [9]: groups 0..7 (blockIdx % 8), then the count of finished groups; counted up across launches, never reset
    uint64_t* out_keys;
    int64_t* out_idx;         // may be device-visible pinned host memory
    float* out_score;
    uint32_t* done_word;      // null: no completion word
    uint32_t done_value;
    unsigned base[9];         // what each counter holds before this launch's arrivals (the host keeps the books: lone_tail_bases)
};

// The arrivals a launch of `grid` workgroups adds to counter g (g < 8), resp. to counter 8 (host and device agree on this).
__host__ __device__ inline unsigned lone_tail_groups(unsigned grid) { return grid < 8u ? grid : 8u; }
__host__ __device__ inline unsigned lone_tail_members(unsigned grid, unsigned g) {
    const unsigned groups = lone_tail_groups(grid);
    return g < groups ? (grid - g + groups - 1u) / groups : 0u;
}

// Every thread of every workgroup calls this after block_rank_and_store<.., true>; `s_flag` is any LDS word the
// caller can spare.  n_lists = gridDim.x lists of topk keys at `lists`.
// A workgroup is "last" only when the counter reaches exactly base + members: the counters are never reset, so no
// missed or repeated reset can make two workgroups (or none of this launch's own) believe they are the last one;
// a counter that does not add up leaves the completion word unwritten and the host call fails loudly (wait_done).
template <typename MergeSmem>
__device__ __forceinline__ void lone_tail(MergeSmem& msm, int* s_flag, const uint64_t* lists, int topk, const LoneTail& lt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through list stores have completed (a workgroup-scope fence does not wait for them)
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned groups = lone_tail_groups(gridDim.x);
        const unsigned g = blockIdx.x % groups;
        const unsigned members = lone_tail_members(gridDim.x, g);
        int last = 0;
        if (__hip_atomic_fetch_add(&lt.counters[g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == lt.base[g] + members)
            last = __hip_atomic_fetch_add(&lt.counters[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == lt.base[8] + groups;
        *s_flag = last;
    }
    __syncthreads();
    if (!*s_flag) return;   // uniform
    __syncthreads();          // (the flag may live in the union the merge is about to use)
    merge_body<true>(msm, lists, static_cast<int>(gridDim.x), topk, static_cast<int64_t>(topk), static_cast<int64_t>(0), topk,
                     lt.out_keys, lt.out_idx, lt.out_score, static_cast<int64_t>(0), static_cast<int64_t>(0),
                     static_cast<int64_t>(0));
    if (lt.done_word) {   // uniform
        // the waves that stored results order their stores before ... (the others have nothing to release: a system-scope
        // fence is an L2 write-back per wave, and sixteen of them queue up)
        if (static_cast<int>(threadIdx.x) < ((topk + 63) & ~63)) __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(lt.done_word, lt.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... the word
    }
}

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
    const uint64_t* __restrict__ upper_ptr, PrevMerge prev) {
    constexpr int kBlock = Cfg::kBlock;
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
    if constexpr (kWithMerge) {
        if (blockIdx.x == gridDim.x - 1) {      // the merger (the LAST workgroup: the scanners keep blockIdx = tile slot)
            if (prev.lists)
                merge_body(s_ride.merge, prev.lists, prev.n_lists, prev.topk, static_cast<int64_t>(prev.topk),
                           static_cast<int64_t>(0), prev.topk, prev.out_keys, static_cast<int64_t*>(nullptr),
                           static_cast<float*>(nullptr), static_cast<int64_t>(0), static_cast<int64_t>(0),
                           static_cast<int64_t>(0));
            return;
        }
    }
    // this workgroup's index among the scanning workgroups, and their number
    const unsigned bid = blockIdx.x;
    const unsigned nblocks = kWithMerge ? gridDim.x - 1u : gridDim.x;
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

// Wave-level twin of block_select_threshold: the calling wave holds `c` unique
// keys in registers (0 = empty), `hist` is 256 ints of LDS private to the wave.
// No barriers: LDS operations of one wave execute in order.
template <int kKeys>
__device__ inline uint64_t wave_select_threshold(const uint64_t (&mine)[kKeys], int need, bool exact,
                                                 int slack, int* hist) {
    const int lane = threadIdx.x & 63;
    uint32_t mx = 0, mn = ~0u;
#pragma unroll
    for (int r = 0; r < kKeys; ++r) {
        if (mine[r]) {
            const uint32_t hi = static_cast<uint32_t>(mine[r] >> 32);
            mx = hi > mx ? hi : mx;
            mn = hi < mn ? hi : mn;
        }
    }
    mx = wave_max_u32(mx);
    mn = wave_min_u32(mn);
    const uint64_t base = static_cast<uint64_t>(mn) << 32;
    const uint64_t span = (static_cast<uint64_t>(mx - mn) << 32) | 0xffffffffull;
    int shift = (64 - __clzll(static_cast<long long>(span))) - 8;
    if (shift < 0) shift = 0;
    uint64_t prefix = 0, mask = 0;
    const int top = 255 - 4 * lane;
    hist[top] = 0; hist[top - 1] = 0; hist[top - 2] = 0; hist[top - 3] = 0;
    for (;;) {
#pragma unroll
        for (int r = 0; r < kKeys; ++r) {
            const uint64_t k = mine[r];
            if (k) {
                const uint64_t v = k - base;
                if ((v & mask) == prefix) atomicAdd(&hist[static_cast<int>((v >> shift) & 255u)], 1);
            }
        }
        const int h0 = hist[top], h1 = hist[top - 1], h2 = hist[top - 2], h3 = hist[top - 3];
        hist[top] = 0; hist[top - 1] = 0; hist[top - 2] = 0; hist[top - 3] = 0;
        const int lane_sum = h0 + h1 + h2 + h3;
        int cum = wave_inclusive_scan(lane_sum) - lane_sum;
        const int hs[4] = {h0, h1, h2, h3};
        int digit = 0, above = 0, in_bin = 0;
        bool found = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (cum < need && cum + hs[b] >= need) {
                found = true;
                digit = top - b;
                above = cum;
                in_bin = hs[b];
            }
            cum += hs[b];
        }
        const uint64_t who = __ballot(found);
        const int src = __ffsll(static_cast<long long>(who)) - 1;  // exactly one lane when the precondition holds
        digit = __builtin_amdgcn_readlane(digit, src);
        above = __builtin_amdgcn_readlane(above, src);
        in_bin = __builtin_amdgcn_readlane(in_bin, src);
        prefix |= static_cast<uint64_t>(digit) << shift;
        mask |= 255ull << shift;
        need -= above;
        if (shift == 0 || in_bin == need || (!exact && in_bin - need <= slack)) break;
        shift = shift > 8 ? shift - 8 : 0;
    }
    return base + prefix;
}

// Wave-level compaction of one query's candidate buffer: keep the keys >= T where
// T bounds the topk-th best.  Returns the new filter threshold (key > thr passes).
template <int kKeys>
__device__ inline uint64_t wave_compact(uint64_t* cand, int* count, int topk, bool exact, int* hist) {
    const int lane = threadIdx.x & 63;
    const int c = *count;
    if (c <= topk) return 0ull;
    uint64_t mine[kKeys];
#pragma unroll
    for (int r = 0; r < kKeys; ++r) {
        const int i = lane + r * 64;
        mine[r] = i < c ? cand[i] : 0ull;
    }
    int slack = topk / 4;
    if (slack < 16) slack = 16;
    const uint64_t t = wave_select_threshold<kKeys>(mine, topk, exact, slack, hist);
    int base = 0;
#pragma unroll
    for (int r = 0; r < kKeys; ++r) {
        const bool keep = mine[r] >= t;  // t >= 1, so empty slots drop out
        const uint64_t b = __ballot(keep);
        if (keep) cand[base + lanes_below(b)] = mine[r];
        base += __popcll(b);
    }
    if (lane == 0) *count = base;
    return t - 1ull;
}

// The same with the count kept in a (wave-uniform) REGISTER.  A count that lane 0 stores to LDS and
// the other lanes load back right away is a data race as far as the compiler is concerned (each lane
// is a thread: it may keep using the value it loaded before) — callers that compact between two
// workgroup barriers can use wave_compact; a wave that compacts in the middle of its own work must
// carry the count itself.
template <int kKeys>
__device__ inline uint64_t wave_compact_reg(uint64_t* cand, int& c, int topk, bool exact, int* hist) {
    const int lane = threadIdx.x & 63;
    if (c <= topk) return 0ull;   // uniform
    uint64_t mine[kKeys];
#pragma unroll
    for (int r = 0; r < kKeys; ++r) {
        const int i = lane + r * 64;
        mine[r] = i < c ? cand[i] : 0ull;
    }
    int slack = topk / 4;
    if (slack < 16) slack = 16;
    const uint64_t t = wave_select_threshold<kKeys>(mine, topk, exact, slack, hist);
    int base = 0;
#pragma unroll
    for (int r = 0; r < kKeys; ++r) {
        const bool keep = mine[r] >= t;  // t >= 1, so empty slots drop out
        const uint64_t b = __ballot(keep);
        if (keep) cand[base + lanes_below(b)] = mine[r];
        base += __popcll(b);
    }
    c = base;
    return t - 1ull;
}

// Wave-level ranking of c <= kRankDirectMax unique keys into dst (descending, best topk).
__device__ inline void wave_rank_and_store(const uint64_t* keys, int c, uint64_t* dst, int topk) {
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < topk; i += 64)
        if (i >= c) dst[i] = 0ull;
    for (int i = lane; i < c; i += 64) {
        const uint64_t mine = keys[i];
        int rank = 0;
        int j = 0;
        for (; j + 8 <= c; j += 8) {
            const uint64_t k0 = keys[j], k1 = keys[j + 1], k2 = keys[j + 2], k3 = keys[j + 3];
            const uint64_t k4 = keys[j + 4], k5 = keys[j + 5], k6 = keys[j + 6], k7 = keys[j + 7];
            rank += (k0 > mine) + (k1 > mine) + (k2 > mine) + (k3 > mine) +
                    (k4 > mine) + (k5 > mine) + (k6 > mine) + (k7 > mine);
        }
        for (; j < c; ++j) rank += (keys[j] > mine);
        if (rank < topk) dst[rank] = mine;
    }
}

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
// every batched call paid for): the workgroup that finishes LAST merges every queued query's lists
// into that query's output row.  Rare path, so the hand-off is the plain one: a device-wide fence
// before each workgroup counts itself out and one behind the count of the last (`arrive` is zero
// between launches: the last workgroup resets it, and a launch with an empty queue never touches it).
template <typename Cfg>
__global__ __launch_bounds__(Cfg::kBlock, Cfg::kMinWaves) void scan_multi_queued_kernel(
    const float* __restrict__ feats, int64_t n, int iters, int64_t row_base,
    const float* __restrict__ queries_dev, const long long* __restrict__ exclude_dev,
    const int* __restrict__ queue, const int* __restrict__ queue_count, int topk,
    uint64_t* __restrict__ block_lists, unsigned* __restrict__ arrive,
    uint64_t* __restrict__ out_keys_base, int64_t* __restrict__ out_idx_base, float* __restrict__ out_score_base,
    int64_t out_query_stride) {
    const int count = *queue_count;
    if (count == 0) return;   // uniform over the whole grid
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
    }
    __shared__ int s_last;
    __threadfence();   // this workgroup's lists are visible device-wide ...
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool last = atomicAdd(arrive, 1u) + 1u == gridDim.x;   // ... before it is counted
        if (last) *arrive = 0u;
        s_last = last ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;   // uniform
    __threadfence();       // ... and the last one sees everybody's
    __shared__ MergeSmemT<Cfg::kBlock, 1024, kMergeSurvCap> s_merge;   // (one workgroup per CU fits with this: the launch has no more)
    for (int b = 0; b < count; ++b) {
        merge_body(s_merge, block_lists, static_cast<int>(gridDim.x), topk, static_cast<int64_t>(topk),
                   static_cast<int64_t>(gridDim.x) * topk, topk, out_keys_base, out_idx_base, out_score_base, out_query_stride,
                   static_cast<int64_t>(b), static_cast<int64_t>(queue[b]));
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

// ---- merge kernels (the merge body itself is defined above the streaming scan) ------

__global__ __launch_bounds__(kMergeBlock) void merge_kernel(
    const uint64_t* __restrict__ lists_base, int n_lists, int list_len, int64_t list_stride,
    int64_t lists_query_stride, int topk, uint64_t* __restrict__ out_keys_base,
    int64_t* __restrict__ out_idx_base, float* __restrict__ out_score_base,
    int64_t out_query_stride) {
    __shared__ MergeSmemT<kMergeBlock, kMergeMaxLists, kMergeSurvCap> sm;
    merge_body(sm, lists_base, n_lists, list_len, list_stride, lists_query_stride, topk, out_keys_base, out_idx_base,
               out_score_base, out_query_stride, blockIdx.x, blockIdx.x);
}

// The same merge for a caller that WAITS ON THE HOST (mi355rec_query_row_topn): out_idx / out_score are
// device-visible addresses of pinned host memory, and after them the workgroup stores `done_value` to
// *done_word (pinned host memory as well), so the host can poll one word instead of going through
// hipStreamSynchronize's completion path (~3 us of a 60 us query).
__global__ __launch_bounds__(kMergeBlock) void merge_notify_kernel(
    const uint64_t* __restrict__ lists_base, int n_lists, int list_len, int64_t list_stride, int topk,
    uint64_t* __restrict__ out_keys_base, int64_t* __restrict__ out_idx_base, float* __restrict__ out_score_base,
    uint32_t* done_word, uint32_t done_value) {
    __shared__ MergeSmemT<kMergeBlock, kMergeMaxLists, kMergeSurvCap> sm;
    merge_body(sm, lists_base, n_lists, list_len, list_stride, static_cast<int64_t>(0), topk, out_keys_base, out_idx_base,
               out_score_base, static_cast<int64_t>(0), static_cast<int64_t>(0), static_cast<int64_t>(0));
    // the waves that stored results order their stores before ... (the others have nothing to release: a system-scope
    // fence is an L2 write-back per wave, and sixteen of them queue up)
    if (static_cast<int>(threadIdx.x) < ((topk + 63) & ~63)) __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done_word, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... the word
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
