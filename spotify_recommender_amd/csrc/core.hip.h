// core.hip.h — what every kernel of the MI355X cosine top-N engine shares (gfx950 only): the exact score of one
// row, packed keys, and the wave- and workgroup-level selection primitives.
//
// Arithmetic follows the reference's CPU path bit for bit (calculateSimilaritiesCPU, Recommender.cu:256-273):
// sequential j = 0..11, multiply and add rounded separately (contraction is OFF for these files), correctly rounded
// sqrtf and '/', threshold 1e-8f, std::min/std::max clamp.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "experiments.hip.h"

#pragma clang fp contract(off)

namespace mi355 {

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
constexpr int kMergeChunk = 16;       // keys probed per list in the FIRST deeper merge round (then 32, 64, 128: merge.hip.h)
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

// ---- wave-level selection (no workgroup barrier: LDS operations of one wave execute in order) ---------------------------
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

}  // namespace mi355
