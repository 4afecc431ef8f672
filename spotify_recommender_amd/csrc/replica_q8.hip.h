// replica_q8.hip.h — the 8-BIT replica of the catalogue and the single-query scan over it.
//
// replica.hip.h halves the bytes a query streams (24 B per row instead of the reference's 48,
// Recommender.cu:184-254) with a copy that is only good enough to rule rows OUT.  The same idea one step
// further: 12 B per row.  Each row is L2-normalised in fp32 and every component quantised to a SIGNED byte,
//     k_j = round(127 r^_j)   in [-127, 127],          |k_j - 127 r^_j| <= 1/2,
// and the query (normalised q^ = q / |q|) to 16 bits, held as two balanced int8 digits per component,
//     Q_j = round(S q^_j) = 256 h_j + l_j,   S = 32000,   h_j in [-125, 125],   l_j in [-128, 127],
// so that the dot product the reference takes with cublasSgemv (Recommender.cu:217-223) is, for ruling rows out,
// SIX v_dot4_i32_i8 per row — three over the high digits, three over the low ones — and one shift-add:
//     D = sum_j k_j Q_j = 256 sum_j k_j h_j + sum_j k_j l_j          (int32, exact: |D| < 4.2e6)
//     approx = D / (127 S),        approx - r^ . q^ = sum_j (k_j/127 - r^_j) Q_j/S + sum_j r^_j (Q_j/S - q^_j)
//     |approx - r^ . q^| <= l1(Q) / (254 S) + l1(r^) / (2 S) <= l1(Q) / (254 S) + sqrt(12) / (2 S).
// (Until round 5 the query stayed fp32 and a row cost 12 v_cvt_f32_ubyte + 12 v_fmac: 96 vector instructions per lane
// and 48 B, which kept the kernel at 0.75 of a plain read of its own buffer.)  The candidate test itself is an INTEGER
// compare: a row is out iff D < floor(cutoff * 127 S) - 1 (q8_threshold: the fp32 product is off by < 0.25).
// The bound is PER QUERY (0.0137 at worst, ~0.012 for a typical query) and derived, not tuned: the row's
// quantisation error of component j is at most 1/254 and enters multiplied by |Q_j| / S, the query's is at most
// 1 / (2 S) and enters multiplied by |r^_j| (5.5e-5 in all); the normalisations in fp32 (v_rsq_f32: the byte may sit
// 4e-5 of a step off its real-number position), the one rounding of D / (127 S) where a float is wanted and the
// reference chain's own rounding are covered by kQ8Slack = 3e-5 (together < 8e-6).  tests/test_q8_margin.py checks
// it on hostile data with a numpy model of exactly this arithmetic.  Everything else is replica.hip.h's scheme:
//   * the contract per query is recommendByIndex's (Recommender.cu:275-318): every key that leaves the
//     kernel is cosine_score() on the fp32 row (calculateSimilaritiesCPU, :256-273), bit for bit;
//   * valid row: |row|^2 in [kBqMinNorm2, kBqMaxNorm2]; an exactly-zero row stores k = 0 everywhere
//     (approx = 0 = its exact score); every other row (tiny, huge, inf, NaN) stores 0x80 = -128 — a byte no
//     valid row contains — and is sent to the exact chain every time;
//   * an invalid query (|q| outside [kBqMinNorm, kBqMaxNorm]) switches the pre-filter off for the launch;
//   * the launch-wide cutoff comes from a spread sample of 256-row wave tiles (5 % of the catalogue): per tile the
//     EXACT score of the row with the largest approx (q8_region_pick / q8_region_store); v = the topk-th largest of
//     those <= 2048 exact scores, cutoff = v - margin (ONE margin: v is the exact score of a real row; shards too
//     small to afford the extra fetch keep approximate sample values and two margins); workgroup-local thresholds
//     tighten it whenever a local list fills.
// With a margin twelve times the fp16 replica's, ~0.1 % of the rows (~9 000 of 10 M at top-100) go to the exact
// chain instead of 0.05 % — 0.6 MB of random 48 B fetches beside 120 MB of stream.
//
// One lane = FOUR rows = 48 B (3 x dwordx4: the same load pattern once more; row r of the lane is dwords
// 3r .. 3r + 2), tiles of 4 * kBlock rows dealt round-robin over the scanning workgroups.
//
// -DMI355_Q8_QUERY_BITS=8 (A/B builds only) quantises the query to ONE int8 digit, P_j = round(127 q^_j): three
// v_dot4 per row, and a margin of l1(P) / (254 * 127) + sqrt(12) / 254 — twice the rows to the exact chain.
#pragma once

#include "replica.hip.h"

#ifndef MI355_Q8_QUERY_BITS
#define MI355_Q8_QUERY_BITS 16
#endif

namespace mi355 {

constexpr int kQ8QueryBits = MI355_Q8_QUERY_BITS;
static_assert(kQ8QueryBits == 16 || kQ8QueryBits == 8, "the query is one or two int8 digits per component");
constexpr float kQ8Step = 1.0f / 254.0f;     // half a quantisation step of a normalised row component
constexpr float kQ8QueryScale = kQ8QueryBits == 16 ? 32000.0f : 127.0f;   // S: Q_j = round(S q^_j)
constexpr float kQ8DotScale = 127.0f * kQ8QueryScale;                      // approx = D / (127 S)
constexpr float kQ8QueryResidual = 3.4642f * 0.5f / kQ8QueryScale;         // l1(r^) / (2 S), l1(r^) <= sqrt(12)
constexpr float kQ8Slack = 3e-5f;            // fp32 normalisations of row and query, the byte's rsq-induced offset, the rounding of
                                             // D / (127 S) and the reference chain's own rounding (together < 8e-6), with room to spare
constexpr uint32_t kQ8Special = 0x80u;       // first byte of a row the bound is not claimed for (-128: no valid row holds it)

// ---- building the replica ---------------------------------------------------------------------------
// One thread per row; rows [n, n_padded) (n_padded a multiple of 4) are padding and hold the special marker.
__global__ __launch_bounds__(256) void q8_build_kernel(const float* __restrict__ feats, int64_t n, int64_t n_padded,
                                                       uint32_t* __restrict__ q8) {
    const int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= n_padded) return;
    uint32_t d0 = 0x80808080u, d1 = 0x80808080u, d2 = 0x80808080u;   // special
    if (row < n) {
        const float4* p = reinterpret_cast<const float4*>(feats) + row * 3;
        const float4 a = p[0], b = p[1], c = p[2];
        // the normalisation of the fp16 replica and of the batched passes (replica_build_kernel)
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
        if (valid || tot == 0.0f) {
            const float inv = valid ? __builtin_amdgcn_rsqf(tot) * 127.0f : 0.0f;
            auto q = [&](float x) {   // round to nearest; |x * inv| <= 127 (1 + 1e-6); two's complement byte
                int k = static_cast<int>(__builtin_rintf(x * inv));
                k = k > 127 ? 127 : (k < -127 ? -127 : k);
                return static_cast<uint32_t>(k) & 0xffu;
            };
            d0 = q(a.x) | (q(a.y) << 8) | (q(a.z) << 16) | (q(a.w) << 24);
            d1 = q(b.x) | (q(b.y) << 8) | (q(b.z) << 16) | (q(b.w) << 24);
            d2 = q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | (q(c.w) << 24);
        }
    }
    uint32_t* dst = q8 + row * 3;
    dst[0] = d0;
    dst[1] = d1;
    dst[2] = d2;
}

// ---- the query ------------------------------------------------------------------------------------------
struct Q8Query {
    int hi[3];       // the high int8 digits h_j, four per dword in the row's byte order (wave-uniform: scalar registers)
    int lo[3];       // the low digits l_j (16-bit query only)
    float inv;       // 1 / (127 S):   approx = D * inv
    float margin;    // l1(Q) / (254 S) + l1(r^) / (2 S) + slack
    bool ok;         // the bound may be claimed for this query
};

__device__ __forceinline__ Q8Query q8_query(const float (&q)[kDim], float qn) {
    Q8Query r;
    r.ok = qn >= kBqMinNorm && qn <= kBqMaxNorm;   // false for NaN
    const float inv = r.ok ? 1.0f / qn : 0.0f;
    r.hi[0] = r.hi[1] = r.hi[2] = 0;
    r.lo[0] = r.lo[1] = r.lo[2] = 0;
    int l1 = 0;
    constexpr int kMax = static_cast<int>(kQ8QueryScale);
#pragma unroll
    for (int j = 0; j < kDim; ++j) {
        int Q = static_cast<int>(__builtin_rintf(q[j] * inv * kQ8QueryScale));   // |q^_j| <= 1 + 1e-6
        Q = Q > kMax ? kMax : (Q < -kMax ? -kMax : Q);
        l1 += Q < 0 ? -Q : Q;
        if constexpr (kQ8QueryBits == 16) {
            const int h = (Q + 128) >> 8;      // balanced digits: Q = 256 h + l, l in [-128, 127], |h| <= 125
            const int l = Q - 256 * h;
            r.hi[j >> 2] |= (h & 0xff) << (8 * (j & 3));
            r.lo[j >> 2] |= (l & 0xff) << (8 * (j & 3));
        } else {
            r.hi[j >> 2] |= (Q & 0xff) << (8 * (j & 3));
        }
    }
    r.inv = 1.0f / kQ8DotScale;
    r.margin = static_cast<float>(l1) * (kQ8Step / kQ8QueryScale) * (1.0f + 1e-5f) + kQ8QueryResidual + kQ8Slack;
    return r;
}

// D of one row (3 dwords); `special` = the row is sent to the exact chain whatever D says
__device__ __forceinline__ int q8_dot(const Q8Query& q, uint32_t d0, uint32_t d1, uint32_t d2, bool& special) {
    special = (d0 & 0xffu) == kQ8Special;
    int hi = __builtin_amdgcn_sdot4(static_cast<int>(d0), q.hi[0], 0, false);
    hi = __builtin_amdgcn_sdot4(static_cast<int>(d1), q.hi[1], hi, false);
    hi = __builtin_amdgcn_sdot4(static_cast<int>(d2), q.hi[2], hi, false);
    if constexpr (kQ8QueryBits == 16) {
        int lo = __builtin_amdgcn_sdot4(static_cast<int>(d0), q.lo[0], 0, false);
        lo = __builtin_amdgcn_sdot4(static_cast<int>(d1), q.lo[1], lo, false);
        lo = __builtin_amdgcn_sdot4(static_cast<int>(d2), q.lo[2], lo, false);
        return (hi << 8) + lo;   // (v_lshl_add_u32)
    } else {
        return hi;
    }
}

__device__ __forceinline__ void q8_dot4(const Q8Query& q, const HalfTile& t, int (&a)[4], bool (&special)[4]) {
    a[0] = q8_dot(q, t.t0.x, t.t0.y, t.t0.z, special[0]);
    a[1] = q8_dot(q, t.t0.w, t.t1.x, t.t1.y, special[1]);
    a[2] = q8_dot(q, t.t1.z, t.t1.w, t.t2.x, special[2]);
    a[3] = q8_dot(q, t.t2.y, t.t2.z, t.t2.w, special[3]);
}

// approx as a float (samples, diagnostics): |D| < 2^24 converts exactly, one rounding in the product
__device__ __forceinline__ float q8_approx(const Q8Query& q, int d) { return static_cast<float>(d) * q.inv; }

// The integer image of a cutoff: a row whose D is below it has approx < cutoff.  The fp32 product cutoff * 127 S is off
// by < 0.25 (|cutoff| <= 2: 2^-24 relative of at most 8.2e6), floor(...) - 1 stays below the real-number product.
// -inf / NaN (no bound) => INT_MIN: every row is a candidate.
__device__ __forceinline__ int q8_threshold(float cutoff) {
    if (!(cutoff > -2.0f)) return static_cast<int>(0x80000000u);
    const float c = cutoff < 2.0f ? cutoff : 2.0f;
    return static_cast<int>(__builtin_floorf(c * kQ8DotScale)) - 1;
}

// ---- the sample that seeds the launch-wide cutoff ---------------------------------------------------------
// A REGION is 2048 rows from row g * stride_rows on (stride_rows a multiple of 4 and >= 2048): one ordered-u32
// approx maximum per 256-row wave tile (0 = nothing usable) goes to seed_vals[g * 8 + wave].
struct Q8Region {
    HalfTile t;
    int64_t quad;
    bool have;
};

__device__ __forceinline__ Q8Region q8_region_load(const uint4* __restrict__ q8, int64_t n_quads, int64_t stride_rows, int64_t g) {
    Q8Region s;
    s.quad = ((g * stride_rows) >> 2) + threadIdx.x;
    s.have = s.quad < n_quads;
    s.quad = s.have ? s.quad : n_quads - 1;
    const uint4* p = q8 + s.quad * 3;
    s.t.t0 = p[0];
    s.t.t1 = p[1];
    s.t.t2 = p[2];
    return s;
}

// The sample value of one wave's 256 rows is the EXACT score of the row whose approximate score is the largest
// (one 48 B fetch per wave, all lanes the same address): an exact score v of a real row needs only ONE margin in
// the cutoff (approx < v - margin => exact < v), where the approximate maximum needed two.  At top-100 over
// 10 M rows that is the difference between ~28 000 and ~10 000 rows sent to the exact chain per query.
// (kExact = false keeps the approximate maximum and skips the fetch: small shards, where the riders' extra round
// trip is on the critical path of the launch and a few hundred more candidates are not.)
struct Q8Pick {
    int64_t row;     // wave-uniform: local row of the wave's best usable row
    uint32_t top;    // wave-uniform: its approximate score, ordered (0: no usable row)
    bool any;        // wave-uniform
};

__device__ __forceinline__ Q8Pick q8_region_pick(const Q8Region& s, const Q8Query& q, int64_t n, int64_t row_base,
                                                 int64_t exclude_global) {
    int a[4];
    bool special[4];
    q8_dot4(q, s.t, a, special);
    const int64_t r0 = s.quad * 4;
    uint32_t v = 0u;
    int best = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool use = s.have && q.ok && !special[u] && r0 + u < n && row_base + r0 + u != exclude_global;
        const uint32_t w = use ? score_to_ordered(q8_approx(q, a[u])) : 0u;
        best = w > v ? u : best;
        v = w > v ? w : v;
    }
    const uint32_t top = wave_max_u32(v);
    const uint64_t holders = __ballot(v == top);   // never empty
    const int lane = static_cast<int>(__builtin_ctzll(holders));
    Q8Pick p;
    p.top = top;
    p.any = top != 0u;
    const int64_t mine = r0 + best;
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(mine)), lane));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(mine >> 32)), lane));
    p.row = p.any ? static_cast<int64_t>((static_cast<uint64_t>(hi) << 32) | lo) : 0;
    return p;
}

template <bool kExact>
__device__ __forceinline__ void q8_region_store(const Q8Pick& p, const Row& row, const float (&q)[kDim], float qn,
                                                unsigned long long* __restrict__ seed_vals, uint32_t epoch, int64_t g) {
    uint32_t v = p.top;
    if constexpr (kExact) {
        const float exact = cosine_score(q, qn, row);
        v = p.any ? score_to_ordered(exact) : 0u;
    }
    // written THROUGH to device scope, under the query's epoch: a rider of the same launch may read it
    // (scan_q8_kernel, last rider out; replica.hip.h, "hand-offs that fail safe")
    if ((threadIdx.x & 63) == 0)
        __hip_atomic_store(&seed_vals[g * kHalfSeedWaves + (threadIdx.x >> 6)], tag_value(epoch, v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void q8_load_query(const float* __restrict__ query_ptr, const float (&by_value)[kDim], float (&q)[kDim]) {
    if (query_ptr) {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = query_ptr[j];
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = by_value[j];
    }
}

// One workgroup per region (single queries; the first query of a stream) and, when the grid has one more workgroup
// than regions, the neighbourhood of the excluded row (handoff.hip.h) by that last one.
template <bool kExact>
__global__ __launch_bounds__(kHalfSeedBlock) void seed_q8_kernel(
    const float* __restrict__ feats, const uint4* __restrict__ q8, int64_t n, int64_t stride_rows, int64_t row_base, QueryArg qarg,
    const float* __restrict__ query_ptr /* null: the query is qarg.q */, int64_t exclude_global,
    unsigned long long* __restrict__ seed_vals, uint32_t epoch, int regions, int topk,
    const float* __restrict__ anchors /* the handle's anchor table (handoff.hip.h), or null */) {
    if (static_cast<int>(blockIdx.x) >= regions) {   // uniform
        // (1024 rows: this workgroup is the last one out of the sample launch of a query alone, and the scan subtracts a margin
        // of ~0.01 from whatever bound it is given — the 10th percentile of the neighbourhood serves it as well as the 5th)
        __shared__ int s_scratch[Nbhd<kHalfSeedBlock, 1024>::kScratch];
        nbhd_to_slot<kHalfSeedBlock, 1024>(feats, n, row_base, query_ptr, qarg.q, exclude_global, topk, epoch, seed_vals, s_scratch, anchors);
        return;
    }
    // the region's rows are requested FIRST: they need nothing of the query, whose 12 floats sit behind two dependent
    // scalar loads and a norm (this launch is on the critical path of a query alone)
    const Q8Region s = q8_region_load(q8, (n + 3) >> 2, stride_rows, blockIdx.x);
    float q[kDim];
    q8_load_query(query_ptr, qarg.q, q);
    const float qn = query_norm(q);
    const Q8Query hq = q8_query(q, qn);
    const Q8Pick p = q8_region_pick(s, hq, n, row_base, exclude_global);
    Row row;
    row.a = row.b = row.c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (kExact) row = load_row(feats, p.row);
    q8_region_store<kExact>(p, row, q, qn, seed_vals, epoch, blockIdx.x);
}

// The seed riders of a streamed launch (handoff.hip.h, NextSeed): four regions per memory round trip.
__device__ __forceinline__ Q8Query q8_seed_rider(const float* __restrict__ feats, const uint4* __restrict__ q8, int64_t n,
                                                 int64_t row_base, const NextSeed& next, int rider) {
    const int64_t n_quads = (n + 3) >> 2;
    unsigned long long* const out = static_cast<unsigned long long*>(next.out);
    constexpr int kAhead = 4;
    // A round's regions are requested before anything that does not need them: the first round before the QUERY is
    // (two dependent scalar loads and a norm: on a 1 M-row shard the riders' chain is the longest thing in the launch),
    // every later one right after the round before has been reduced, under its winners' fetches and stores.
    Q8Region s[kAhead];
    auto load_round = [&](int g0) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int g = g0 + u * next.n_wgs;
            s[u] = q8_region_load(q8, n_quads, next.stride_rows, g < next.regions ? g : rider);
        }
    };
    if (rider < next.regions) load_round(rider);
    float q[kDim];
    q8_load_query(next.query_ptr, next.q, q);
    const float qn = query_norm(q);
    const Q8Query hq = q8_query(q, qn);
    for (int g0 = rider; g0 < next.regions; g0 += kAhead * next.n_wgs) {
        Q8Pick pick[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) pick[u] = q8_region_pick(s[u], hq, n, row_base, next.exclude_global);
        if (g0 + kAhead * next.n_wgs < next.regions) load_round(g0 + kAhead * next.n_wgs);   // uniform
        if (next.exact) {   // uniform: the four winners' rows, one more round trip with four fetches in flight
            Row best[kAhead];
#pragma unroll
            for (int u = 0; u < kAhead; ++u) best[u] = load_row(feats, pick[u].row);
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const int g = g0 + u * next.n_wgs;
                if (g < next.regions && g >= next.debug_skip) q8_region_store<true>(pick[u], best[u], q, qn, out, next.epoch, g);   // uniform
            }
        } else {
            Row none;
            none.a = none.b = none.c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int u = 0; u < kAhead; ++u) {
                const int g = g0 + u * next.n_wgs;
                if (g < next.regions && g >= next.debug_skip) q8_region_store<false>(pick[u], none, q, qn, out, next.epoch, g);   // uniform
            }
        }
    }
    return hq;
}

// The launch-wide cutoff from n_seed sample values (handoff.hip.h: sample_kth_value): a row whose approximate score
// is below it cannot be among the best topk.  -inf when the sample cannot say, or the bound cannot be claimed for
// the query.  One margin below an EXACT score of a real row, two below an approximate one (the margin carries its
// own slack).
template <int kBlock>
__device__ __forceinline__ float q8_cutoff_from_sample(const Sample& sample, int n_seed, int topk, bool exact_values,
                                                       const Q8Query& hq, int* s_seeds /* zeroed */, SelectSmem& s_sel,
                                                       int* s_bins /* kSelScratch ints nobody else is using */) {
    if (!hq.ok) return -__builtin_inff();   // uniform
    const float v = sample_kth_value<kBlock>(sample, n_seed, topk, s_seeds, s_sel, s_bins);
    return exact_values ? v - hq.margin : v - 2.0f * hq.margin;   // (-inf stays -inf)
}

// ---- the scan ----------------------------------------------------------------------------------------------
template <int kBlockT, int kMinWavesT, int kDepthT>
struct Q8Cfg {
    static constexpr int kBlock = kBlockT;
    static constexpr int kMinWaves = kMinWavesT;
    static constexpr int kDepth = kDepthT;
    static constexpr int kTileRows = 4 * kBlockT;
    static constexpr int kCandCap = kCandLimit + kTileRows;
    static constexpr int kCandPerThread = (kCandCap + kBlockT - 1) / kBlockT;
};
using DefaultQ8Cfg = Q8Cfg<512, 4, 2>;

// kWithMerge (streamed queries): workgroups [0, S) scan, workgroup S merges the PREVIOUS streamed query, the next
// next.n_wgs workgroups are seed riders for the NEXT one and (next.nbhd) the last of the grid takes that query's
// neighbourhood (handoff.hip.h); S = gridDim.x - 1 - next.n_wgs - next.nbhd.
// Slot kNbhdSlot of `seed_vals` holds, under this query's epoch, the neighbourhood's EXACT bound v (0: none): keys below
// it are dropped at once and v - margin is one more lower bound on the cutoff — what keeps a catalogue sorted by genre
// (the query's cluster in ONE sampled region at most) from scanning with another cluster's cutoff.
//
// Tiles are dealt round-robin.  Handing them out dynamically (ticket counters, the merger and the riders joining
// once their job was done) was built and measured: every workgroup then finished within 2 us of the others
// instead of 8 — and the launch took as long as before, because between prologue and last tile the launch
// already moves its bytes at ~6.5 TB/s and early finishers only leave that bandwidth to the rest.
//
// kLoneTail (a lone query whose caller waits on the host): the workgroup that finishes last merges the lists of
// THIS launch into the caller's buffers and raises the completion word (kernels.hip.h, lone_tail).
template <typename Cfg, bool kQueryFromRow, bool kWithMerge, bool kLoneTail = false>
__global__ __launch_bounds__(Cfg::kBlock, Cfg::kMinWaves) void scan_q8_kernel(
    const float* __restrict__ feats, const uint4* __restrict__ q8, int64_t n, int iters, int64_t row_base,
    QueryArg qarg, const float* __restrict__ query_ptr, int64_t exclude_global, int topk, uint64_t* __restrict__ block_lists,
    const unsigned long long* __restrict__ seed_vals /* tagged with `epoch` */, int n_seed /* negative: |n_seed| EXACT sample values */,
    unsigned long long* __restrict__ rescored /* [scanning workgroups] */,
    PrevMerge prev, NextSeed next, const unsigned long long* __restrict__ cutoff_ready /* tagged; null: select from seed_vals here */,
    LoneTail lone, uint32_t epoch /* of this query */) {
    constexpr int kBlock = Cfg::kBlock;
    static_assert(!(kWithMerge && kLoneTail), "a streamed query's merge rides in the next launch");
    __shared__ typename std::conditional<kWithMerge || kLoneTail, HalfScanOrMergeSmem<Cfg>, HalfScanSmemT<Cfg>>::type s_mem;
    __shared__ int s_lone_flag;
    HalfScanSmemT<Cfg>* sm;
    unsigned nblocks = gridDim.x;   // scanning workgroups
    MI355REC_PHASE(0);
    if constexpr (kWithMerge) {
        nblocks = gridDim.x - 1u - static_cast<unsigned>(next.n_wgs) - static_cast<unsigned>(next.nbhd);
        if (blockIdx.x >= nblocks) {
            if (blockIdx.x == nblocks) {
                if (prev.lists)
                    merge_body(s_mem.merge, prev.lists, prev.n_lists, prev.topk, static_cast<int64_t>(prev.topk),
                               static_cast<int64_t>(0), prev.topk, prev.out_keys, static_cast<int64_t*>(nullptr),
                               static_cast<float*>(nullptr), static_cast<int64_t>(0), static_cast<int64_t>(0),
                               static_cast<int64_t>(0));
            } else if (next.nbhd && blockIdx.x == gridDim.x - 1u) {   // the next query's neighbourhood: read by the NEXT launch
                nbhd_to_slot<kBlock>(feats, n, row_base, next.query_ptr, next.q, next.exclude_global, next.topk, next.epoch,
                                     static_cast<unsigned long long*>(next.out), reinterpret_cast<int*>(s_mem.scan.cand), next.anchors);
            } else {
                const Q8Query nq = q8_seed_rider(feats, q8, n, row_base, next, static_cast<int>(blockIdx.x - nblocks - 1u));
                // Last rider out turns the sample into the cutoff (handoff.hip.h, sample_arrive_and_select: no device-wide
                // fence under the scanners).  Values and cutoff carry the next query's epoch and the counter is never reset.
                if (next.ctl) {   // uniform; null: nobody reads the sample inside this launch (small shards: the next launch selects)
                    float v;
                    if (sample_arrive_and_select<kBlock>(next.ctl, next.done_base, static_cast<unsigned>(next.n_wgs),
                                                         static_cast<const unsigned long long*>(next.out), next.regions * kHalfSeedWaves,
                                                         next.topk, next.epoch, &s_mem.scan.count, &s_mem.scan.seeds, s_mem.scan.sel,
                                                         reinterpret_cast<int*>(s_mem.scan.cand), v)) {
                        // one margin below an EXACT score of a real row, two below an approximate one
                        const float c = !nq.ok ? -__builtin_inff() : (next.exact != 0 ? v - nq.margin : v - 2.0f * nq.margin);
                        if (threadIdx.x == 0) next.ctl->cutoff = tag_value(next.epoch, __float_as_uint(c));
                    }
                }
            }
            __syncthreads();
            MI355REC_PHASE(5);
            return;
        }
        sm = &s_mem.scan;
    } else if constexpr (kLoneTail) {
        (void)next;
        sm = &s_mem.scan;
    } else {
        (void)next;
        (void)lone;
        (void)s_lone_flag;
        sm = &s_mem;
    }
    uint64_t* const s_cand = sm->cand;
    SelectSmem& s_sel = sm->sel;
    int& s_count = sm->count;

    const unsigned bid = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;

    const int64_t n_quads = (n + 3) >> 2;
    const int64_t last_quad = n_quads - 1;
    const int64_t quad_begin = static_cast<int64_t>(bid) * kBlock + tid;
    const int64_t quad_stride = static_cast<int64_t>(nblocks) * kBlock;

    auto load_tile = [&](HalfTile& dst, int it) {
        int64_t quad = quad_begin + static_cast<int64_t>(it) * quad_stride;
        quad = quad < n_quads ? quad : last_quad;   // unconditional prefetch (see scan_kernel)
        const uint4* p = q8 + quad * 3;
        dst.t0 = p[0];
        dst.t1 = p[1];
        dst.t2 = p[2];
    };

    // everything the first tile needs is asked for at once: the finished cutoff — or, where this workgroup selects it
    // itself, the sample values, FIRST: they come back before the tile does and the selection runs under its latency —
    // the tile's 48 B per lane, the query
    constexpr int kDepth = Cfg::kDepth;
    float cutoff_left = 0.0f;
    const unsigned long long nbhd_raw = seed_vals[kNbhdSlot];   // (wave-uniform: a scalar load)
    SampleRaw sample_raw;
#pragma unroll
    for (int r = 0; r < kHalfSeedPerThread; ++r) sample_raw.t[r] = 0ull;
    const int n_sample = n_seed < 0 ? -n_seed : n_seed;
    if (cutoff_ready) {   // uniform: the riders of the launch before this one left it (under this query's epoch, or it does not count)
        cutoff_left = untag_cutoff(*cutoff_ready, epoch);
    } else {              // ... or this workgroup selects it from the sample values itself
        sample_raw = sample_request<kBlock>(seed_vals, n_sample);
    }
    HalfTile ring[kDepth];
#pragma unroll
    for (int d = 0; d < kDepth - 1; ++d) load_tile(ring[d], d);
    float q[kDim];
    q8_load_query(kQueryFromRow ? query_ptr : nullptr, qarg.q, q);
    const float qn = query_norm(q);
    const Q8Query hq = q8_query(q, qn);
    const Sample sample = sample_finish(sample_raw, epoch);
    if (hq.hi[0] != 0 || n >= 0) MI355REC_PHASE(1);   // (depends on the query: not hoisted above its load)
    if (sample.v[0] != 0x12345u || n >= 0) MI355REC_PHASE(6);   // (depends on this thread's sample values: they have arrived)

    // ---- launch-wide cutoff (while the first tiles are in flight)
    if (tid == 0) {
        s_count = 0;
        sm->seeds = 0;
        sm->rescored = 0;
    }
    int n_rescored = 0;   // wave-uniform: rows this wave sent to the exact chain (diagnostics)
    __syncthreads();
    float cutoff;
    if (cutoff_ready) {   // uniform
        cutoff = cutoff_left;
    } else {
        cutoff = q8_cutoff_from_sample<kBlock>(sample, n_sample, topk, n_seed < 0, hq, &sm->seeds, s_sel, reinterpret_cast<int*>(s_cand));
    }
    uint64_t thr = 0;
    if (const uint32_t nbv = untag_value(nbhd_raw, epoch)) {   // uniform: at least topk rows score >= this EXACT bound
        thr = (static_cast<uint64_t>(nbv) << 32) - 1ull;   // (a key AT the bound passes: key > thr)
        if (hq.ok) {
            const float nb_cut = ordered_to_score(nbv) - hq.margin;
            cutoff = nb_cut > cutoff ? nb_cut : cutoff;
        }
    }
    int cut_d = q8_threshold(cutoff);   // the cutoff's integer image: a row is out iff D < cut_d
    MI355REC_PHASE(2);
    int compact_at = 2 * topk > 256 ? 2 * topk : 256;
    if (compact_at > kCandLimit) compact_at = kCandLimit;

    // A row the replica cannot rule out costs one random 48 B fetch from the fp32 matrix, ISSUED when the
    // row is found and CONSUMED one tile later (a one-entry pending slot per lane), so its latency overlaps
    // the next tile.  A lane's second, third and fourth candidate of one tile are scored on the spot (rare).
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
        const int64_t quad = quad_begin + static_cast<int64_t>(it) * quad_stride;
        const int64_t r0 = quad * 4;
        int a[4];
        bool special[4];
        q8_dot4(hq, t, a, special);
        // The common case — no lane of the wave holds a candidate — is decided on the scalar side: four integer compares and four
        // byte compares per lane, their wave masks OR-ed.  Whether a row lies inside the shard at all (the last quad's
        // padding, the clamped prefetch past the end) only matters once something hit, and is settled there.
        bool hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) hit[u] = special[u] | (a[u] >= cut_d);
        const uint64_t maybe = __ballot(hit[0] | hit[1] | hit[2] | hit[3]);
        consume();   // the rows fetched while the previous tile was scanned
        uint32_t mask = 0u;   // which of the lane's four rows are candidates
        if (maybe) {          // uniform
            const int64_t left = n - r0;
            const int valid = quad < n_quads ? (left < 4 ? static_cast<int>(left) : 4) : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) mask |= (hit[u] && u < valid) ? (1u << u) : 0u;
        }
        const uint64_t any = maybe ? __ballot(mask != 0u) : 0ull;
        if (any) {
            n_rescored += __popcll(any);
            const int first = mask ? __builtin_ctz(mask) : 0;
            // lanes without a candidate re-read row 0: one cached line
            pend_r = r0 + first;
            pend = load_row(feats, mask ? pend_r : static_cast<int64_t>(0));
            pend_on = mask != 0u;
            uint32_t rest = mask & (mask - 1u);   // the lane's further candidates, if any
            while (__ballot(rest != 0u)) {         // uniform, rare
                const int u = rest ? __builtin_ctz(rest) : 0;
                const bool have = rest != 0u;
                n_rescored += __popcll(__ballot(have));
                const Row extra = load_row(feats, have ? r0 + u : static_cast<int64_t>(0));
                append(have, extra, r0 + u);
                rest &= rest - 1u;
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
                    const float local_cut = ordered_to_score(static_cast<uint32_t>(thr >> 32)) - hq.margin;
                    cutoff = local_cut > cutoff ? local_cut : cutoff;
                    cut_d = q8_threshold(cutoff);
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
    MI355REC_PHASE(3);
    if (tid == 0) rescored[bid] += static_cast<unsigned long long>(sm->rescored);   // launches of a handle are stream-ordered
    // (from kRankCountMax keys up the ranking is a bitonic sort of the next power of two — 45 barrier stages, ~2.7 us
    // for the ~250 keys a scan without a launch-wide cutoff ends with; an INEXACT cut to topk + topk / 4 first costs a
    // radix pass or two and leaves a set the counting rank handles)
    if (s_count > kRankCountMax && s_count > topk)  // uniform
        compact_candidates<kBlock, Cfg::kCandPerThread>(s_cand, &s_count, topk, false, s_sel);
    __syncthreads();
    block_rank_and_store<kBlock, kLoneTail>(s_cand, s_count, block_lists + static_cast<int64_t>(bid) * topk, topk);
    if constexpr (kLoneTail) lone_tail(s_mem.merge, &s_lone_flag, block_lists, topk, lone);
    __syncthreads();
    MI355REC_PHASE(4);
}

}  // namespace mi355
