// replica_multi.hip.h — ONE pass over the fp16 replica for 2 ... 32 queries.
//
// The missing middle between one query (scan_half_kernel, replica.hip.h: 24 B per row, 12
// v_fma_mix_f32 per row) and the two-pass matrix-core path for hundreds of queries (batched.hip.h).
// Per query the contract is recommendByIndex's (Recommender.cu:275-318): every key that leaves the
// kernel is the exact fp32 chain of calculateSimilaritiesCPU (:256-273, cosine_score()) on the fp32
// row; the replica only rules rows OUT (bound and special rows: replica.hip.h's header).
//
// Why the matrix core.  The first version of this kernel evaluated the fp16 dot products on the
// vector ALU: 24 v_fma_mix_f32 per lane (= pair of rows) per query.  Measured at 10 M rows x 12
// queries: 118 us, ~8 cycles per v_fma_mix_f32 — instruction-bound at 3x the HBM time.  One
// v_mfma_f32_32x32x16_f16 multiplies 32 rows x 32 queries in the 32 cycles the vector ALU needs for
// FOUR fma_mix, so the pass is HBM-bound again for up to 32 queries (batched.hip.h's operand
// layout, its threshold-in-the-K-slots trick and its max3 hit test are reused):
//   A (rows)     lane = one PAIR of rows = 48 B of replica (scan_half_kernel's load pattern); the even
//                and the odd rows of 64 lanes become two 32-row tiles each by v_permlane32_swap;
//                k = 12, 13 carry 1.0, k = 14, 15 zero.
//   B (queries)  column c = query c: its L2-normalised fp16 pairs in k = 0..11, and -T' (the query's
//                approx cutoff, fp16 hi + lo) in k = 12, 13 — so D = approx - T' comes out of the matrix
//                core and a row is a candidate iff D's sign bit is clear.  The fragment lives in LDS
//                (1 KiB) and is re-read every step: the cutoffs tighten while the pass runs.
// A candidate (query, row) is only NOTED in the hot loop (a per-wave staging buffer in LDS).  When a
// wave's buffer fills — and once at the end — the wave RESOLVES it: one candidate per lane, the
// fp32 row fetched (random 48 B), scored with cosine_score(), and the keys that beat the query's
// current threshold are appended to the workgroup's key list of that query under a per-query spin
// lock in LDS (a wave-level append of <= 64 keys; a full list is cut back to topk by the wave-level
// radix select, which raises the threshold and with it the cutoff in the B fragment).  No workgroup
// barrier in the hot loop, nothing that can overflow: hostile data (no usable sample, mass ties,
// catalogues of special rows) only makes the resolves frequent.  With a seeded cutoff a wave notes
// ~20 candidates in a whole launch and resolves once, at the end.
//
// Cutoff.  seed_half_multi_kernel takes the sample of replica.hip.h (1024 rows of up to 256 evenly
// spaced regions, one approx maximum per 128-row wave tile) for every query of the group.  Each
// scanning workgroup derives the launch-wide cutoff itself, per query, by one wave: lanes keep the
// four largest of their 32 sample maxima and the topk-th largest v of those 256 values is found by
// bisection on ballots (topk <= 128 = kMultiMaxTopK).  The 256 values are maxima of 256 DIFFERENT
// wave tiles, so topk distinct rows have approx >= v, exact >= v - margin, and every row of the true
// top-k has approx >= v - 2 margin - slack =: T' — the single-query scan's bound with a v that is at
// most the exact topk-th of all 2048 maxima (equal unless a lane holds five of the top-k): valid.
#pragma once

#include "replica.hip.h"
#include "replica_q8.hip.h"

namespace mi355 {

typedef int hm_v4i __attribute__((ext_vector_type(4)));
typedef int hm_v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float bq_h2f(uint32_t bits16) {
    return static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(bits16)));
}

constexpr int kHmQueries = 32;                     // queries per pass: the columns of one MFMA tile
constexpr int kHmBlock = 512;
constexpr int kHmWaves = kHmBlock / 64;
#ifndef MI355_HM_CHUNKS
#define MI355_HM_CHUNKS 2
#endif
constexpr int kHmChunks = MI355_HM_CHUNKS;         // 128-row chunks (one pair load per lane) a wave handles per step
constexpr int kHmStepRows = 128 * kHmChunks;
#ifndef MI355_HM_EXP
#define MI355_HM_EXP 0   // tools/hm_exp.sh builds the timing experiments 1 ... 3 under gpurun_out/ (never the product library)
#endif
constexpr int kHmStage = 128;                      // staged (query, row) candidates per wave
constexpr int kHmDenseHits = 12;                   // more hits than this in ONE MFMA: the step takes the exact chain whole
constexpr int kHmPrivate = 4;                      // keys per (wave, query) kept in the wave's own list before the shared one is touched
constexpr int kHmKeyCap = 192;                     // kept keys per query and workgroup
constexpr int kHmKeysPerLane = kHmKeyCap / 64;
static_assert(kMultiMaxTopK + 64 <= kHmKeyCap, "a wave-level append of 64 keys always fits after an exact compaction");
// A batch's sample buffer: [query][<= 2048 tagged sample maxima], then one tagged NEIGHBOURHOOD bound per query
// (handoff.hip.h: the exact topk-th best score among the rows around the query's excluded row; 0 = none).
constexpr int kHmNbhdBase = kHmQueries * kHalfSeedMaxGrid * kHalfSeedWaves;
constexpr int kHmSampleSlots = kHmNbhdBase + kHmQueries;

struct HalfMultiArg {
    float q[kHmQueries][kDim];            // query i by value — or, when bit i of ptr_mask is set, q[i][0..1] hold a POINTER to
                                          // its 12 floats (a resident row, possibly of another shard): 48 B per query either way,
                                          // so that two batches (this one and the next one's sample) fit one kernel's arguments
    long long exclude[kHmQueries];        // global row to skip, -1 = none
    uint32_t ptr_mask;
    float margin;                         // error bound the pre-filter may claim on this device (set by the host)
    const float* anchors;                 // the handle's anchor table (handoff.hip.h: queries that exclude no row of this shard), or null
};

__host__ __device__ inline void hm_set_pointer(HalfMultiArg& arg, int i, const float* p) {
    union { const float* p; float f[2]; } u;
    u.p = p;
    arg.q[i][0] = u.f[0];
    arg.q[i][1] = u.f[1];
    arg.ptr_mask |= 1u << i;
}

__device__ __forceinline__ void hm_load_query(const HalfMultiArg& arg, int i, float (&q)[kDim]) {
    if ((arg.ptr_mask >> i) & 1u) {
        // The pointer's two words, put together in registers (a union here became an 8-byte stack slot that the
        // backend never touched but still reserved: every dispatch of the kernel then set up scratch), and read
        // through an explicitly GLOBAL pointer: with a generic one the compiler folds the two branches into one
        // load through a phi of addresses, which forces the whole 1800-byte argument into scratch.
        typedef const float __attribute__((address_space(1))) * global_floats;
        const float lo = arg.q[i][0], hi = arg.q[i][1];
        const uint64_t bits = static_cast<uint64_t>(__builtin_bit_cast(uint32_t, lo)) |
                              (static_cast<uint64_t>(__builtin_bit_cast(uint32_t, hi)) << 32);
        global_floats p = reinterpret_cast<global_floats>(bits);
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = p[j];
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = arg.q[i][j];
    }
}

// The B operand of one batch in LDS (64 lanes x 4 dwords): lane c: k 0..7 of query c; lane 32 + c: k 8..11, the
// threshold slots, 0.  `slots`: what real queries start with in k = 12, 13 (0 for the sample: D = approx; +inf
// for the scan: every row is a candidate until a cutoff is known); a padding column can never hit (-65504).
// Called by threads 0..31; returns the query (for the caller's own bookkeeping).
// (two steps, so that a caller can REQUEST the query before other loads and look at it after them: loads complete in order)
__device__ __forceinline__ void hm_request_query(const HalfMultiArg& arg, int n_queries, int c, float (&q)[kDim]) {
    if (c < n_queries) {
        hm_load_query(arg, c, q);
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = 0.0f;
    }
}
__device__ __forceinline__ bool hm_finish_fragment(const float (&q)[kDim], int n_queries, int c, uint32_t slots, uint4* bfrag, float& qn) {
    qn = query_norm(q);
    const bool ok = c < n_queries && qn >= kBqMinNorm && qn <= kBqMaxNorm;   // false for NaN
    const float inv = ok ? 1.0f / qn : 0.0f;
    uint32_t hp[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) hp[p] = ok ? bq_pack_h2(q[2 * p] * inv, q[2 * p + 1] * inv) : 0u;
    bfrag[c] = make_uint4(hp[0], hp[1], hp[2], hp[3]);
    bfrag[32 + c] = make_uint4(hp[4], hp[5], c < n_queries ? slots : 0x0000fbffu, 0u);
    return ok;
}
__device__ __forceinline__ bool hm_build_fragment(const HalfMultiArg& arg, int n_queries, int c, uint32_t slots, uint4* bfrag,
                                                  float (&q)[kDim], float& qn) {
    hm_request_query(arg, n_queries, c, q);
    return hm_finish_fragment(q, n_queries, c, slots, bfrag, qn);
}

// The two 32-row A operands of one row set of a chunk (a lane's even rows, S = 0, or its odd rows, S = 1);
// rows that are out of range or special are zeroed (their D is then -T', resp. 0 in the sample) and reported.
struct HmTiles {
    bq_h8 a0, a1;          // the rows of lanes 0..31, of lanes 32..63
    uint64_t special;      // lanes whose row is special (tiny / huge / inf / NaN: exact chain for every query)
};

__device__ __forceinline__ HmTiles hm_make_tiles(const HalfTile& t, int S, uint32_t chunk_row, uint32_t n32) {
    const int lane = threadIdx.x & 63;
    uint32_t p0 = S ? t.t1.z : t.t0.x, p1 = S ? t.t1.w : t.t0.y, p2 = S ? t.t2.x : t.t0.z;
    uint32_t p3 = S ? t.t2.y : t.t0.w, p4 = S ? t.t2.z : t.t1.x, p5 = S ? t.t2.w : t.t1.y;
    const uint32_t row = chunk_row + 2u * lane + S;
    const bool in_range = row < n32;
    const bool is_special = in_range && p0 == kBqNaN2;
    const bool keep = in_range && !is_special;
    p0 = keep ? p0 : 0u; p1 = keep ? p1 : 0u; p2 = keep ? p2 : 0u;
    p3 = keep ? p3 : 0u; p4 = keep ? p4 : 0u; p5 = keep ? p5 : 0u;
    // k = 12, 13 multiply the threshold slots of B by 1.0 for ALL rows; k = 14, 15 are zero
    const auto s0 = __builtin_amdgcn_permlane32_swap(p0, p4, false, false);
    const auto s1 = __builtin_amdgcn_permlane32_swap(p1, p5, false, false);
    const auto s2 = __builtin_amdgcn_permlane32_swap(p2, 0x3c003c00u, false, false);
    const auto s3 = __builtin_amdgcn_permlane32_swap(p3, 0u, false, false);
    uint4 aw0, aw1;
    aw0.x = s0[0]; aw0.y = s1[0]; aw0.z = s2[0]; aw0.w = s3[0];
    aw1.x = s0[1]; aw1.y = s1[1]; aw1.z = s2[1]; aw1.w = s3[1];
    HmTiles r;
    r.a0 = __builtin_bit_cast(bq_h8, aw0);
    r.a1 = __builtin_bit_cast(bq_h8, aw1);
    r.special = __ballot(is_special);
    return r;
}

__device__ __forceinline__ int hm_tile_max(const bq_f16v& d) {   // max over the 16 results of one MFMA, on the bit patterns
    auto bits = [&](int i) { return static_cast<int>(__float_as_uint(d[i])); };
    auto max3 = [](int x, int y, int z) { return max(max(x, y), z); };
    const int t0 = max3(bits(0), bits(1), bits(2));
    const int t1 = max3(bits(3), bits(4), bits(5));
    const int t2 = max3(bits(6), bits(7), bits(8));
    const int t3 = max3(bits(9), bits(10), bits(11));
    const int t4 = max3(bits(12), bits(13), bits(14));
    return max(max3(t0, t1, t2), max3(t3, t4, bits(15)));
}

// ---- the sample: [query][region * 8 + wave] ordered-u32 approx maxima -----------------------------
// The sample of replica.hip.h — up to 256 evenly spaced regions, one maximum per wave tile — for every query of a
// batch at once, on the matrix core: a wave takes 128-row chunks = four MFMAs each against the batch's B fragment
// with ZERO threshold slots (D = approx), a max over the 64 results a lane holds for its query column, the two
// half-waves combined.  Masked rows (special, past the end) contribute an approx of exactly 0 and the query's own
// row is not masked at all, so only POSITIVE maxima are published (0 = nothing usable) and the cutoff is taken from
// the (topk + 1)-th largest: topk + 1 tiles with a positive maximum >= v hold topk + 1 distinct genuine rows with
// approx >= v, at least topk of them not the excluded one.
// HOW MUCH is sampled depends on the batch: a region is 1024 << log2_mult rows (wave w: chunks w, 8 + w, ... of it,
// ONE maximum over them, so the sample stays 2048 values per query whatever its size).  The rows a cutoff lets through
// to the exact chain are about topk / (sampled fraction) plus the margin band — ~5 100 per query from the 2.6 % of
// 10 M rows that 1024-row regions are, ~1 400 from 10 % — and they are paid per QUERY while the sample is paid once per
// BATCH; the host picks log2_mult from the batch size (hm_sample_log2).  Regions are >= 1024 << log2_mult rows apart
// (host), so no row is sampled twice.
// Workgroup `first`, `first + every`, ... of the regions; called by all 512 threads.  kDepth chunks are in flight per wave.
// `make_fragment` builds the batch's B fragment at `bfrag` (and ends with a barrier); it is called AFTER the first chunks
// have been requested — the caller has requested the queries before calling this, so that the fragment's inputs come
// back ahead of the cold rows (loads complete in order).
template <int kDepth, typename MakeFragment>
__device__ __forceinline__ void hm_sample_regions(const uint4* __restrict__ half, int64_t n, int64_t stride_rows, int regions,
                                                  int first, int every, const uint4* bfrag, int n_queries,
                                                  unsigned long long* __restrict__ seed_vals, uint32_t epoch, int log2_mult,
                                                  int debug_skip, MakeFragment&& make_fragment) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t n_pairs = (n + 1) >> 1;
    const uint32_t n32 = static_cast<uint32_t>(n);
    const bq_f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t per_query = static_cast<int64_t>(regions) * kHalfSeedWaves;
    const int mult = 1 << log2_mult;
    const int mine = first < regions ? (regions - first + every - 1) / every : 0;
    const int total = mine << log2_mult;   // this wave's chunks: item i = chunk (i & (mult - 1)) of its region i >> log2_mult
    auto region_of = [&](int i) { return first + (i >> log2_mult) * every; };
    auto chunk_of = [&](int i) { return (i & (mult - 1)) * kHalfSeedWaves + wave; };
    auto load = [&](HalfTile& t, int i) {
        int64_t pair = ((static_cast<int64_t>(region_of(i)) * stride_rows) >> 1) + chunk_of(i) * 64 + lane;
        pair = pair < n_pairs ? pair : n_pairs - 1;
        const uint4* p = half + pair * 3;
        t.t0 = p[0];
        t.t1 = p[1];
        t.t2 = p[2];
    };
    HalfTile t[kDepth];
#pragma unroll
    for (int k = 0; k < kDepth; ++k)
        if (k < total) load(t[k], k);
    make_fragment();
    const bq_h8 B = __builtin_bit_cast(bq_h8, bfrag[lane]);
    int m = static_cast<int>(0x80000000u);
    for (int i0 = 0; i0 < total; i0 += kDepth) {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            const int i = i0 + k;
            if (i >= total) break;   // uniform
            const HalfTile cur = t[k];
            if (i + kDepth < total) load(t[k], i + kDepth);   // later chunks are on their way while this one is reduced
            const int g = region_of(i);
            const uint32_t chunk_row = static_cast<uint32_t>(((static_cast<int64_t>(g) * stride_rows) & ~1ll) + chunk_of(i) * 128);
#pragma unroll
            for (int S = 0; S < 2; ++S) {
                const HmTiles a = hm_make_tiles(cur, S, chunk_row, n32);
                const bq_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.a0, B, zero, 0, 0, 0);
                const bq_f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.a1, B, zero, 0, 0, 0);
                m = max(m, max(hm_tile_max(D0), hm_tile_max(D1)));
            }
            if ((i & (mult - 1)) != mult - 1) continue;   // uniform: more chunks of this region to come
            m = max(m, __shfl_xor(m, 32));   // lanes c and 32 + c hold the two row halves of query c
            // a positive float's bits are a positive int; its ordered image sets the top bit
            const uint32_t v = m > 0 ? (static_cast<uint32_t>(m) | 0x80000000u) : 0u;
            // written THROUGH to device scope, under the batch's epoch: the last seed rider of the same launch may read it
            // (hoisted cutoffs; replica.hip.h, "hand-offs that fail safe")
            if (lane < n_queries && g >= debug_skip)
                __hip_atomic_store(&seed_vals[lane * per_query + static_cast<int64_t>(g) * kHalfSeedWaves + wave], tag_value(epoch, v),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m = static_cast<int>(0x80000000u);
        }
    }
}

// The neighbourhood bounds of a batch (handoff.hip.h): queries first, first + every, ... of the batch by this
// workgroup, one after the other, each to its slot behind the sample values under the batch's epoch.  Read by the NEXT
// launch on the stream (the pass), so plain stores do.  Called by all kHmBlock threads.
__device__ __forceinline__ void hm_nbhd_queries(const float* __restrict__ feats, int64_t n, int64_t row_base, const HalfMultiArg& arg,
                                                int n_queries, int first, int every, int topk, uint32_t epoch,
                                                unsigned long long* __restrict__ seed_vals, int* s_scratch /* Nbhd<kHmBlock>::kScratch ints */) {
    for (int j = first; j < n_queries; j += every) {   // uniform
        if (j != first) __syncthreads();   // the select before is done with the shared memory
        float q[kDim];   // (the query first: with the rows requested ahead of it the riding variant of the pass kernel spilled)
        hm_load_query(arg, j, q);
        const uint32_t v = nbhd_bound<kHmBlock>(feats, n, row_base, arg.exclude[j], q, query_norm(q), topk, s_scratch, arg.anchors);
        if (threadIdx.x == 0) seed_vals[kHmNbhdBase + j] = tag_value(epoch, v);
    }
}

// ---- shared memory of one scanning workgroup ---------------------------------------------------------
struct alignas(16) HalfMultiSmem {
    uint4 bfrag[64];                            // the B operand: lane c: k 0..7 of query c; lane 32 + c: k 8..11, -T' (hi, lo), 0
    uint64_t keys[kHmQueries][kHmKeyCap];       // exact keys kept so far, per query (guarded by lock[q])
    uint2 stage[kHmWaves][kHmStage];            // (query, local row) candidates noted by each wave
    uint64_t pkeys[kHmWaves][kHmQueries][kHmPrivate];   // each wave's own first few keys per query: appended without a lock
    int pcount[kHmWaves][kHmQueries];           // (may count past kHmPrivate: the surplus went to the shared list)
    int hist[kHmWaves][256];                    // scratch of the wave-level radix select
    float qf[kHmQueries][kDim];                 // the fp32 queries of the exact chain
    float qn[kHmQueries];
    float cut[kHmQueries];                      // the cutoff currently in the B fragment (-inf: every row is a candidate)
    unsigned long long thr[kHmQueries];         // a key must be > thr to be kept (only ever rises)
    long long excl[kHmQueries];
    int nkeys[kHmQueries];
    int lock[kHmQueries];
    uint32_t ok[kHmQueries];
    int rescored;
    float margin;                               // the error bound this launch claims (HalfMultiArg::margin)
    // the 8-bit front end (scan_half_multi_kernel<.., true>): query c's B operand bytes — k 0..11 = round(127 q^_j),
    // k 12..15 = the integer threshold (hm_q8_slots) — and the bound of its 8-bit approx
    uint4 bfrag8[kHmQueries];
    float m8[kHmQueries];
    int q8;
    uint32_t select_here;                       // bit q: nobody left a cutoff for query q under this batch's epoch — this workgroup selects it
};

// -T' as the fp16 pair (hi, lo) of the B fragment's threshold slots.  T' = -inf ("every row is a
// candidate") becomes +inf: D = approx + inf = +inf, sign bit clear.
__device__ __forceinline__ uint32_t hm_threshold_slots(float cut) {
    if (!(cut > -3.0e38f)) return 0x00007c00u;   // (hi, lo) = (+inf, 0)
    const _Float16 hh = static_cast<_Float16>(-cut);
    const float hi = static_cast<float>(hh);
    return bq_pack_h2(hi, -cut - hi);
}

// ---- the 8-bit front end ----------------------------------------------------------------------------------
// Rows come from the 8-bit replica (replica_q8.hip.h: signed bytes r_j in [-127, 127], |r_j / 127 - r^_j| <= 1/254),
// the query is quantised the same way (qh_j = round(127 q^_j)), and v_mfma_i32_32x32x32_i8 forms
//     D = sum_j r_j qh_j + 127 b12 + 127 b13 + b14 ,      approx8 = sum_j r_j qh_j / 127^2 ,
// exactly, for 32 rows x 32 queries per instruction: a lane's 16 operand bytes are ITS row (12 signed bytes)
// + the constants (127, 127, 1, 0), so the A operand is the load itself — no shuffles — and half the
// bytes of the fp16 front end are streamed.  |approx8 - r^ . q^| <= (l1(r^) + l1(q^)) / 254 + 12 / 254^2 with
// l1(r^) <= sqrt(12): m8(q) = (l1(q^) + 3.4642) / 254 + 2.2e-4 (<= 0.0276).  A row is out for query q if its exact
// score is below L(q) = cut(q) + margin16 + slack (cut = the fp16 cutoff T' this kernel keeps anyway: L is the
// exact lower bound of the top-k threshold it was derived from), so it is a candidate iff approx8 >= L - m8, i.e.
// D >= 0 with 127 b12 + 127 b13 + b14 = -(floor((L - m8) 127^2) - 1).  That margin is 25x the fp16 one and lets
// ~0.3 % of the (row, query) pairs through, so a candidate is first re-checked against its fp16 row (24 B, bound
// margin16: hm_resolve_stage) and only what survives that takes the exact chain.
__device__ __forceinline__ uint32_t hm_q8_slots(float t8) {   // t8 = L - m8; -inf: every row is a candidate
    if (!(t8 > -3.0e38f)) return 0x00007f7fu;                 // (b12, b13, b14) = (127, 127, 0): +32258 > any |sum|
    float xf = 1.0f - __builtin_floorf(t8 * 16129.0f);
    xf = xf > 32258.0f ? 32258.0f : (xf < -32258.0f ? -32258.0f : xf);
    const int x = static_cast<int>(xf);
    const int s127 = x / 127;                                 // truncating: |x - 127 s127| < 127
    const int rem = x - 127 * s127;
    const int b12 = s127 / 2, b13 = s127 - b12;               // |s127| <= 254
    return (static_cast<uint32_t>(b12) & 0xffu) | ((static_cast<uint32_t>(b13) & 0xffu) << 8) | ((static_cast<uint32_t>(rem) & 0xffu) << 16);
}
constexpr uint32_t kHmQ8NeverHit = 0x00008181u;               // (-127, -127, 0): -32258 < any sum
constexpr uint32_t kHmQ8RowConst = 0x00017f7fu;               // the A operand's k 12..15: (127, 127, 1, 0)

// Query c's operand bytes and bound (threads 0..31, beside hm_build_fragment).
__device__ __forceinline__ void hm_build_fragment8(HalfMultiSmem& sm, int c, bool real, bool ok, const float (&q)[kDim], float qn) {
    const float inv = ok ? 1.0f / qn : 0.0f;
    uint32_t w[3] = {0u, 0u, 0u};
    float l1 = 0.0f;
#pragma unroll
    for (int j = 0; j < kDim; ++j) {
        const float u = q[j] * inv;
        l1 += __builtin_fabsf(u);
        int k = static_cast<int>(__builtin_rintf(u * 127.0f));
        k = k > 127 ? 127 : (k < -127 ? -127 : k);
        w[j >> 2] |= (static_cast<uint32_t>(k) & 0xffu) << (8 * (j & 3));
    }
    sm.m8[c] = (l1 + 3.4642f) * kQ8Step * (1.0f + 1e-5f) + 2.2e-4f;
    // a real query starts with "every row is a candidate"; a padding column can never hit
    sm.bfrag8[c] = make_uint4(w[0], w[1], w[2], real ? 0x00007f7fu : kHmQ8NeverHit);
}

// The launch-wide cutoff of one query from its sample maxima, by one wave (see the header).
// Values under another epoch than the reader's count as absent (replica.hip.h, "hand-offs that fail safe").
template <bool kSameLaunch = false /* the values were written by other workgroups of THIS launch: read past the L2 */,
          int kBurstMax = 16 /* requests in flight per lane: 32 = all of them, one round trip (a sample launch has the
                                registers for it; the scanning kernel does not) */>
__device__ __forceinline__ float hm_seed_cutoff(const unsigned long long* vals, int n_seed, int topk, float margin, uint32_t epoch) {
    const int lane = threadIdx.x & 63;
    // the lane's (up to 32) sample maxima, sixteen requests in flight at a time; each goes straight into the lane's
    // four largest (an insertion network per value)
    constexpr int kPer = kHalfSeedMaxGrid * kHalfSeedWaves / 64;
    constexpr int kBurst = kPer < kBurstMax ? kPer : kBurstMax;
    static_assert(kPer % kBurst == 0, "whole bursts");
    uint32_t m1 = 0u, m2 = 0u, m3 = 0u, m4 = 0u;
#pragma unroll
    for (int j0 = 0; j0 < kPer; j0 += kBurst) {
        // Unconditional loads (a conditional load per element made the compiler carry the burst as one 1024-bit
        // register tuple and spill it); what lies past n_seed is masked below.  In bounds because every sample
        // buffer is allocated for kHmQueries x kHalfSeedMaxGrid x kHalfSeedWaves entries whatever n_seed is.
        const unsigned long long* const p = vals + lane + 64 * j0;
        unsigned long long t[kBurst];
#pragma unroll
        for (int j = 0; j < kBurst; ++j)
            t[j] = kSameLaunch ? __hip_atomic_load(&p[64 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p[64 * j];
#pragma unroll
        for (int j = 0; j < kBurst; ++j) {
            uint32_t v = lane + 64 * (j0 + j) < n_seed ? untag_value(t[j], epoch) : 0u, w;
            w = v < m1 ? v : m1; m1 = v > m1 ? v : m1; v = w;
            w = v < m2 ? v : m2; m2 = v > m2 ? v : m2; v = w;
            w = v < m3 ? v : m3; m3 = v > m3 ? v : m3; v = w;
            m4 = v > m4 ? v : m4;
        }
    }
    // the largest v with |{values >= v}| >= topk, by bisection (uniform: ballots and scalar counts)
    uint32_t lo = 0u, hi = 0xffffffffu;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1) + ((hi - lo) & 1u);   // ceil of the midpoint: > lo
        const int c = __popcll(__ballot(m1 >= mid)) + __popcll(__ballot(m2 >= mid)) + __popcll(__ballot(m3 >= mid)) +
                      __popcll(__ballot(m4 >= mid));
        if (c >= topk) lo = mid; else hi = mid - 1u;
    }
    if (lo == 0u) return -__builtin_inff();   // fewer than `topk` usable (positive) maxima: no seed
    return ordered_to_score(lo) - 2.0f * margin - kBqSlack;
}

// Per-query spin lock in LDS, taken by a whole wave (lane 0 spins; the holder is another wave of the
// workgroup, which never waits for anything while it holds the lock).
__device__ __forceinline__ void hm_lock(int* lock) {
    if ((threadIdx.x & 63) == 0) {
        while (atomicCAS(lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void hm_unlock(int* lock) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) atomicExch(lock, 0);
}

// Wave-level append to query q0's key list (q0 wave-uniform): the lanes with `mine` hold one exact key each.
// Under the query's lock: the kept keys are cut back to topk when the new ones would not fit (or pile up),
// which raises the query's threshold and, through the B fragment, its cutoff.
__device__ __forceinline__ void hm_append_locked(HalfMultiSmem& sm, int q0, bool mine, uint64_t key, int topk) {
    const float margin = sm.margin;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // an append of up to 64 keys must fit behind the kept ones: exact cut when topk is large
    const bool exact_cut = topk + 32 + 64 > kHmKeyCap;
    hm_lock(&sm.lock[q0]);
    int nk = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&sm.nkeys[q0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    uint64_t t = __hip_atomic_load(&sm.thr[q0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint64_t t_before = t;
    bool keep = mine && key > t;
    uint64_t mk = __ballot(keep);
    if (mk) {
        if (nk + __popcll(mk) > kHmKeyCap) {   // uniform: cut the kept keys back first
            const uint64_t tn = wave_compact_reg<kHmKeysPerLane>(sm.keys[q0], nk, topk, exact_cut, sm.hist[wave]);
            t = tn > t ? tn : t;
            keep = mine && key > t;
            mk = __ballot(keep);
        }
        if (keep) sm.keys[q0][nk + lanes_below(mk)] = key;
        nk += __popcll(mk);
        if (nk > topk + 32) {   // uniform: tighten the threshold for what is still to come
            const uint64_t tn = wave_compact_reg<kHmKeysPerLane>(sm.keys[q0], nk, topk, exact_cut, sm.hist[wave]);
            t = tn > t ? tn : t;
        }
        if (lane == 0) {
            __hip_atomic_store(&sm.nkeys[q0], nk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (t > t_before) {
                __hip_atomic_store(&sm.thr[q0], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (sm.ok[q0]) {
                    const float local_cut = ordered_to_score(static_cast<uint32_t>(t >> 32)) - margin - kBqSlack;
                    if (local_cut > sm.cut[q0]) {
                        sm.cut[q0] = local_cut;
                        reinterpret_cast<uint32_t*>(&sm.bfrag[32 + q0])[2] = hm_threshold_slots(local_cut);
                        if (sm.q8) reinterpret_cast<uint32_t*>(&sm.bfrag8[q0])[3] = hm_q8_slots(local_cut + margin + kBqSlack - sm.m8[q0]);
                    }
                }
            }
        }
    }
    hm_unlock(&sm.lock[q0]);
}

// The wave's staged candidates -> exact keys -> the workgroup's per-query key lists.  Returns the rows that took
// the exact chain.  half16 != null (8-bit front end): a candidate is first re-checked against its row of the fp16
// replica — approx16 < cut rules it out exactly as the fp16 front end would have — unless bit 31 of its query
// word says that the row is special (exact chain, always).
constexpr uint32_t kHmStageExact = 0x80000000u;
__device__ __forceinline__ int hm_resolve_stage(HalfMultiSmem& sm, int staged, const float* __restrict__ feats, int64_t row_base,
                                                int topk, const uint32_t* __restrict__ half16 = nullptr) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint2* stage = sm.stage[wave];
    int n_exact = 0;
    for (int e0 = 0; e0 < staged; e0 += 64) {
        bool have = e0 + lane < staged;
        const uint2 e = have ? stage[e0 + lane] : make_uint2(0u, 0u);
        const int q = static_cast<int>(e.x & ~kHmStageExact);
        if (half16) {
            const uint2* hp = reinterpret_cast<const uint2*>(half16 + static_cast<int64_t>(e.y) * 6);   // 24 B rows, 8 B aligned
            const uint2 h0 = hp[0], h1 = hp[1], h2 = hp[2];
            const float inv = sm.ok[q] ? 1.0f / sm.qn[q] : 0.0f;
            float acc = 0.0f;
            auto two = [&](uint32_t w, int j) {
                acc = __builtin_fmaf(bq_h2f(w & 0xffffu), sm.qf[q][j] * inv, acc);
                acc = __builtin_fmaf(bq_h2f(w >> 16), sm.qf[q][j + 1] * inv, acc);
            };
            two(h0.x, 0); two(h0.y, 2); two(h1.x, 4); two(h1.y, 6); two(h2.x, 8); two(h2.y, 10);
            // NaN rows of the fp16 replica give acc = NaN: !(NaN < cut) keeps them; an invalid query has cut = -inf
            const bool keep = (e.x & kHmStageExact) != 0u || h0.x == kBqNaN2 || !sm.ok[q] || !(acc < sm.cut[q]);
            have = have && keep;
            if (!__ballot(have)) continue;   // uniform
        }
        n_exact += __popcll(__ballot(have));
        const Row r = load_row(feats, have ? static_cast<int64_t>(e.y) : static_cast<int64_t>(0));   // idle lanes re-read row 0: one cached line
        float qv[kDim];
#pragma unroll
        for (int j = 0; j < kDim; ++j) qv[j] = sm.qf[q][j];
        const float s = cosine_score(qv, sm.qn[q], r);
        const int64_t g = row_base + e.y;
        uint64_t key = pack_key(s, static_cast<uint32_t>(g));
        if (g == sm.excl[q]) key = 0;
        // the threshold only ever rises: a stale value lets a key through to the locked check, nothing more
        const uint64_t seen = __hip_atomic_load(&sm.thr[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool pass = have && key > seen;
        // first the wave's own list of that query: no other wave touches it, so no lock — one LDS atomic per key
        if (pass) {
            const int slot = atomicAdd(&sm.pcount[wave][q], 1);
            if (slot < kHmPrivate) {
                sm.pkeys[wave][q][slot] = key;
                pass = false;
            }
        }
        uint64_t rem = __ballot(pass);
        while (rem) {   // uniform, rare: the keys that did not fit there, one query at a time, to the shared list
            const int src = __ffsll(static_cast<long long>(rem)) - 1;
            const int q0 = __builtin_amdgcn_readlane(q, src);
            const bool mine = pass && q == q0;
            rem &= ~__ballot(mine);
            hm_append_locked(sm, q0, mine, key, topk);
        }
    }
    return n_exact;
}

// A step whose candidates did not fit the staging buffer (no usable cutoff, special rows everywhere — or simply a
// step that lies INSIDE a query's cluster on a catalogue sorted by genre: 256 candidates for that one query): the
// step's rows through the exact chain, no pre-filter, for the queries in `cols` — the columns that noted a candidate in
// this step (every column when a special row was met).  A column that noted none has none: its rows were all ruled out.
__device__ __forceinline__ void hm_exact_step(HalfMultiSmem& sm, const float* __restrict__ feats, int64_t n, int64_t row_base,
                                              int64_t first_row, uint32_t cols, int topk) {
    const int lane = threadIdx.x & 63;
    constexpr int kGroups = kHmStepRows / 64;
    // ALL of the step's rows are requested before the first is scored: one memory round trip under a saturated memory
    // system (~2-3 us), not kGroups of them.  (The caller drops the replica rows it has in flight for its next step and
    // requests them again afterwards: their registers are what makes room for these.)
    Row r[kGroups];
#pragma unroll
    for (int u = 0; u < kGroups; ++u) {
        const int64_t row = first_row + u * 64 + lane;
        r[u] = load_row(feats, row < n ? row : static_cast<int64_t>(0));
    }
    for (uint32_t rest = cols; rest; rest &= rest - 1u) {   // wave-uniform
        const int q0 = __builtin_ctz(rest);
        float qv[kDim];   // (wave-uniform: kept in scalar registers — the vector ones hold the rows)
#pragma unroll
        for (int j = 0; j < kDim; ++j) qv[j] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sm.qf[q0][j])));
        const float qn = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sm.qn[q0])));
        const long long excl = sm.excl[q0];
#pragma unroll
        for (int u = 0; u < kGroups; ++u) {
            const int64_t row = first_row + u * 64 + lane;
            const float s = cosine_score(qv, qn, r[u]);
            const int64_t g = row_base + row;
            uint64_t key = pack_key(s, static_cast<uint32_t>(g));
            if (g == excl) key = 0;
            const uint64_t seen = __hip_atomic_load(&sm.thr[q0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool pass = row < n && key > seen;
            if (__ballot(pass)) hm_append_locked(sm, q0, pass, key, topk);
        }
    }
}

// The end of a sampling workgroup (a sample launch's, or a seed rider's): it arrives, and the LAST ones to arrive select
// the batch's cutoffs from the sample and leave them, tagged, in `cuts` — eight queries (one per wave) per workgroup, so
// a batch of 32 is shared by the last four.  They do not wait for each other: a sample that still misses the values of
// the few workgroups behind it (their slots carry another epoch and count as empty) gives a LOWER (topk + 1)-th largest
// maximum, i.e. a looser cutoff that is just as valid.  The hand-off is replica.hip.h's ("hand-offs that fail safe"):
// write-through sample stores, vmcnt(0), a monotonic arrival counter whose base the host tracks, epoch-tagged values; a
// cutoff nobody wrote reads as "every row is a candidate" in the pass.  Called by all 512 threads; `s_round` is theirs.
template <int kBurstMax = 16>
__device__ __forceinline__ void hm_arrive_and_select(SeedCtl* ctl, unsigned done_base, unsigned total, int n_queries,
                                                     const uint32_t* ok /* shared: the bound can be claimed for query q */,
                                                     const unsigned long long* seed_vals, int n_seed, int topk, float margin,
                                                     uint32_t epoch, unsigned long long* cuts, unsigned* s_round) {
    wait_own_stores();   // this wave's write-through sample stores have completed
    __syncthreads();
    if (threadIdx.x == 0)   // arrival a (0-based) takes round total - 1 - a: the last to arrive round 0, the one before it round 1, ...
        *s_round = done_base + total - 1u - __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    MI355REC_PHASE(3);   // arrived
    const unsigned rounds = static_cast<unsigned>((n_queries + kHmWaves - 1) / kHmWaves);
    const int lane = threadIdx.x & 63;
    for (unsigned r = *s_round; r < rounds; r += total) {   // uniform; (fewer workgroups than rounds: the last ones take several)
        const int qi = static_cast<int>(r) * kHmWaves + static_cast<int>(threadIdx.x >> 6);
        if (qi >= n_queries) continue;   // wave-uniform
        // (a query the bound cannot be claimed for keeps "every row is a candidate")
        const float cut = ok[qi] ? hm_seed_cutoff<true, kBurstMax>(seed_vals + static_cast<int64_t>(qi) * n_seed, n_seed, topk + 1, margin, epoch)
                                 : -__builtin_inff();
        if (lane == 0) cuts[qi] = tag_value(epoch, __float_as_uint(cut));
    }
    if (*s_round < rounds) MI355REC_PHASE(4);   // (a selecting workgroup) selected
}

// The sample launch of a call on its own (and of the head of a stream).  Its last workgroups also select the cutoffs
// (hm_arrive_and_select) — the pass behind it then reads 32 words where each of its 512 workgroups would otherwise
// select from 16 KB of sample values per query (measured at 10 M rows: the pass of 12 queries 49.8 -> 42.4 us, of 32
// queries 62.6 -> 45.8 us).
__global__ __launch_bounds__(kHmBlock) void seed_half_multi_kernel(
    const float* __restrict__ feats, const uint4* __restrict__ half, int64_t n, int64_t row_base, int64_t stride_rows, HalfMultiArg arg,
    int n_queries, int regions /* the first `regions` workgroups sample; the rest take the queries' neighbourhoods */,
    unsigned long long* __restrict__ seed_vals /* [query][regions * kHalfSeedWaves] + the neighbourhood slots, tagged with `epoch` */, uint32_t epoch,
    int log2_mult /* rows per region = 1024 << log2_mult */, SeedCtl* __restrict__ ctl /* arrival counter (null: no cutoffs) */,
    unsigned done_base, unsigned long long* __restrict__ cuts /* [n_queries] tagged cutoffs */, int topk,
    int debug_skip /* test hook (mi355rec_debug_handoff): the regions below this are sampled but not stored */) {
    __shared__ uint4 s_b[64];
    __shared__ uint32_t s_ok[kHmQueries];
    __shared__ unsigned s_round;
    MI355REC_PHASE(0);
    if (static_cast<int>(blockIdx.x) >= regions) {   // uniform: a neighbourhood workgroup (they do not arrive: the PASS reads their slots)
        __shared__ int s_scratch[Nbhd<kHmBlock>::kScratch];
        hm_nbhd_queries(feats, n, row_base, arg, n_queries, static_cast<int>(blockIdx.x) - regions, static_cast<int>(gridDim.x) - regions, topk,
                        epoch, seed_vals, s_scratch);
        return;
    }
    float q_pre[kDim];   // the queries are requested BEFORE the rows (and looked at behind them: loads complete in order)
    if (threadIdx.x < kHmQueries) hm_request_query(arg, n_queries, threadIdx.x, q_pre);
    hm_sample_regions<4>(half, n, stride_rows, regions, blockIdx.x, regions, s_b, n_queries, seed_vals, epoch, log2_mult, debug_skip,
                         [&]() {
                             if (threadIdx.x < kHmQueries) {
                                 float qn;
                                 s_ok[threadIdx.x] = hm_finish_fragment(q_pre, n_queries, threadIdx.x, 0u, s_b, qn) ? 1u : 0u;
                             }
                             __syncthreads();
                             MI355REC_PHASE(1);   // fragment built
                         });
    MI355REC_PHASE(2);   // sampled, values stored
    if (ctl)   // uniform
        hm_arrive_and_select<32>(ctl, done_base, static_cast<unsigned>(regions), n_queries, s_ok, seed_vals, regions * kHalfSeedWaves, topk,
                                 arg.margin, epoch, cuts, &s_round);
}

// What else a launch of a STREAM of batches carries beside the scanners (mi355rec_enqueue_batch_keys_streamed):
// one merging workgroup per query of the PREVIOUS batch (merge_body over that batch's per-workgroup lists, as
// scan_half_kernel's riding merger) and a few "seed riders" that take the sample of the NEXT batch.
struct HmRide {
    const uint64_t* prev_lists;   // [prev_queries][prev_n_lists][prev_topk]
    uint64_t* prev_out;           // [prev_queries][prev_topk]
    int prev_queries;             // queries of the previous batch to merge in this launch (0 = none) ...
    int merge_wgs;                // ... by this many merging workgroups: query q by workgroup q % merge_wgs (~10 us per
                                  // query at 10 M rows, so three or four fit beside a pass; every workgroup that does not
                                  // scan takes a scanner's place among the resident ones)
    int prev_n_lists;
    int prev_topk;
    int seed_wgs;                 // seed riders in this launch (0 = none)
    int nb_wgs;                   // ... and, behind them, workgroups that take the next batch's neighbourhood bounds (query j by
                                  // workgroup j % nb_wgs; ~4 us per query)
    int next_queries;
    int regions;
    long long stride_rows;
    unsigned long long* next_seed_vals;   // [next_queries][regions * 8], tagged with next_epoch
    // the seed rider that finishes LAST turns the sample into the next batch's cutoffs (replica_q8.hip.h's hand-off:
    // write-through stores, a counter, L2-bypassing loads — no fence under the scanners), so that the next launch's
    // scanners start with one load instead of a selection per query (2.5 us per round of eight queries)
    SeedCtl* next_ctl;            // null: the next launch selects its cutoffs itself
    unsigned long long* next_cuts;   // [kHmQueries]: tag_value(next_epoch, bits of the cutoff)
    int next_topk;
    int sample_log2;              // rows per sampled region of the next batch = 1024 << this (hm_sample_regions)
    uint32_t next_epoch;          // of the next batch: the tag of its sample values and of its cutoffs
    uint32_t done_base;           // next_ctl->done before this launch's riders arrive (counted up, never reset)
    int debug_skip;               // test hook (0 in the product): the riders do NOT store regions below this one
};

template <bool kRide>
union HmSmemU {
    HalfMultiSmem scan;
    MergeSmemT<kHmBlock, kRideMaxLists, kHalfRideSurvCap> merge;
};

// block_lists[(slot0 + query) * S + workgroup][topk], each list sorted descending, 0-padded; S = the scanning
// workgroups = gridDim.x - ride.merge_wgs - ride.seed_wgs - ride.nb_wgs.
// seed_vals (when given) also holds the batch's neighbourhood bounds behind the sample values (kHmNbhdBase): EXACT
// scores v with at least topk rows at or above them, so a query starts with thr AT v and a cutoff of at least
// v - margin whatever the sample said.
// kQ8: the rows are streamed from the 8-bit replica `q8` (12 B per row) through the integer matrix core and a
// candidate is re-checked against its fp16 row before the exact chain (see "the 8-bit front end" above); the
// sample of the NEXT batch is still taken over the fp16 replica.
template <bool kRide, bool kQ8 = false>
__global__ __launch_bounds__(kHmBlock, 4) void scan_half_multi_kernel(
    const float* __restrict__ feats, const uint4* __restrict__ half, const uint32_t* __restrict__ q8, int64_t n, int64_t row_base,
    HalfMultiArg arg, int n_queries,
    int slot0, int topk, uint64_t* __restrict__ block_lists, const unsigned long long* __restrict__ seed_vals /* tagged with `epoch` */,
    int n_seed /* sample maxima per query, 0 = none */, unsigned long long* __restrict__ rescored /* [workgroups] */,
    HmRide ride, HalfMultiArg next, const unsigned long long* __restrict__ cuts_ready /* [n_queries] tagged cutoffs left by the launch before, or null */,
    uint32_t epoch /* of this batch */) {
    __shared__ typename std::conditional<kRide, HmSmemU<true>, HalfMultiSmem>::type s_mem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int hh = lane >> 5;
    const unsigned bid = blockIdx.x;
    unsigned nblocks = gridDim.x;   // scanning workgroups
    HalfMultiSmem* smp;
    MI355REC_PHASE(0);
    MI355REC_PHASE_ZERO(2);
    MI355REC_PHASE_ZERO(3);
    MI355REC_PHASE_ZERO(6);
    MI355REC_PHASE_ZERO(7);
    if constexpr (kRide) {
        nblocks = gridDim.x - static_cast<unsigned>(ride.merge_wgs) - static_cast<unsigned>(ride.seed_wgs) - static_cast<unsigned>(ride.nb_wgs);
        if (bid >= nblocks) {
            const int extra = static_cast<int>(bid - nblocks);
            if (extra >= ride.merge_wgs + ride.seed_wgs) {   // the next batch's neighbourhood bounds (read by the next launch)
                hm_nbhd_queries(feats, n, row_base, next, ride.next_queries, extra - ride.merge_wgs - ride.seed_wgs, ride.nb_wgs,
                                ride.next_topk, ride.next_epoch, ride.next_seed_vals, reinterpret_cast<int*>(&s_mem.scan.keys[0][0]));
            } else if (extra < ride.merge_wgs) {   // a merger: queries extra, extra + merge_wgs, ... of the previous batch, one after the other
                for (int pq = extra; pq < ride.prev_queries; pq += ride.merge_wgs) {
                    if (pq != extra) __syncthreads();   // the merge before is done with the shared memory
                    int t = tid;
                    asm volatile("" : "+v"(t));   // (see merge_body)
                    merge_body(s_mem.merge, ride.prev_lists, ride.prev_n_lists, ride.prev_topk, static_cast<int64_t>(ride.prev_topk),
                               static_cast<int64_t>(ride.prev_n_lists) * ride.prev_topk, ride.prev_topk, ride.prev_out,
                               static_cast<int64_t*>(nullptr), static_cast<float*>(nullptr), static_cast<int64_t>(ride.prev_topk),
                               static_cast<int64_t>(pq), static_cast<int64_t>(pq), t);
                }
            } else {                           // a seed rider: its share of the next batch's sample
                uint4* const fb = s_mem.scan.bfrag;
                float q_pre[kDim];   // (requested before the rows: see seed_half_multi_kernel)
                if (tid < kHmQueries) hm_request_query(next, ride.next_queries, tid, q_pre);
                hm_sample_regions<4>(half, n, ride.stride_rows, ride.regions, extra - ride.merge_wgs, ride.seed_wgs, fb,
                                     ride.next_queries, ride.next_seed_vals, ride.next_epoch, ride.sample_log2, ride.debug_skip,
                                     [&]() {
                                         if (tid < kHmQueries) {
                                             float qn;
                                             s_mem.scan.ok[tid] = hm_finish_fragment(q_pre, ride.next_queries, tid, 0u, fb, qn) ? 1u : 0u;
                                         }
                                         __syncthreads();
                                     });
                if (ride.next_ctl)   // uniform: the last riders out select the next batch's cutoffs
                    hm_arrive_and_select(ride.next_ctl, ride.done_base, static_cast<unsigned>(ride.seed_wgs), ride.next_queries,
                                         s_mem.scan.ok, ride.next_seed_vals, ride.regions * kHalfSeedWaves, ride.next_topk, next.margin,
                                         ride.next_epoch, ride.next_cuts, reinterpret_cast<unsigned*>(&s_mem.scan.rescored));
            }
            MI355REC_PHASE(5);   // a merger or a rider is done
            return;
        }
        smp = &s_mem.scan;
    } else {
        (void)ride;
        (void)next;
        smp = &s_mem;
    }
    HalfMultiSmem& sm = *smp;

    const int64_t n_pairs = (n + 1) >> 1;
    const int64_t last_pair = n_pairs - 1;
    const int64_t n_steps = (n_pairs + 64 * kHmChunks - 1) / (64 * kHmChunks);
    const int64_t total_waves = static_cast<int64_t>(nblocks) * kHmWaves;
    // Consecutive steps go to consecutive WORKGROUPS (wave w of workgroup b: steps w * nblocks + b, + total_waves, ...): the
    // chip still reads one moving window of total_waves steps, but a run of steps that lies inside a query's cluster (a
    // catalogue sorted by genre: thirteen steps of 256 rows for a 3300-row cluster, every row a candidate) is spread over
    // as many workgroups, one wave each, instead of queueing on ONE workgroup's per-query lock (measured at 10 M rows, 3000
    // contiguous clusters, 12 queries: 82 us per pass with consecutive steps on consecutive waves).
#if MI355_HM_EXP == 5   // EXPERIMENT (right results): round 4's mapping — the eight waves of a workgroup take eight consecutive steps
    int64_t step = static_cast<int64_t>(bid) * kHmWaves + wave;
#else
    int64_t step = static_cast<int64_t>(wave) * nblocks + bid;
#endif
    auto load_chunk = [&](HalfTile& dst, int64_t st, int u) {
        int64_t pair = (st * kHmChunks + u) * 64 + lane;
        pair = pair < n_pairs ? pair : last_pair;   // unconditional prefetch (see scan_kernel)
        const uint4* p = half + pair * 3;
        dst.t0 = p[0];
        dst.t1 = p[1];
        dst.t2 = p[2];
    };
    // 8-bit front end: a step is the same kHmStepRows rows, as kHmGroups groups of 64 rows, one row (12 B) per lane
    constexpr int kHmGroups = kHmStepRows / 64;
    struct Row8 {
        uint32_t d0, d1, d2;
    };
    auto load_group = [&](Row8& dst, int64_t st, int g) {
        int64_t row = st * kHmStepRows + g * 64 + lane;
        row = row < n ? row : n - 1;
        const uint32_t* p = q8 + row * 3;
        dst.d0 = p[0];
        dst.d1 = p[1];
        dst.d2 = p[2];
    };
    // The queries and the cutoffs left for them are REQUESTED before the first rows: loads complete in order, and behind
    // twelve cold 16-byte loads per lane the B fragment was not ready before 3.5 us into the launch (tools/seed_clock.py
    // on the sample launch, which had the same order) — with the cutoffs' load another round trip behind a barrier.
    float q_pre[kDim];
    unsigned long long cut_pre = 0ull, nb_pre = 0ull;
    if (tid < kHmQueries) {
        hm_request_query(arg, n_queries, tid, q_pre);
        if (cuts_ready && tid < n_queries) cut_pre = cuts_ready[tid];
        if (seed_vals && tid < n_queries) nb_pre = seed_vals[kHmNbhdBase + tid];
    }
    HalfTile T[kQ8 ? 1 : kHmChunks];
    Row8 G[kQ8 ? kHmGroups : 1];
    if constexpr (kQ8) {
#pragma unroll
        for (int g = 0; g < kHmGroups; ++g) load_group(G[g], step, g);
    } else {
#pragma unroll
        for (int u = 0; u < kHmChunks; ++u) load_chunk(T[u], step, u);
    }

    // ---- per-query state, the B fragment and the cutoffs, while the first rows are in flight
    if (tid == 0) {   // (the same wave as the block below: its LDS operations are performed in order)
        sm.rescored = 0;
        sm.margin = arg.margin;
        sm.q8 = kQ8 ? 1 : 0;
        sm.select_here = 0u;
    }
    if (tid < kHmQueries) {
        float qn;
        // a real query starts with "every row is a candidate" (+inf in the threshold slots)
        const bool ok = hm_finish_fragment(q_pre, n_queries, tid, 0x00007c00u, sm.bfrag, qn);
        if constexpr (kQ8) hm_build_fragment8(sm, tid, tid < n_queries, ok, q_pre, qn);
#pragma unroll
        for (int j = 0; j < kDim; ++j) sm.qf[tid][j] = q_pre[j];
        sm.qn[tid] = qn;
        sm.cut[tid] = -__builtin_inff();
        sm.thr[tid] = 0ull;
        sm.excl[tid] = tid < n_queries ? arg.exclude[tid] : -1ll;
        sm.nkeys[tid] = 0;
        sm.lock[tid] = 0;
#pragma unroll
        for (int w = 0; w < kHmWaves; ++w) sm.pcount[w][tid] = 0;
        sm.ok[tid] = ok ? 1u : 0u;
        // The cutoffs: normally left, tagged with this batch's epoch, by the last sampling workgroups (the sample
        // launch's, or the seed riders of the launch before).  A cutoff that is NOT there under this epoch (a hand-off
        // that went wrong: replica.hip.h, "hand-offs that fail safe") is selected below from whatever sample values
        // carry the epoch — what every workgroup did for every query before round 4 — so a broken hand-off costs
        // microseconds, not a scan without a cutoff.
        if (tid < n_queries && ok) {
            const bool have = cuts_ready && static_cast<uint32_t>(cut_pre >> 32) == epoch;
            if (have) {
                const float cut = __uint_as_float(static_cast<uint32_t>(cut_pre));
                sm.cut[tid] = cut;
                reinterpret_cast<uint32_t*>(&sm.bfrag[32 + tid])[2] = hm_threshold_slots(cut);
                if constexpr (kQ8) reinterpret_cast<uint32_t*>(&sm.bfrag8[tid])[3] = hm_q8_slots(cut + arg.margin + kBqSlack - sm.m8[tid]);
            } else if (n_seed > 0) {
                atomicOr(&sm.select_here, 1u << tid);
            }
        }
        // the neighbourhood's bound (valid whatever the pre-filter may claim): at least topk rows score >= it
        if (const uint32_t nbv = tid < n_queries ? untag_value(nb_pre, epoch) : 0u) {
            sm.thr[tid] = (static_cast<unsigned long long>(nbv) << 32) - 1ull;   // (a key AT the bound passes: key > thr)
            const float nb_cut = ordered_to_score(nbv) - arg.margin - kBqSlack;
            if (ok && nb_cut > sm.cut[tid]) {
                sm.cut[tid] = nb_cut;
                reinterpret_cast<uint32_t*>(&sm.bfrag[32 + tid])[2] = hm_threshold_slots(nb_cut);
                if constexpr (kQ8) reinterpret_cast<uint32_t*>(&sm.bfrag8[tid])[3] = hm_q8_slots(nb_cut + arg.margin + kBqSlack - sm.m8[tid]);
            }
        }
    }
    __syncthreads();
    if (const uint32_t todo = sm.select_here) {   // uniform, rare
        for (int qi = wave; qi < n_queries; qi += kHmWaves) {
            if ((todo >> qi) & 1u) {   // wave-uniform
                // the (topk + 1)-th largest: the query's own row may be among the sampled ones (hm_sample_regions)
                float cut = hm_seed_cutoff(seed_vals + static_cast<int64_t>(qi) * n_seed, n_seed, topk + 1, arg.margin, epoch);
                cut = cut > sm.cut[qi] ? cut : sm.cut[qi];   // (the neighbourhood's may be there already)
                if (lane == 0) {
                    sm.cut[qi] = cut;
                    reinterpret_cast<uint32_t*>(&sm.bfrag[32 + qi])[2] = hm_threshold_slots(cut);
                    if constexpr (kQ8) reinterpret_cast<uint32_t*>(&sm.bfrag8[qi])[3] = hm_q8_slots(cut + arg.margin + kBqSlack - sm.m8[qi]);
                }
            }
        }
        __syncthreads();
    }

    MI355REC_PHASE(1);    // fragment and cutoffs in place
    int staged = 0;       // wave-uniform
    int n_rescored = 0;   // wave-uniform (diagnostics)
    uint2* const stage = sm.stage[wave];

    const bq_f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t n32 = static_cast<uint32_t>(n);   // n <= 2^32 - 2 (mi355rec_create)
#ifdef MI355REC_PHASE_CLOCK
    unsigned long long drain_total = 0ull, hot_total = 0ull;
#endif
    bool overflow = false;   // wave-uniform: this step's candidates did not fit the staging buffer ...
    uint32_t step_cols = 0u; // ... and which query columns have noted a candidate in this step so far (bit c = query c)
    // D layout: lane holds column c = lane & 31 (the query) and, in register i, the row that lane
    // (i & 3) + 8 (i >> 2) + 4 (lane >> 5) of the tile's 32 lanes loaded.  `first_row` = row of the tile's
    // lane 0, rows of consecutive lanes are 2 apart (a lane holds a pair).
    // Each lane packs the SIGN bits of its 16 results into a mask (one v_alignbit per register: mask = mask << 1 | sign)
    // and the lanes with a clear bit walk it, one hit per wave-uniform round — instead of a ballot and a branch per
    // result register, which at 32 queries per pass ran for nearly every MFMA (batched.hip.h's push_hits, measured
    // there: the blocks that hold a hit cost three times the others).
    auto push_hits = [&](const bq_f16v& d, uint32_t first_row, int lane0, uint64_t special) {
        uint32_t signs = 0u;
#pragma unroll
        for (int i = 15; i >= 0; --i) signs = __builtin_amdgcn_alignbit(signs, __float_as_uint(d[i]), 31);   // bit i = sign of d[i]
        uint32_t hits = ~signs & 0xffffu;   // D >= +0: approx >= T'
        // opaque on purpose: otherwise the compiler hoists the row ids of all 16 registers out of this rare path
        // into the tile prologue and spills them
        uint32_t lr0 = 4u * static_cast<uint32_t>(hh);
        asm volatile("" : "+v"(lr0));
        {
            const uint64_t any0 = __ballot(hits != 0u);
            step_cols |= static_cast<uint32_t>(any0) | static_cast<uint32_t>(any0 >> 32);   // lanes c and 32 + c hold query column c
        }
        // MORE THAN A DOZEN HITS IN ONE MFMA (32 rows x 32 queries; a seeded pass over spread rows sees 0.03 per query): the
        // tile lies inside a query's cluster (a catalogue sorted by genre) — or there is no cutoff at all.  Either way the
        // step will not fit the staging buffer: it goes through the exact chain for the columns that hit (hm_exact_step),
        // and noting its hits first, one wave-uniform round each — two lanes busy per round when ONE column is dense —
        // would be wasted (measured with stamps: 12-14 us of a hot step's 20, the exact step itself 5).
        if (overflow) return;   // uniform: already decided for this step
        {
            const int n_hits = __builtin_amdgcn_readlane(wave_inclusive_scan(__builtin_popcount(hits)), 63);
            if (n_hits > kHmDenseHits) {   // uniform
                overflow = true;
                return;
            }
        }
        for (uint64_t any = __ballot(hits != 0u); any; any = __ballot(hits != 0u)) {   // wave-uniform rounds
            const int i = hits ? __builtin_ctz(hits) : 0;
            const uint32_t lr = lr0 + static_cast<uint32_t>((i & 3) + 8 * (i >> 2));   // the lane of the tile that loaded the row
            const uint32_t row = first_row + 2u * lr;
            // out-of-range rows and special rows were zeroed in A (D = -T'): never through this path
            const bool hit = hits != 0u && row < n32 && !((special >> (lane0 + lr)) & 1ull);
            hits &= hits - 1u;
            const uint64_t who = __ballot(hit);
            if (who) {
                const int n_hit = __popcll(who);
                if (staged + n_hit > kHmStage) {
                    overflow = true;   // the whole step is redone through the exact chain (hm_exact_step)
                } else {
                    if (hit) stage[staged + lanes_below(who)] = make_uint2(static_cast<uint32_t>(lane & 31), row);
                    staged += n_hit;
                }
            }
        }
    };

    for (; step < n_steps; step += total_waves) {
        const int staged_before = staged;
        step_cols = 0u;
        MI355REC_PHASE_T0(t_step);
        // the fragment is re-read every step (its cutoffs tighten); the barrier keeps the compiler from hoisting it
        asm volatile("" ::: "memory");
        if constexpr (!kQ8) {
#if MI355_HM_EXP == 4   // EXPERIMENT (wrong results): only query 0's column of the B operand is live, the others hold zeros
            uint4 bw = sm.bfrag[lane];
            if ((lane & 31) >= 1) {
                bw.x = bw.y = 0u;
                if (lane < 32) bw.z = bw.w = 0u;
            }
#else
            const uint4 bw = sm.bfrag[lane];
#endif
            const bq_h8 B = __builtin_bit_cast(bq_h8, bw);
    #pragma unroll
            for (int u = 0; u < kHmChunks; ++u) {
                const uint32_t chunk_row = static_cast<uint32_t>((step * kHmChunks + u) * 128);   // row of lane 0's first row
    #pragma unroll
                for (int S = 0; S < 2; ++S) {   // the lanes' even rows, then their odd rows
                    const HmTiles a = hm_make_tiles(T[u], S, chunk_row, n32);
                    if (a.special) {   // uniform, rare: one candidate per (special row, query)
                        step_cols = 0xffffffffu;
                        uint64_t sp = a.special;
                        while (sp) {
                            const int src = __ffsll(static_cast<long long>(sp)) - 1;
                            sp &= sp - 1ull;
                            if (staged + n_queries > kHmStage) {
                                overflow = true;
                            } else {
                                if (lane < n_queries) stage[staged + lane] = make_uint2(static_cast<uint32_t>(lane), chunk_row + 2u * src + S);
                                staged += n_queries;
                            }
                        }
                    }
                    const bq_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.a0, B, zero, 0, 0, 0);
                    const bq_f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.a1, B, zero, 0, 0, 0);
                    const int ma = hm_tile_max(D0), mb = hm_tile_max(D1);
#if MI355_HM_EXP == 1   // EXPERIMENT (wrong results): hits are seen but not extracted
                    if (__ballot(max(ma, mb) >= 0) == 0x12345ull) staged = 1;
#else
                    if (__builtin_expect(__ballot(max(ma, mb) >= 0) != 0ull, 0)) {   // some D >= +0: approx >= T'
                        if (__ballot(ma >= 0)) push_hits(D0, chunk_row + S, 0, a.special);
                        if (__ballot(mb >= 0)) push_hits(D1, chunk_row + 64u + S, 32, a.special);
                    }
#endif
                }
                load_chunk(T[u], step + total_waves, u);   // this chunk's registers are free again: the next step's rows
            }
        } else {
            // every lane fetches its query column's 16 operand bytes; the two half-waves take turns as the K half that is live
            const uint4 fw = sm.bfrag8[lane & 31];
            const hm_v4i frag = {static_cast<int>(fw.x), static_cast<int>(fw.y), static_cast<int>(fw.z), static_cast<int>(fw.w)};
            const hm_v4i none = {0, 0, 0, 0};
            const hm_v4i B_lo = hh ? none : frag;   // k 0..15: the rows lanes 0..31 loaded
            const hm_v4i B_hi = hh ? frag : none;   // k 16..31: the rows lanes 32..63 loaded
            const hm_v16i zero_i = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < kHmGroups; ++g) {
                const uint32_t group_row = static_cast<uint32_t>(step * kHmStepRows + g * 64);   // row of lane 0
                const bool in_range = group_row + static_cast<uint32_t>(lane) < n32;
                const bool is_special = in_range && (G[g].d0 & 0xffu) == kQ8Special;
                const bool keep = in_range && !is_special;
                // the lane's row IS its slice of the A operand: its 12 signed bytes + the constants
                const hm_v4i A = {keep ? static_cast<int>(G[g].d0) : 0, keep ? static_cast<int>(G[g].d1) : 0,
                                  keep ? static_cast<int>(G[g].d2) : 0, static_cast<int>(kHmQ8RowConst)};
                const uint64_t special = __ballot(is_special);
                if (special) {   // uniform, rare: one candidate per (special row, query), straight to the exact chain
                    step_cols = 0xffffffffu;
                    uint64_t sp = special;
                    while (sp) {
                        const int src = __ffsll(static_cast<long long>(sp)) - 1;
                        sp &= sp - 1ull;
                        if (staged + n_queries > kHmStage) {
                            overflow = true;
                        } else {
                            if (lane < n_queries) stage[staged + lane] = make_uint2(static_cast<uint32_t>(lane) | kHmStageExact, group_row + src);
                            staged += n_queries;
                        }
                    }
                }
                const hm_v16i D0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B_lo, zero_i, 0, 0, 0);
                const hm_v16i D1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B_hi, zero_i, 0, 0, 0);
                load_group(G[g], step + total_waves, g);   // the registers are free again: the next step's rows
                auto max16 = [](const hm_v16i& d) {
                    auto max3 = [](int x, int y, int z) { return max(max(x, y), z); };
                    return max(max3(max3(d[0], d[1], d[2]), max3(d[3], d[4], d[5]), max3(d[6], d[7], d[8])),
                               max3(max3(d[9], d[10], d[11]), max3(d[12], d[13], d[14]), d[15]));
                };
                const int ma = max16(D0), mb = max16(D1);
                if (__builtin_expect(__ballot(max(ma, mb) >= 0) != 0ull, 0)) {   // some D >= 0: approx8 >= L - m8
                    auto push8 = [&](const hm_v16i& d, uint32_t first_row, int lane0) {
#pragma unroll
                        for (int g4 = 0; g4 < 16; g4 += 4) {
                            if (!__ballot(max(max(d[g4], d[g4 + 1]), max(d[g4 + 2], d[g4 + 3])) >= 0)) continue;   // wave-uniform
#pragma unroll
                            for (int i = g4; i < g4 + 4; ++i) {
                                if (__ballot(d[i] >= 0)) {   // wave-uniform
                                    uint32_t lr = 4u * static_cast<uint32_t>(hh);
                                    asm volatile("" : "+v"(lr));   // (see push_hits)
                                    lr += static_cast<uint32_t>((i & 3) + 8 * (i >> 2));   // the lane of the tile that loaded the row
                                    const uint32_t row = first_row + lr;
                                    const bool hit = d[i] >= 0 && row < n32 && !((special >> (lane0 + lr)) & 1ull);
                                    const uint64_t who = __ballot(hit);
                                    step_cols |= static_cast<uint32_t>(who) | static_cast<uint32_t>(who >> 32);
                                    if (who) {
                                        const int n_hit = __popcll(who);
                                        if (staged + n_hit > kHmStage) {
                                            overflow = true;
                                        } else {
                                            if (hit) stage[staged + lanes_below(who)] = make_uint2(static_cast<uint32_t>(lane & 31), row);
                                            staged += n_hit;
                                        }
                                    }
                                }
                            }
                        }
                    };
                    if (__ballot(ma >= 0)) push8(D0, group_row, 0);
                    if (__ballot(mb >= 0)) push8(D1, group_row + 32u, 32);
                }
            }
        }
        // The buffer is drained when it is half full, so that only a step with more than 64 candidates of
        // its own can overflow it; such a step forgets what it noted and goes through the exact chain whole.
        if (__builtin_expect(overflow || staged >= kHmStage / 2, 0)) {   // uniform, rare
            MI355REC_PHASE_T0(t_drain);
            if (overflow) MI355REC_PHASE_WAVE_COUNT(7);
#ifdef MI355REC_PHASE_CLOCK   // [3] = the most any wave of the workgroup spent in the matrix-core / hit-noting part of steps that overflowed
            if (overflow) {
                hot_total += t_drain - t_step;
                MI355REC_PHASE_WAVE_MAX(3, hot_total);
            }
#endif
            if (overflow) staged = staged_before;
            n_rescored += hm_resolve_stage(sm, staged, feats, row_base, topk, kQ8 ? reinterpret_cast<const uint32_t*>(half) : nullptr);
            staged = 0;
            if (overflow) {
                const uint32_t cols = step_cols & (n_queries >= 32 ? 0xffffffffu : (1u << n_queries) - 1u);
                hm_exact_step(sm, feats, n, row_base, step * kHmStepRows, cols, topk);
                n_rescored += kHmStepRows * __builtin_popcount(cols);
                overflow = false;
                // (the next step's replica rows again: see hm_exact_step)
                if constexpr (kQ8) {
#pragma unroll
                    for (int g = 0; g < kHmGroups; ++g) load_group(G[g], step + total_waves, g);
                } else {
#pragma unroll
                    for (int u = 0; u < kHmChunks; ++u) load_chunk(T[u], step + total_waves, u);
                }
            }
#ifdef MI355REC_PHASE_CLOCK   // [6] = the most any wave of the workgroup spent in drains and exact steps, [7] = the workgroup's exact steps
            drain_total += wall_clock64() - t_drain;
            MI355REC_PHASE_WAVE_MAX(6, drain_total);
#endif
        }
    }
    MI355REC_PHASE_WAVE_LAST(2);    // the last wave of the workgroup has done its steps
#if MI355_HM_EXP == 2   // EXPERIMENT (wrong results): the candidates are extracted but never scored
    if (staged > 100000) n_rescored = 1;
#else
    n_rescored += hm_resolve_stage(sm, staged, feats, row_base, topk, kQ8 ? reinterpret_cast<const uint32_t*>(half) : nullptr);
#endif
#ifndef MI355REC_PHASE_CLOCK
    MI355REC_PHASE(3);    // ... and its last candidates resolved
#endif
    if (lane == 0) atomicAdd(&sm.rescored, n_rescored);
    __syncthreads();
    if (tid == 0) rescored[bid] += static_cast<unsigned long long>(sm.rescored);   // launches of a handle are stream-ordered

#if MI355_HM_EXP == 3   // EXPERIMENT (wrong results): no lists are written
    if (n_rescored >= 0) return;
#endif
    // ---- every query's best topk of this workgroup, sorted, one wave per query
    static_assert(kHmWaves * kHmPrivate <= 64, "one lane per private key in the final gather");
    for (int qi = wave; qi < n_queries; qi += kHmWaves) {
        int c = __builtin_amdgcn_readfirstlane(sm.nkeys[qi]);   // final: written before the barrier
        if (c + kHmWaves * kHmPrivate > kHmKeyCap) wave_compact_reg<kHmKeysPerLane>(sm.keys[qi], c, topk, true, sm.hist[wave]);
        {   // the waves' own lists join the shared one
            const int w = lane / kHmPrivate, e = lane % kHmPrivate;
            const bool have = w < kHmWaves && e < sm.pcount[w][qi];   // a count past kHmPrivate: the list is full
            const uint64_t key = have ? sm.pkeys[w][qi][e] : 0ull;
            const bool keep = have && key > sm.thr[qi];
            const uint64_t mk = __ballot(keep);
            if (keep) sm.keys[qi][c + lanes_below(mk)] = key;
            c += __popcll(mk);
        }
        if (c > topk) wave_compact_reg<kHmKeysPerLane>(sm.keys[qi], c, topk, true, sm.hist[wave]);
        uint64_t* dst = block_lists + (static_cast<int64_t>(slot0 + qi) * nblocks + bid) * topk;
        wave_rank_and_store(sm.keys[qi], c, dst, topk);
    }
    MI355REC_PHASE(4);    // (wave 0) its share of the lists stored
}

}  // namespace mi355
