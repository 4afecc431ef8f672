// batched.hip.h — the batched ("GEMM-shaped") path of BASELINE configs[4]:
// hundreds to 1024 queries against one pass-pair over the catalogue shard.
//
// The reference has no batched path (one query per process, main.cpp:46-131);
// per query the contract is still recommendByIndex's (Recommender.cu:275-318):
// the exact fp32 cosine of calculateSimilaritiesCPU (:256-273) and the best
// topN rows.  N x Q exact scores are 24 flop each and cannot be materialised
// (12.5 M x 1024 x 4 B = 51 GB), so the work is split into
//
//   1. a CONSERVATIVE pre-filter on the matrix cores: rows and queries are
//      L2-normalised in fp32, rounded to fp16 and multiplied 32 rows x 32 queries
//      at a time with v_mfma_f32_32x32x16_f16 (K = 16: 12 features + 2 slots that
//      carry the query's threshold + 2 zeros).  |approx - exact| <= kBqMargin for
//      every (row, query) pair the bound is claimed for (derivation below);
//   2. the EXACT chain (cosine_score(), kernels.hip.h) on the few hundred rows
//      per query that the pre-filter cannot rule out, and an exact top-N of those.
//
// Results are therefore bit-identical to the single-query path.
//
// Pipeline for one chunk of up to 1024 queries (all launches on one stream, no
// host synchronisation, no allocation):
//   bq_prepare_kernel   queries -> fp16 B fragments (registers of every wave)
//   bq_pass_kernel<NB,false>  pass 1: per (query, workgroup-half) the MAXIMUM approx
//                       cosine over that group's rows            -> gmax
//   bq_select_kernel    per query the (topN+1)-th largest group maximum T: topN+1
//                       distinct rows have approx >= T, so (self excluded) topN rows
//                       have exact >= T - margin; everything in the true top-N has
//                       approx >= T - 2*margin =: T'.  T' goes into the B fragment's
//                       threshold slots (as -T', split hi/lo over two fp16).
//   bq_pass_kernel<NB,true>   pass 2: D = approx - T' straight out of the MFMA; the
//                       sign is the test.  Hits (a few hundred per query) append the
//                       row id to the query's candidate list.
//   bq_finalize_kernel  per query: exact scores of its candidates, exact top-N,
//                       sorted packed keys out.
// Queries (or whole chunks) the bound cannot be claimed for are QUEUED on the
// device and served by the exact multi-query scan (scan_multi_queued_kernel):
// tiny / huge / non-finite query norms, T' <= 0 (fewer than topN+1 groups with a
// clearly positive maximum), more candidates than the per-query list holds (mass ties at the
// threshold), more special rows than kBqSpecialCap.
//
// Rows: "valid" = |row|^2 in [kBqMinNorm2, kBqMaxNorm2] (then, with a valid
// query, |row||q| > 1e-8 so the reference's zero branch is not taken, and no fp32
// sum can overflow in any order).  Exactly-zero rows score 0 against every query
// (Recommender.cu:271) and can never reach a positive threshold: they are
// skipped.  Every other row (tiny norm, huge, inf, NaN) is SPECIAL: it is listed
// once and scored exactly against every query in the finalize step.
//
// Margin.  r^ = r / |r| and q^ = q / |q| in fp32 (relative error < 1e-6 each),
// then x~ = fp16(x) with |x~ - x| <= 2^-11 |x| for normal results and <= 2^-25
// absolute for subnormal ones (2^-14 if they were flushed to zero).  The products
// of fp16 values are exact in the MFMA's fp32 accumulator.  So
//   |sum r~_j q~_j - sum r^_j q^_j| <= (2^-10 + 2^-22) sum |r^_j q^_j|
//                                      + e_sub (sum |r^_j| + sum |q^_j|)
// with sum |r^_j q^_j| <= |r^||q^| <= 1 + 2e-6 (Cauchy-Schwarz) and
// sum |x_j| <= sqrt(12): 9.77e-4 + 2.1e-7 with subnormals kept, 9.77e-4 + 4.2e-4
// if they flush; plus < 6e-6 for the normalisations, the 16-term fp32
// accumulation and the reference chain's own rounding.  gfx950 keeps fp16
// subnormals both in v_cvt_pk_f16_f32 and as MFMA operands (default float mode);
// this is CHECKED on the device when the path is first used
// (bq_selfcheck_kernel): kBqMargin = 1.0e-3 if they are kept, kBqMarginFlush =
// 1.5e-3 otherwise.  tests/test_batched_margin.py checks both bounds on hostile
// data with a numpy model of exactly this arithmetic.
#pragma once

#include "kernels.hip.h"

namespace mi355 {

typedef _Float16 bq_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 bq_h2 __attribute__((ext_vector_type(2)));
typedef float bq_f16v __attribute__((ext_vector_type(16)));

constexpr int kBqMaxBlocks = 32;             // 32 queries per block -> 1024 queries per chunk
constexpr int kBqMaxQueries = kBqMaxBlocks * 32;
// Candidate RECORDS kept per query: a power of two near rows / 64 between these two (the host's choice, engine_batch.hip.h;
// 16 MiB ... 512 MiB per 1024-query chunk).  A record = what ONE lane of pass 2 found for its query in one 32-row sub-tile:
// `row_lo` (low word) and a 16-bit mask (bits 32..47), bit i = row row_lo + (i & 3) + 8 (i >> 2) — up to 16 rows per
// appended word.  On shuffled rows a record holds one row; where a query's neighbours lie next to each other (a
// catalogue sorted by genre: 33 000 rows of a cluster within the fp16 margin of the cutoff) it holds most of sixteen, and
// round 4's one-atomic-per-row lists made pass 2 eight times longer there (1.75 ms against 0.2).
// The passes read the cap from counters[6] where they flush their staging buffers — not a kernel argument: the pass
// kernel's register allocation is what it is.
constexpr int kBqCapMin = 2048;
constexpr int kBqCapMax = 65536;
constexpr int kBqSpecialCap = 1024;          // special rows kept per chunk
constexpr int kBqNbhdRows = 1024;            // rows around a query's excluded row that give its neighbourhood bound (handoff.hip.h)
constexpr int kBqNbProbeChunks = 3;          // chunks that compute it before the host may decide it never wins on this catalogue
constexpr int kBqNbProbeEvery = 32;          // ... after which only every this-many-th chunk does (one win switches it back on)
constexpr int kBqFinalBlock = 256;           // threads of the finalize workgroup
constexpr int kBqFinalChunk = 1536;          // candidate rows scored between two cuts of the finalize workgroup's key buffer
                                             // (with the 16 KiB of expanded rows beside it the workgroup stays under 40 KiB of LDS: four per CU)
constexpr int kBqFinalKeys = kBqFinalChunk + 1024;   // keys that buffer holds: a chunk + what a cut may leave (<= 1024)
constexpr int kBqFinalPerThread = kBqFinalKeys / kBqFinalBlock;
constexpr int kBqFinalRows = kBqFinalBlock * 16;     // rows one batch of records (a record per thread) can expand to
constexpr int kBqFinalAhead = 3;             // rows a finalize thread keeps in flight (a chunk = two rounds of three)
constexpr int kBqFinalRecs = 3;              // candidate records a finalize thread takes at a time (768 per workgroup and lot)
constexpr int kBqPassBlock = 256;            // 4 waves, one per SIMD; 4-5 workgroups per CU
constexpr float kBqMargin = 1.0e-3f;         // fp16 subnormals kept (verified per device by bq_selfcheck_kernel)
constexpr float kBqMarginFlush = 1.5e-3f;    // bound if they were flushed
constexpr float kBqSlack = 4e-6f;            // fp16 hi/lo split of T' and the fused subtraction
constexpr float kBqMinNorm2 = 1.01e-8f;      // |x| >= 1.005e-4 for rows and queries alike
constexpr float kBqMaxNorm2 = 1e36f;
constexpr float kBqMinNorm = 1.005e-4f;
constexpr float kBqMaxNorm = 1e18f;
// Every query's candidate counter sits in its own 128-byte line: the counters of 32 queries in
// ONE line made every returning atomic of a small batch queue up at one L2 channel (measured at
// 10 M rows x 32 queries: ~20 k atomics in a burst as the waves drain, 37 of pass 2's 83 us).
#ifndef MI355_BQ_COUNT_STRIDE
#define MI355_BQ_COUNT_STRIDE 32
#endif
constexpr int kBqCountStride = MI355_BQ_COUNT_STRIDE;
constexpr int kBqGroupsPerBlock = 2;         // the two lane halves of a workgroup stay separate groups

// per-query flags written by prepare / select, read by finalize
// The fp16 replica of the catalogue (replica.hip.h) stores rows in exactly the form the A
// operand wants (normalised, six fp16 pairs); rows the bound is not claimed for are all-NaN.
constexpr uint32_t kBqNaN2 = 0x7e007e00u;   // two fp16 quiet NaNs

constexpr uint32_t kBqFlagOk = 0u;
constexpr uint32_t kBqFlagQueue = 1u;        // serve through the exact multi-query scan
constexpr uint32_t kBqFlagPad = 2u;          // not a query (padding up to a multiple of 32)

__device__ __forceinline__ uint32_t bq_pack_h2(float a, float b) {
    const float __attribute__((ext_vector_type(2))) f = {a, b};
    const bq_h2 h = __builtin_convertvector(f, bq_h2);   // v_cvt_pk_f16_f32, round to nearest even
    return __builtin_bit_cast(uint32_t, h);
}

// ---- does this device keep fp16 subnormals? ----------------------------------------
// out[0] = MFMA(2^-20 (an fp16 subnormal) x 1.0), out[1] = fp16(3e-6) converted back:
// both are non-zero iff neither the conversion nor the matrix core flushes.
__global__ void bq_selfcheck_kernel(float* out) {
    const int lane = threadIdx.x;
    bq_h8 A = {0, 0, 0, 0, 0, 0, 0, 0}, B = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) {
        A[0] = __builtin_bit_cast(_Float16, static_cast<unsigned short>(0x0010));   // 2^-20
        B[0] = __builtin_bit_cast(_Float16, static_cast<unsigned short>(0x3c00));   // 1.0
    }
    const bq_f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const bq_f16v D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, zero, 0, 0, 0);
    const uint32_t packed = bq_pack_h2(3.0e-6f * (1.0f + static_cast<float>(lane)), 0.0f);   // lane 0: 3e-6
    if (lane == 0) {
        out[0] = D[0];
        out[1] = static_cast<float>(__builtin_bit_cast(bq_h2, packed)[0]);
    }
}

// ---- queries -> B fragments -----------------------------------------------------
// B operand of v_mfma_f32_32x32x16_f16: lane l (c = l & 31, h = l >> 5) holds
// B[k = 8h + j][col c], j = 0..7, i.e. 4 dwords.  bfrag[blk][lane][4].
// k = 0..11 the normalised query, k = 12, 13 the threshold (select kernel), 14, 15 zero.
// One query -> its two fragment rows (lo: k 0..7, hi: k 8..11 + zeroed threshold slots), its norm and its flag.
struct BqPrepared {
    uint4 lo, hi;
    float qn;
    uint32_t flag;
};
__device__ __forceinline__ BqPrepared bq_prepare_query(const float* __restrict__ queries, int q, int n_queries) {
    const bool real = q < n_queries;
    float v[kDim];
#pragma unroll
    for (int j = 0; j < kDim; ++j) v[j] = real ? queries[static_cast<int64_t>(q) * kDim + j] : 0.0f;
    BqPrepared p;
    p.qn = query_norm(v);
    const bool ok = real && p.qn >= kBqMinNorm && p.qn <= kBqMaxNorm;   // false for NaN
    const float inv = ok ? 1.0f / p.qn : 0.0f;
    float u[kDim];
#pragma unroll
    for (int j = 0; j < kDim; ++j) u[j] = ok ? v[j] * inv : 0.0f;
    p.lo = make_uint4(bq_pack_h2(u[0], u[1]), bq_pack_h2(u[2], u[3]), bq_pack_h2(u[4], u[5]), bq_pack_h2(u[6], u[7]));
    p.hi = make_uint4(bq_pack_h2(u[8], u[9]), bq_pack_h2(u[10], u[11]), 0u, 0u);
    p.flag = ok ? kBqFlagOk : (real ? kBqFlagQueue : kBqFlagPad);
    return p;
}

// (A launch of its own only for chunks whose pass 1 does not prepare the queries itself: see bq_pass_kernel.)
// The first prep_blocks workgroups prepare the queries.  The launch may carry n_queries more: workgroup prep_blocks + q
// then takes query q's NEIGHBOURHOOD bound (handoff.hip.h: the exact topk-th best score among the 2048 rows around the
// row the query excludes — its own, for recommendByIndex, Recommender.cu:275-318) into nb_vals[q] (ordered-u32 image,
// 0 = none).  bq_select_kernel takes it as one more lower bound on the query's threshold: pass 1's group maxima come
// from every 4th tile dealt round-robin over the waves, and on a catalogue sorted by genre only a handful of its 2048
// groups ever see the query's own cluster.
__global__ __launch_bounds__(256) void bq_prepare_kernel(
    const float* __restrict__ queries, int n_queries, int n_blocks, uint32_t* __restrict__ bfrag,
    float* __restrict__ qnorm, uint32_t* __restrict__ qflags, int* __restrict__ cand_count,
    int* __restrict__ counters /* [0] special rows, [1] queued queries, [2] chunk-wide queue flag,
                                  [3] (tile, query block) pairs pass 2 ran its MFMAs for (diagnostics) */,
    int prep_blocks, const float* __restrict__ feats, int64_t n, int64_t row_base, const long long* __restrict__ exclude /* may be null */,
    int topk, uint32_t* __restrict__ nb_vals /* [n_queries] */, const float* __restrict__ anchors /* the handle's anchor table, or null */) {
    if (static_cast<int>(blockIdx.x) >= prep_blocks) {   // uniform: a neighbourhood workgroup
        __shared__ int s_scratch[Nbhd<256, kBqNbhdRows>::kScratch];
        const int nq = static_cast<int>(blockIdx.x) - prep_blocks;
        // (1024 rows here, not the single queries' 2048: a thousand of these workgroups ride in every chunk's first launch)
        const long long excl = exclude ? exclude[nq] : -1ll;
        float qv[kDim];
        uint32_t v;
        if (nbhd_has_center(n, row_base, excl)) {   // uniform: the rows are requested before the query is loaded
            const Nbhd<256, kBqNbhdRows> nb = nbhd_request<256, kBqNbhdRows>(feats, n, row_base, excl, topk);
#pragma unroll
            for (int j = 0; j < kDim; ++j) qv[j] = queries[static_cast<int64_t>(nq) * kDim + j];
            v = nbhd_finish<256, kBqNbhdRows>(nb, qv, query_norm(qv), topk, s_scratch);
        } else {                                    // a query by value: around its anchor (handoff.hip.h)
#pragma unroll
            for (int j = 0; j < kDim; ++j) qv[j] = queries[static_cast<int64_t>(nq) * kDim + j];
            v = nbhd_bound<256, kBqNbhdRows>(feats, n, row_base, excl, qv, query_norm(qv), topk, s_scratch, anchors);
        }
        if (threadIdx.x == 0) nb_vals[nq] = v;
        return;
    }
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0) {
        counters[0] = 0;
        counters[1] = 0;
        counters[2] = 0;
        counters[3] = 0;
    }
    if (q >= n_blocks * 32) return;
    const BqPrepared p = bq_prepare_query(queries, q, n_queries);
    const int blk = q >> 5, c = q & 31;
    reinterpret_cast<uint4*>(bfrag)[blk * 64 + c] = p.lo;
    reinterpret_cast<uint4*>(bfrag)[blk * 64 + 32 + c] = p.hi;
    qnorm[q] = p.qn;
    qflags[q] = p.flag;
    cand_count[q * kBqCountStride] = 0;
}

// ---- the two passes ---------------------------------------------------------------
// One wave = 64 rows at a time (two 32-row MFMA tiles) against all NB query blocks.
// Lane l loads ROW tile*64 + l whole (3 x dwordx4, every fetched line fully used), sums its
// squares and normalises it in-lane, and packs it to six fp16 pairs p0..p5.  The A
// operand wants lane l (r = l & 31, h = l >> 5) to hold A[row r][k = 8h + j]: k 0..7 of
// row r in the lower half-wave, k 8..15 in the upper.  One v_permlane32_swap per
// register builds BOTH tiles' operands at once: swap(p0, p4) leaves
//   x' = [p0 of rows 0..31  | p4 of rows 0..31 ]  = register 0 of tile 0's operand
//   y' = [p0 of rows 32..63 | p4 of rows 32..63]  = register 0 of tile 1's operand
// likewise swap(p1, p5); swap(p2, ONE) and swap(p3, ZERO) put the constant threshold
// multiplier (k = 12, 13 -> 1.0) and the zero padding into the upper halves.  Four swaps,
// no selects, and the per-row work (norm, 12 multiplies, 6 conversions) is done once per
// row instead of once per half-row as in the first version of this kernel.
// Tiles are dealt round-robin over all waves of the grid, so the chip reads one moving
// window of the matrix.  The B fragments (NB KiB) live in LDS: one ds_read_b128 feeds TWO
// MFMAs; in registers they would cost 128 VGPRs and leave two waves per SIMD, too few to
// cover the MFMA -> VALU latency of the reduction (measured: 454 / 584 us per pass at
// 12.5 M rows x 1024 queries with B in registers; tools/mfma_probe.hip has the
// issue-rate model).
// C/D layout (cdna_hip_programming.md §3): lane holds column c = l & 31 (the query)
// and rows (i & 3) + 8 (i >> 2) + 4 h for register i = 0..15.
// Per MFMA (1024 outputs) the VALU does 8 three-operand integer maxima; with the
// MFMA's own 8 issue cycles that is ~40 cycles per block and SIMD against the 32 of
// the matrix pipe: the passes are VALU-issue bound by construction.
// Pass 2 stages its hits per wave in LDS; a flush hands 64 of them per round trip to
// the global per-query lists.  Wave-synchronous: one wave's LDS operations execute in
// order, no barrier involved.
constexpr int kBqStage = 256;   // entries per wave (2 KiB); a block adds at most 64 per register

// Scheduling experiments of pass 2's inner loop (tools/bq_sched.sh builds them under gpurun_out/):
// MI355_BQ_SCHED & 1: raise the wave's priority while it issues its two MFMAs; & 2: ask the scheduler to interleave the
// MFMAs of a block with the reduction of the block before (one MFMA, eight VALU, one MFMA, the rest).
#ifndef MI355_BQ_SCHED
#define MI355_BQ_SCHED 0
#endif
// The reduction of pass 2's hit test: 0 = the v_max3_i32 tree (the product), 1 = packed signs (an A/B build).
#ifndef MI355_BQ_REDUCE
#define MI355_BQ_REDUCE 0
#endif
#if MI355_BQ_SCHED & 1
#define MI355_BQ_PRIO(p) __builtin_amdgcn_s_setprio(p)
#else
#define MI355_BQ_PRIO(p) \
    do {                 \
    } while (0)
#endif
#if MI355_BQ_SCHED & 2
#define MI355_BQ_INTERLEAVE()                                  \
    do {                                                       \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);     \
    } while (0)
#else
#define MI355_BQ_INTERLEAVE() \
    do {                      \
    } while (0)
#endif

// staged entry: x = query | mask << 16, y = row_lo
__device__ __forceinline__ void bq_flush_stage(const uint2* stage, int staged, int lane, int* __restrict__ cand_count,
                                               uint64_t* __restrict__ cand_recs, int cand_cap) {
    for (int e = lane; e < staged; e += 64) {
        const uint2 qr = stage[e];
        const uint32_t q = qr.x & 0xffffu;
        const int pos = atomicAdd(&cand_count[q * kBqCountStride], 1);
        if (pos < cand_cap) cand_recs[static_cast<int64_t>(q) * cand_cap + pos] = (static_cast<uint64_t>(qr.x >> 16) << 32) | qr.y;
    }
}

template <bool kCollect>
struct BqPassCfg {
    static constexpr int kMinBlocksPerCu = 4;   // waves per SIMD the register budget must allow
};

// kVariant (development A/B only, tools/bqbench.hip; 0 in the product): 4 = synthetic rows
// (no HBM reads).
// kFromReplica: the rows come from the fp16 replica (24 B per row, already normalised and
// packed: `half`, replica.hip.h) instead of the fp32 matrix — half the bytes and none of the
// per-row norm / scale / convert work; the bound is the same because the replica is built
// with this kernel's own arithmetic.
// kTileMax (NB >= 16, rows from the replica): PASS 2 SKIPS WHAT PASS 1 HAS ALREADY RULED OUT.  Pass 1 looks at every
// tile_step-th tile; for those tiles it now also stores, per lane and query block, the maximum approx over the
// lane's 32 rows of the tile (its two MFMA sub-tiles), as fp16: tile_max[visited tile][block pair][lane],
// 4 KiB per visited 64-row tile at NB = 32.  Once the thresholds are known, a (tile, query block) pair can hold a
// candidate only if some lane's maximum reaches its query's T' (both rounded toward zero — rounding is monotone, so
// the comparison of the images never says "below" for a maximum that is not): pass 2 compares the 64 x NB
// stored maxima of a visited tile with the thresholds (one packed 16-bit subtraction per two blocks) and runs the two
// MFMAs and their hit test only for the blocks that pass — about one in eight at top-100 (0.13 candidates per pair).
// Valid because the margin covers ANY summation order of the 16-term sum: a row of the true top-N has approx >= T'
// + kBqSlack in pass 1's sum as in pass 2's.  Pass 2 deals its tiles so that every wave meets visited and
// unvisited tiles alike (k-th tile of wave w: k * waves + (w + k) mod waves).
template <int NB, bool kCollect, int kVariant = 0, bool kFromReplica = false, bool kTileMax = false>
__global__ __launch_bounds__(kBqPassBlock, BqPassCfg<kCollect>::kMinBlocksPerCu) void bq_pass_kernel(
    const float* __restrict__ feats, int64_t n, int64_t n_tiles, int tile_step, const uint32_t* __restrict__ bfrag,
    float* __restrict__ gmax /* [NB][8][2 * grid][4] */, int* __restrict__ cand_count,
    uint64_t* __restrict__ cand_recs /* [query][counters[6]] records */, int* __restrict__ counters,
    uint32_t* __restrict__ special_rows, const uint2* __restrict__ half = nullptr,
    uint4* __restrict__ tile_max = nullptr /* [visited tile][NB / 8][64] */, int max_step = 1 /* pass 2: pass 1's tile_step */,
    const float* __restrict__ qthr = nullptr, const uint32_t* __restrict__ qflags = nullptr,
    // pass 1 can PREPARE THE QUERIES ITSELF (prep_queries given: no bq_prepare_kernel launch in front of it): every
    // workgroup builds the B fragments straight into its LDS from the raw queries, and workgroup 0 also leaves them —
    // with the norms, the flags and the zeroed counters — in global memory for the kernels behind it.  Built for
    // VERDICT r3 item 1(a), measured, and left off (see launch_bq_passes)
    const float* __restrict__ prep_queries = nullptr, int prep_count = 0, float* __restrict__ prep_qnorm = nullptr,
    uint32_t* __restrict__ prep_qflags = nullptr) {
    static_assert(!kTileMax || (NB >= 16 && kFromReplica && kVariant == 0), "tile maxima: replica rows, 16 or 32 query blocks");
    // n_tiles counts 64-row tiles.  tile_step = 1: every tile.  tile_step > 1 (pass 1 only):
    // every tile_step-th tile — a threshold derived from ANY subset of the rows is a valid
    // lower bound; from a quarter of them it lets about four times as many candidates
    // through and costs a quarter of a pass.
    // Pass 2 sits exactly at four workgroups per CU (40 KiB of the CU's 160): with tile maxima the staging buffers are
    // halved to make room for the thresholds' table and for one DUMMY fragment (block NB: threshold slots -65504,
    // nothing ever passes) that pads a tile's list of blocks to an even count.
    constexpr int kStageCap = (kCollect && kTileMax) ? kBqStage / 2 : kBqStage;
    __shared__ uint4 s_b[NB + ((kCollect && kTileMax) ? 1 : 0)][64];   // reused for the group maxima at the end of pass 1
    __shared__ uint2 s_stage[kCollect ? kBqPassBlock / 64 : 1][kCollect ? kStageCap : 1];   // (query, row) per wave
    // (known to the compiler as wave-uniform: the tile index and everything derived from it stay in scalar registers)
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    uint2* const stage = s_stage[kCollect ? wave : 0];
    int staged = 0;   // wave-uniform
    int pairs_done = 0;   // wave-uniform: (tile, query block) pairs this wave ran its MFMAs for (pass 2 with tile maxima)
    const int lane = threadIdx.x & 63;
    const int r = lane & 31;
    const int h = lane >> 5;

    if (!kCollect && kFromReplica && prep_queries) {   // uniform (replica-sourced pass 1 only; NOT used by the product:
                                                        // measured 13 us slower than the 4.4 us launch it saves, mi355rec.hip)
        uint32_t* const bfrag_out = const_cast<uint32_t*>(bfrag);
        for (int q = threadIdx.x; q < NB * 32; q += kBqPassBlock) {
            const BqPrepared p = bq_prepare_query(prep_queries, q, prep_count);
            const int blk = q >> 5, c = q & 31;
            s_b[blk][c] = p.lo;
            s_b[blk][32 + c] = p.hi;
            if (blockIdx.x == 0) {
                reinterpret_cast<uint4*>(bfrag_out)[blk * 64 + c] = p.lo;
                reinterpret_cast<uint4*>(bfrag_out)[blk * 64 + 32 + c] = p.hi;
                prep_qnorm[q] = p.qn;
                prep_qflags[q] = p.flag;
                cand_count[q * kBqCountStride] = 0;
            }
        }
        if (blockIdx.x == 0 && threadIdx.x < 4) counters[threadIdx.x] = 0;
    } else {
        for (int i = threadIdx.x; i < NB * 64; i += kBqPassBlock) (&s_b[0][0])[i] = reinterpret_cast<const uint4*>(bfrag)[i];
    }
    // pass 2 with tile maxima: the thresholds as packed fp16 pairs in the maxima's layout — dword d of query column c
    // holds T' of query c of blocks 2d and 2d + 1, rounded DOWN; +inf for a query that is not served here (queued /
    // padding: its threshold slots hold -65504 and nothing ever passes)
    __shared__ uint32_t s_tq[(kCollect && kTileMax) ? NB / 2 : 1][(kCollect && kTileMax) ? 32 : 1];
    if constexpr (kCollect && kTileMax) {
        for (int i = threadIdx.x; i < NB / 2 * 32; i += kBqPassBlock) {
            const int d = i >> 5, c = i & 31;
            const int q0 = (2 * d) * 32 + c, q1 = q0 + 32;
            const float t0 = qflags[q0] == kBqFlagOk ? qthr[q0] : __builtin_inff();
            const float t1 = qflags[q1] == kBqFlagOk ? qthr[q1] : __builtin_inff();
            // Maxima and thresholds are rounded the SAME way (toward zero: one conversion, nothing else, in pass 1's
            // hot loop); rounding is monotone, so a maximum that reaches T' has an image that reaches the image of T'.
            s_tq[d][c] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(t0, t1));
        }
        if (threadIdx.x < 64)   // the dummy block: no query, -65504 in the threshold slot of every column
            s_b[NB][threadIdx.x] = make_uint4(0u, 0u, threadIdx.x >= 32 ? bq_pack_h2(-65504.0f, 0.0f) : 0u, 0u);
    }
    __syncthreads();

    int mx[kCollect ? 1 : NB];   // running group maxima (bit patterns of floats >= 0)
    if constexpr (!kCollect) {
#pragma unroll
        for (int b = 0; b < NB; ++b) mx[b] = 0;
    }

    const int64_t total_waves = static_cast<int64_t>(gridDim.x) * (kBqPassBlock / 64) * tile_step;
    const int64_t first = (static_cast<int64_t>(blockIdx.x) * (kBqPassBlock / 64) + wave) * tile_step;
    const int64_t last_row = n - 1;
    const float4* base = reinterpret_cast<const float4*>(feats);

    // rows past the end re-read the last row (cached) so the prefetch is unconditional
    auto load_row3 = [&](int64_t tile, float4& a, float4& b, float4& c) {
        int64_t row = tile * 64 + lane;
        row = row < n ? row : last_row;
        const float4* p = base + row * 3;
        a = p[0];
        b = p[1];
        c = p[2];
    };

    auto load_half3 = [&](int64_t tile, uint2& x, uint2& y, uint2& z) {
        int64_t row = tile * 64 + lane;
        row = row < n ? row : last_row;
        const uint2* p = half + row * 3;
        x = p[0];
        y = p[1];
        z = p[2];
    };

    // With only a handful of query blocks a tile is a few hundred cycles of work, far less than
    // a memory round trip, and one tile ahead per wave leaves the pass latency-bound (measured at
    // 10 M rows x 32 queries: 83 us, 2.9 TB/s).  Registers that are the target of a load in
    // flight cannot be rotated, and unrolling this loop would multiply the rare hit path, so for
    // NB <= 8 the replica rows go through a per-wave LDS ring: every kPre-th iteration the wave
    // parks the kPre tiles that have arrived in LDS and sends out the loads of the next kPre;
    // the body reads its tile from LDS by a dynamic slot index.  Wave-private slots, no barrier.
    constexpr bool kStaged = kFromReplica && NB <= 8;
    constexpr int kPre = kStaged ? 4 : 1;
    __shared__ uint2 s_rows[kStaged ? kBqPassBlock / 64 : 1][kStaged ? kPre : 1][3][kStaged ? 64 : 1];
    float4 na, nb, nc;
    uint2 pre[kPre][3];
    if constexpr (kFromReplica) {
#pragma unroll
        for (int d = 0; d < kPre; ++d) load_half3(first + d * total_waves, pre[d][0], pre[d][1], pre[d][2]);
    } else {
        load_row3(first, na, nb, nc);
    }
    int slot = 0;   // wave-uniform: position of `tile` in its group of kPre
    // pass 2 with tile maxima: the stored maxima of the NEXT tile (if pass 1 looked at it) are requested a tile ahead,
    // like its rows — asked for at the top of the tile itself they cost the wave a memory round trip per visited tile,
    // most of what skipping its blocks saved (measured: a visited tile took 0.68 of an unvisited one for 0.135 of its MFMAs)
    uint4 mv[(kCollect && kTileMax) ? NB / 8 : 1];
    auto load_maxima = [&](int64_t t) {
        if constexpr (kCollect && kTileMax) {
            if ((t & (max_step - 1)) == 0 && t < n_tiles) {   // wave-uniform; max_step is a power of two
                const uint4* tm = tile_max + (t >> (31 - __builtin_clz(max_step))) * (NB / 8) * 64 + lane;
#pragma unroll
                for (int j = 0; j < NB / 8; ++j) mv[j] = tm[j * 64];
            }
        }
    };
    load_maxima(first);
    // Which tile comes k rounds after `tile`.  Classic: a fixed stride.  Pass 2 with tile maxima: the wave's position
    // inside a round moves on by one per round, so that its tiles cycle through the residues mod max_step (with a
    // fixed stride — a multiple of 4 — a quarter of the waves would own ALL visited tiles and finish early).
    constexpr bool kRotate = kCollect && kTileMax;
    int64_t rot_base = 0, rot_pos = first;   // (tile_step == 1 in pass 2)
    auto next_tile = [&](int64_t t) -> int64_t {
        if constexpr (kRotate) {
            const int64_t p = rot_pos + 1 == total_waves ? 0 : rot_pos + 1;
            return rot_base + total_waves + p;
        } else {
            return t + total_waves;
        }
    };
    // (the rotated deal ends with the ROUND: a wave whose tile of the last round lies past the end skips it)
    for (int64_t tile = first; kRotate ? rot_base < n_tiles : tile < n_tiles; ) {
        const int64_t after = next_tile(tile);
        if constexpr (kRotate) {
            if (tile >= n_tiles) {   // wave-uniform; the prefetched rows (clamped) are simply dropped
                load_half3(after, pre[0][0], pre[0][1], pre[0][2]);
                rot_base += total_waves;
                rot_pos = rot_pos + 1 == total_waves ? 0 : rot_pos + 1;
                tile = after;
                continue;
            }
        }
        const int64_t row = tile * 64 + lane;
        const bool in_range = row < n;
        uint32_t p0, p1, p2, p3, p4, p5;
        if constexpr (kFromReplica) {
            uint2 x, y, z;
            if constexpr (kStaged) {
                if (slot == 0) {
#pragma unroll
                    for (int d = 0; d < kPre; ++d) {
                        s_rows[wave][d][0][lane] = pre[d][0];
                        s_rows[wave][d][1][lane] = pre[d][1];
                        s_rows[wave][d][2][lane] = pre[d][2];
                    }
#pragma unroll
                    for (int d = 0; d < kPre; ++d)
                        load_half3(tile + (kPre + d) * total_waves, pre[d][0], pre[d][1], pre[d][2]);
                }
                x = s_rows[wave][slot][0][lane];
                y = s_rows[wave][slot][1][lane];
                z = s_rows[wave][slot][2][lane];
                slot = slot + 1 == kPre ? 0 : slot + 1;
            } else {
                x = pre[0][0];
                y = pre[0][1];
                z = pre[0][2];
                load_half3(after, pre[0][0], pre[0][1], pre[0][2]);
            }
            const bool special = in_range && x.x == kBqNaN2;   // tiny / huge / inf / NaN row: exact chain only
            if constexpr (kCollect) {
                if (special) {
                    const int pos = atomicAdd(&counters[0], 1);
                    if (pos < kBqSpecialCap) special_rows[pos] = static_cast<uint32_t>(row);
                }
            }
            const bool keep = in_range && !special;
            p0 = keep ? x.x : 0u; p1 = keep ? x.y : 0u; p2 = keep ? y.x : 0u;
            p3 = keep ? y.y : 0u; p4 = keep ? z.x : 0u; p5 = keep ? z.y : 0u;
        } else {
            float4 a = na, b = nb, c = nc;
            if constexpr ((kVariant & 4) != 0) {   // A/B probe: no HBM traffic, synthetic rows
                const float t = static_cast<float>(tile & 1023) * 1e-3f + lane * 0.01f;
                a = make_float4(t, 0.3f, 0.5f, t * 0.5f);
                b = make_float4(0.1f, t, 0.7f, 0.2f);
                c = make_float4(0.4f, 0.6f, t, 0.9f);
            } else {
                load_row3(tile + total_waves, na, nb, nc);
            }

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
            const bool valid = in_range && tot >= kBqMinNorm2 && tot <= kBqMaxNorm2;
            const float inv = valid ? __builtin_amdgcn_rsqf(tot) : 0.0f;
            if constexpr (kCollect) {
                // neither valid nor exactly zero: scored exactly against every query later
                const bool special = in_range && !valid && !(tot == 0.0f);
                if (special) {
                    const int pos = atomicAdd(&counters[0], 1);
                    if (pos < kBqSpecialCap) special_rows[pos] = static_cast<uint32_t>(row);
                }
            }
            // zero (not NaN) for rows the bound is not claimed for: inf * 0 would poison D
            p0 = bq_pack_h2(a.x * inv, a.y * inv); p1 = bq_pack_h2(a.z * inv, a.w * inv);
            p2 = bq_pack_h2(b.x * inv, b.y * inv); p3 = bq_pack_h2(b.z * inv, b.w * inv);
            p4 = bq_pack_h2(c.x * inv, c.y * inv); p5 = bq_pack_h2(c.z * inv, c.w * inv);
            p0 = valid ? p0 : 0u; p1 = valid ? p1 : 0u; p2 = valid ? p2 : 0u;
            p3 = valid ? p3 : 0u; p4 = valid ? p4 : 0u; p5 = valid ? p5 : 0u;
        }
        // k = 12, 13 multiply the threshold slots of B by 1.0 for ALL rows (a masked row
        // yields D = -T' < 0); k = 14, 15 are zero
        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, p4, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, p5, false, false);
        const auto s2 = __builtin_amdgcn_permlane32_swap(p2, 0x3c003c00u, false, false);
        const auto s3 = __builtin_amdgcn_permlane32_swap(p3, 0u, false, false);
        uint4 aw0, aw1;
        aw0.x = s0[0]; aw0.y = s1[0]; aw0.z = s2[0]; aw0.w = s3[0];   // rows tile*64 +  0..31
        aw1.x = s0[1]; aw1.y = s1[1]; aw1.z = s2[1]; aw1.w = s3[1];   // rows tile*64 + 32..63
        const bq_h8 A[2] = {__builtin_bit_cast(bq_h8, aw0), __builtin_bit_cast(bq_h8, aw1)};

        const bq_f16v zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                              0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        // Two accumulator tiles alternate: MFMA number m + 1 (sub-tile (m+1) & 1 of block
        // (m+1) >> 1) is issued before the 16 results of MFMA m are reduced, so the matrix
        // pipe does not wait for the VALU tree.  The reductions run on the BIT PATTERNS as
        // signed integers (v_max3_i32): for the values that matter (>= 0) integer order is
        // float order, negative floats are negative integers, and there is no NaN
        // canonicalisation to pay for.
        // the fragments are loop-invariant: without this the compiler hoists all NB
        // ds_reads out of the tile loop, back into 4 * NB registers (and spills them)
        asm volatile("" ::: "memory");
        auto max3 = [](int x, int y, int z) { return max(max(x, y), z); };
        // depth-3 tree over the 16 results of one MFMA: 7 operations -> (u0, u1)
        auto tree = [&](const bq_f16v& d, int& u0, int& u1) {
            auto bits = [&](int i) { return static_cast<int>(__float_as_uint(d[i])); };
            const int t0 = max3(bits(0), bits(1), bits(2));
            const int t1 = max3(bits(3), bits(4), bits(5));
            const int t2 = max3(bits(6), bits(7), bits(8));
            const int t3 = max3(bits(9), bits(10), bits(11));
            const int t4 = max3(bits(12), bits(13), bits(14));
            u0 = max3(t0, t1, t2);
            u1 = max3(t3, t4, bits(15));
        };
        if constexpr (!kCollect) {
            // pass 1: fully unrolled (the running maxima are a register array indexed by block)
            bq_f16v D[2];
            uint4 bw[2];
            bw[0] = s_b[0][lane];
            if (NB > 1) bw[1] = s_b[1][lane];
            D[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], __builtin_bit_cast(bq_h8, bw[0]), zero, 0, 0, 0);
            // tile maxima: [visited tile][NB / 8][lane] uint4, dword d of a lane = blocks 2d (low half) and 2d + 1
            // (tile_step is a power of two and divides `tile`)
            uint4* const tm_out = kTileMax ? tile_max + (tile >> (31 - __builtin_clz(tile_step))) * (NB / 8) * 64 + lane : nullptr;
            int tu0 = 0, tu1 = 0, t_even = 0;   // first sub-tile's partial maxima; the even block's tile maximum
            uint32_t tw[4];                     // four finished dwords = eight blocks = one 16-byte store per lane
#pragma unroll
            for (int m = 0; m < 2 * NB; ++m) {
                const int blk = m >> 1, sub = m & 1;
                if (m + 1 < 2 * NB)
                    D[(m + 1) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        A[(m + 1) & 1], __builtin_bit_cast(bq_h8, bw[((m + 1) >> 1) & 1]), zero, 0, 0, 0);
                // block blk's fragment register is free once its second MFMA has been issued
                if (sub == 1 && blk + 2 < NB) bw[blk & 1] = s_b[blk + 2][lane];
                int u0, u1;
                tree(D[m & 1], u0, u1);
                if constexpr (!kTileMax) {
                    mx[blk] = max3(mx[blk], u0, u1);   // the 8th operation folds the running maximum in
                } else if (sub == 0) {
                    tu0 = u0;
                    tu1 = u1;
                } else {
                    // this lane's maximum over its 32 rows of the tile for query (blk, r), clamped at 0 (bit patterns
                    // as signed integers: negative floats are negative)
                    const int t = max3(max3(tu0, tu1, u0), u1, 0);
                    mx[blk] = max(mx[blk], t);
                    if ((blk & 1) == 0) {
                        t_even = t;
                    } else {
                        // two blocks per dword, rounded toward zero like the thresholds pass 2 compares them with
                        tw[(blk >> 1) & 3] = __builtin_bit_cast(
                            uint32_t, __builtin_amdgcn_cvt_pkrtz(__int_as_float(t_even), __int_as_float(t)));
                        if ((blk & 7) == 7) tm_out[(blk >> 3) * 64] = make_uint4(tw[0], tw[1], tw[2], tw[3]);
                    }
                }
            }
        } else {
            // pass 2.  Hits go to the wave's LDS staging buffer (ballot + prefix count, no atomics); the
            // global per-query counters are only touched when the buffer is flushed, 64 entries
            // per round trip.  A returning global atomic per hit kept each wave waiting ~1.5 us
            // about 160 times per pass (measured: 555 vs 485 us).
            // The rare path, and not a cheap one: measured (tools/bq_exp.sh), the blocks that hold a hit cost three times
            // what a block without one does — 27 % of a whole pass at 0.13 hits per (tile, block).  So no ballot and
            // branch per result register: each lane packs the SIGN bits of its 16 results into a mask (one v_alignbit
            // each: mask = mask << 1 | sign) and stages ONE record — (its query, the first of its 16 rows, the mask of
            // those that hit) — whatever the number of hits.
            auto push_hits = [&](const bq_f16v& d, int blk, int sub) {
                const uint32_t q = static_cast<uint32_t>(blk * 32 + r);
                uint32_t signs = 0u;
#pragma unroll
                for (int i = 15; i >= 0; --i) signs = __builtin_amdgcn_alignbit(signs, __float_as_uint(d[i]), 31);   // bit i = sign of d[i]
                const uint32_t hits = ~signs & 0xffffu;   // D >= +0: approx >= T'
                // opaque on purpose: otherwise the compiler hoists the row ids out of this rare path into the tile
                // prologue and spills them
                uint32_t row_lo = static_cast<uint32_t>(tile * 64) + static_cast<uint32_t>(sub * 32 + 4 * h);
                asm volatile("" : "+v"(row_lo));
                const uint64_t who = __ballot(hits != 0u);
                const int n_hit = __popcll(who);
                if (staged + n_hit > kStageCap) {   // wave-uniform
                    bq_flush_stage(stage, staged, lane, cand_count, cand_recs, counters[6]);
                    staged = 0;
                }
                if (hits) stage[staged + lanes_below(who)] = make_uint2(q | (hits << 16), row_lo);
                staged += n_hit;
            };
            // ONE hit test per query block (both 32-row sub-tiles): 16 maxima + 1 compare + 1
            // branch per two MFMAs
            auto check2 = [&](const bq_f16v& da, const bq_f16v& db, int blk) {
#if MI355_BQ_REDUCE == 1
                // A/B variant (tools/bq_reduce.sh — never the product): the test only needs SIGNS, so the two
                // sub-tiles' accumulators are packed pairwise to fp16 (v_cvt_pkrtz_f16_f32: the sign survives, a negative
                // underflow is -0) and the sixteen packed registers are AND-ed (three at a time where the compiler finds
                // v_bitop3_b32): a half of the result has its sign bit SET iff every value of that sub-tile is negative.
                // 16 + 8 + 2 vector instructions per two MFMAs against the tree's 14 + 3 + 1.
                uint32_t all_neg = 0xffffffffu;
#pragma unroll
                for (int i = 0; i < 16; ++i) all_neg &= __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(da[i], db[i]));
                const bool hit_a = (all_neg & 0x00008000u) == 0u, hit_b = (all_neg & 0x80000000u) == 0u;
                if (__builtin_expect(__ballot(hit_a | hit_b) != 0ull, 0)) {
                    if (__ballot(hit_a)) push_hits(da, blk, 0);
                    if (__ballot(hit_b)) push_hits(db, blk, 1);
                }
#else
                int u0, u1, v0, v1;
                tree(da, u0, u1);
                tree(db, v0, v1);
                const int ma = max(u0, u1), mb = max(v0, v1);
                if (__builtin_expect(__ballot(max(ma, mb) >= 0) != 0ull, 0)) {   // some D >= +0: approx >= T'
                    // usually ONE of the two tiles holds the hit: skip the other's 16 ballots
                    if (__ballot(ma >= 0)) push_hits(da, blk, 0);
                    if (__ballot(mb >= 0)) push_hits(db, blk, 1);
                }
#endif
            };
            // Four accumulator tiles: while block b's two tiles are reduced and tested, block
            // b + 1's two MFMAs are in flight.  A ROLLED loop over pairs of blocks (unrolled
            // 2 * NB times the rare hit path would be instantiated 64 times and spill the hot
            // loop); straight-line body: the fragment loads past the last block are clamped
            // and the two MFMAs issued for block NB are simply unused.
            const uint4* sb = &s_b[0][0] + lane;
            if constexpr (NB == 1) {
                // a single block of <= 32 queries (a micro-batch): two MFMAs per tile, one hit test
                const uint4 bw0 = sb[0];
                const bq_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                const bq_f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                check2(D0, D1, 0);
            } else if constexpr (kTileMax) {
                // which query blocks of this tile can hold a candidate at all (bit b = block b): every block of a
                // tile pass 1 did not look at; of a visited tile, the blocks in which some lane's stored maximum
                // reaches its query's threshold
                uint32_t need = static_cast<uint32_t>((1ull << NB) - 1ull);
                if ((tile & (max_step - 1)) == 0) {   // wave-uniform; max_step is a power of two
                    // (the thresholds are loop-invariant: without this the compiler keeps all NB / 2 of them in
                    // registers across the tile loop and the kernel loses a wave per SIMD)
                    asm volatile("" ::: "memory");
                    need = 0u;
#pragma unroll
                    for (int d = 0; d < NB / 2; ++d) {
                        const uint4 four = mv[d >> 2];
                        const uint32_t mm = (d & 3) == 0 ? four.x : ((d & 3) == 1 ? four.y : ((d & 3) == 2 ? four.z : four.w));
                        typedef short bq_s2 __attribute__((ext_vector_type(2)));
                        // fp16 bit patterns of values >= 0 order like integers: maximum - threshold >= 0 per half
                        const bq_s2 diff = __builtin_bit_cast(bq_s2, mm) - __builtin_bit_cast(bq_s2, s_tq[d][r]);
                        if (__ballot(diff[0] >= 0)) need |= 1u << (2 * d);
                        if (__ballot(diff[1] >= 0)) need |= 1u << (2 * d + 1);
                    }
                }
                load_maxima(after);   // (this tile's have been consumed)
                // The blocks in `need`, in pairs, exactly as the plain loop below runs through ALL blocks: while one
                // block's two tiles are reduced and tested the next one's MFMAs run and the fragment of the one after
                // that is on its way from LDS.  An odd count is padded with the dummy block, and the two fragments
                // fetched past the end are the dummy's as well.
                auto pop = [&]() -> int {   // wave-uniform: the next block to do
                    if (!need) return NB;
                    const int b = __builtin_ctz(need);
                    need &= need - 1u;
                    return b;
                };
                const int pairs = (__builtin_popcount(need) + 1) >> 1;
                pairs_done += __builtin_popcount(need);
                if (pairs) {   // wave-uniform
                    int b0 = pop(), b1 = pop();
                    uint4 bw0 = sb[b0 * 64], bw1 = sb[b1 * 64];
                    bq_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                    bq_f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                    bq_f16v D2, D3;
#pragma unroll 1
                    for (int i = 0; i < pairs; ++i) {
                        MI355_BQ_PRIO(2);
                        D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], __builtin_bit_cast(bq_h8, bw1), zero, 0, 0, 0);
                        D3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], __builtin_bit_cast(bq_h8, bw1), zero, 0, 0, 0);
                        MI355_BQ_PRIO(0);
                        const int n0 = pop();
                        bw0 = sb[n0 * 64];
                        check2(D0, D1, b0);
                        MI355_BQ_INTERLEAVE();
                        MI355_BQ_PRIO(2);
                        D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                        D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                        MI355_BQ_PRIO(0);
                        const int n1 = pop();
                        bw1 = sb[n1 * 64];
                        check2(D2, D3, b1);
                        MI355_BQ_INTERLEAVE();
                        b0 = n0;
                        b1 = n1;
                    }
                }
            } else {
                static_assert(NB % 2 == 0, "blocks are processed in pairs");
                uint4 bw0 = sb[0], bw1 = sb[64];
                bq_f16v D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                bq_f16v D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                bq_f16v D2, D3;
#pragma unroll 1
                for (int bp = 0; bp < NB; bp += 2) {
                    const int nb0 = bp + 2 < NB ? bp + 2 : NB - 1;
                    const int nb1 = bp + 3 < NB ? bp + 3 : NB - 1;
                    D2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], __builtin_bit_cast(bq_h8, bw1), zero, 0, 0, 0);   // block bp + 1
                    D3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], __builtin_bit_cast(bq_h8, bw1), zero, 0, 0, 0);
                    bw0 = sb[nb0 * 64];
                    check2(D0, D1, bp);
                    D0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);   // block bp + 2
                    D1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1], __builtin_bit_cast(bq_h8, bw0), zero, 0, 0, 0);
                    bw1 = sb[nb1 * 64];
                    check2(D2, D3, bp + 1);
                }
            }
        }
        if constexpr (kRotate) {
            rot_base += total_waves;
            rot_pos = rot_pos + 1 == total_waves ? 0 : rot_pos + 1;
        }
        tile = after;
    }
    if constexpr (kCollect) bq_flush_stage(stage, staged, lane, cand_count, cand_recs, counters[6]);
    if constexpr (kCollect && kTileMax) {
        if (lane == 0) atomicAdd(&counters[3], pairs_done);   // diagnostics: one atomic per wave and pass
    }

    if constexpr (!kCollect) {
        // group = (workgroup, lane half): max over the workgroup's 4 waves through LDS
        // (the B fragments are no longer needed: their NB KiB hold [wave][NB][64] floats)
        __syncthreads();
        float* s_mx = reinterpret_cast<float*>(&s_b[0][0]);
#pragma unroll
        for (int b = 0; b < NB; ++b) s_mx[(wave * NB + b) * 64 + lane] = __uint_as_float(static_cast<uint32_t>(mx[b]));
        __syncthreads();
        // gmax layout [block][eighth of 4 queries][group][4]: the threshold select of one
        // (block, eighth) then reads ONE contiguous run of groups x 16 bytes.
        // group = half * gridDim.x + workgroup
        for (int i = threadIdx.x; i < NB * 64; i += kBqPassBlock) {
            float m = s_mx[i];
#pragma unroll
            for (int w = 1; w < kBqPassBlock / 64; ++w) m = __builtin_fmaxf(m, s_mx[w * NB * 64 + i]);
            const int b = i >> 6, l = i & 63;
            const int half = l >> 5, c = l & 31;
            const int64_t group = static_cast<int64_t>(half) * gridDim.x + blockIdx.x;
            const int64_t groups = 2ll * gridDim.x;
            gmax[((static_cast<int64_t>(b) * 8 + (c >> 2)) * groups + group) * 4 + (c & 3)] = m;
        }
    }
}

// ---- per-query threshold ----------------------------------------------------------
// One workgroup of 256 threads per (query block, eighth): the group maxima of its 4
// queries ([2 halves][grid] values each) are staged in LDS, then wave w selects the
// (topk+1)-th largest value for query w of the eighth (256 workgroups per 1024 queries:
// one round on a 256-CU part).
constexpr int kBqSelectBlock = 256;
constexpr int kBqSelectQueries = 4;           // queries per workgroup
constexpr int kBqMaxPassGrid = 1280;          // 5 workgroups per CU on 256 CUs
constexpr int kBqSelectKeys = kBqMaxPassGrid * kBqGroupsPerBlock / 64;   // group maxima per lane

__global__ __launch_bounds__(kBqSelectBlock) void bq_select_kernel(
    const float* __restrict__ gmax, int grid_pass1, int n_blocks, int topk, float margin,
    uint32_t* __restrict__ bfrag, uint32_t* __restrict__ qflags, float* __restrict__ qthr,
    const uint32_t* __restrict__ nb_vals /* per query: the neighbourhood's exact bound (bq_prepare_kernel), 0 = none; null: not taken */,
    int n_queries, int* __restrict__ counters /* [7] += queries whose neighbourhood bound beat the group maxima's; [4] += chunks with one */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bq_smem[];
    float* s_vals = reinterpret_cast<float*>(bq_smem);                        // [2 * grid][5]
    int* s_hist = reinterpret_cast<int*>(s_vals + static_cast<size_t>(grid_pass1) * 2 * 5);  // [4][256]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int blk = blockIdx.x >> 3;
    const int part = blockIdx.x & 7;
    const int groups = grid_pass1 * kBqGroupsPerBlock;
    // group gi = half * grid + workgroup; this (block, quarter)'s values are one contiguous
    // run of groups x 8 floats (pass 1 wrote them that way)
    const float* mine_vals = gmax + (static_cast<int64_t>(blk) * 8 + part) * groups * kBqSelectQueries;
    for (int i = tid; i < groups * kBqSelectQueries; i += kBqSelectBlock) s_vals[(i >> 2) * 5 + (i & 3)] = mine_vals[i];
    __syncthreads();
    for (int j = wave; j < kBqSelectQueries; j += kBqSelectBlock / 64) {
        const int c = part * kBqSelectQueries + j;
        const int q = blk * 32 + c;
        uint32_t flag = qflags[q];
        float thr_out = 0.0f;
        if (flag == kBqFlagOk) {   // wave-uniform
            // unique keys: value image in the high word, group id in the low word
            uint64_t mine[kBqSelectKeys];
            int positive = 0;
#pragma unroll
            for (int u = 0; u < kBqSelectKeys; ++u) {
                const int gi = lane + u * 64;   // consecutive lanes: stride 5 words, conflict-free
                uint64_t key = 0ull;
                if (gi < groups) {
                    const float v = s_vals[gi * 5 + j];
                    if (v > 0.0f) {
                        key = (static_cast<uint64_t>(score_to_ordered(v)) << 32) | static_cast<uint32_t>(gi + 1);
                        ++positive;
                    }
                }
                mine[u] = key;
            }
            int total = positive;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
            const int need = topk + 1;   // + 1: the query's own row may be among them and is excluded
            float t_prime = 0.0f;
            if (total >= need) {
                // (any T with at least `need` group maxima at or above it is a valid bound: stopping the radix
                // select a few keys early — up to need / 16 + 1 groups more above T — saves most of its passes and
                // moves T by a hair)
                const uint64_t kth = wave_select_threshold<kBqSelectKeys>(mine, need, false, need / 16 + 1, s_hist + wave * 256);
                const float t = ordered_to_score(static_cast<uint32_t>(kth >> 32));
                t_prime = t - 2.0f * margin - kBqSlack;
            }
            if (nb_vals && q < n_queries) {   // wave-uniform: topk rows score at least this EXACTLY — one margin
                const uint32_t nbv = nb_vals[q];
                const float nb_t = nbv ? ordered_to_score(nbv) - margin - kBqSlack : 0.0f;
                if (nb_t > t_prime && lane == 0) atomicAdd(&counters[7], 1);   // (what engine_batch.hip.h's policy reads: is this catalogue sorted?)
                t_prime = nb_t > t_prime ? nb_t : t_prime;
            }
            if (t_prime > 0.0f) {
                thr_out = t_prime;
            } else {
                flag = kBqFlagQueue;   // the bound cannot be claimed: exact scan for this query
            }
        }
        if (lane == 0) {
            // B[k = 12][c] + B[k = 13][c] = -T' (fp16 hi + lo); a query that is not served
            // here gets -65504 so that nothing ever passes
            float hi = -65504.0f, lo = 0.0f;
            if (flag == kBqFlagOk) {
                const _Float16 hh = static_cast<_Float16>(-thr_out);
                hi = static_cast<float>(hh);
                lo = -thr_out - hi;
            }
            bfrag[(blk * 64 + 32 + c) * 4 + 2] = bq_pack_h2(hi, lo);
            qflags[q] = flag;
            qthr[q] = thr_out;
        }
    }
}

// ---- exact scores and top-N of the candidates ---------------------------------------
// One workgroup per query.  Candidates = the rows named by the query's records (pass 2) + the chunk's special rows
// (disjoint sets).  Queries that cannot be served here are appended to the queue of the exact multi-query scan.
__global__ __launch_bounds__(kBqFinalBlock) void bq_finalize_kernel(
    const float* __restrict__ feats, int64_t row_base, const float* __restrict__ queries,
    const long long* __restrict__ exclude /* may be null */, int n_queries, int topk,
    const uint32_t* __restrict__ qflags, const int* __restrict__ cand_count,
    const uint64_t* __restrict__ cand_recs, int cand_cap, int* __restrict__ counters,
    const uint32_t* __restrict__ special_rows, int* __restrict__ queue /* [n_queries] */,
    uint64_t* __restrict__ out_keys, int64_t* __restrict__ out_idx, float* __restrict__ out_score,
    const uint32_t* __restrict__ nb_vals /* the neighbourhood bounds (bq_prepare_kernel), or null */,
    int* __restrict__ cand_examined /* [n_queries]: rows this query's records named (diagnostics) */,
    int* __restrict__ nb_report /* mapped host memory (or null): [0] = counters[7], [1] = counters[4] + 1 once this chunk's select has run */) {
    // Until round 4's end a query kept at most 2048 candidates, all of them in this buffer at once — and a catalogue
    // whose rows CLUSTER (3000 clusters of 3300 rows, spread 0.03: profiles/r04_clustered.jsonl) sent 986 of 1024
    // queries to the exact queue, 43 ms per batch instead of 0.55.  Now the global list holds up to 65536 records per
    // query and is worked in batches: a record per thread is expanded to its rows in LDS, the rows are scored
    // kBqFinalChunk at a time — kBqFinalAhead of a thread's rows in flight: with one, 33 000 candidates of a query in a
    // large cluster were 130 dependent round trips — then the key buffer is cut to a little over topk, with the cut's
    // threshold as a floor for what is appended later.
    __shared__ uint64_t s_keys[kBqFinalKeys];
    __shared__ uint64_t s_top[kMultiMaxTopK];
    __shared__ uint32_t s_rows[kBqFinalRows];
    __shared__ SelectSmem s_sel;
    __shared__ int s_n;
    __shared__ int s_wave_rows[kBqFinalBlock / 64];
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (nb_report && q == 0 && tid == 0) {   // (a chunk that computed neighbourhood bounds: tell the host what they were worth)
        const int chunks = atomicAdd(&counters[4], 1) + 1;
        __hip_atomic_store(&nb_report[0], counters[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&nb_report[1], chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int n_cand = cand_count[q * kBqCountStride];
    const int n_special = counters[0];
    const bool served = qflags[q] == kBqFlagOk && n_cand <= cand_cap && n_special <= kBqSpecialCap;
    if (!served) {   // uniform
        if (tid == 0) {
            queue[atomicAdd(&counters[1], 1)] = q;
            atomicAdd(&counters[5], 1);   // cumulative since the scratch was allocated (mi355rec_stats_t::route_exact_queue)
            cand_examined[q] = 0;
        }
        return;
    }
    float qv[kDim];
#pragma unroll
    for (int j = 0; j < kDim; ++j) qv[j] = queries[static_cast<int64_t>(q) * kDim + j];
    const float qn = query_norm(qv);
    const long long excl = exclude ? exclude[q] : -1ll;
    if (tid == 0) s_n = 0;
    __syncthreads();
    uint64_t floor_key = 0;   // uniform: keys at or below it cannot be among the best topk any more
    if (nb_vals) {            // (at least topk rows score >= the neighbourhood's exact bound: a key AT it stays)
        const uint32_t nbv = nb_vals[q];
        if (nbv) floor_key = (static_cast<uint64_t>(nbv) << 32) - 1ull;
    }
    int c = 0;
    // cut the buffer to a little over topk in O(c) (not to EXACTLY topk: that takes the radix select through all its
    // byte passes — most of this kernel's time — where a cut that may leave up to kRankDirectMax - topk keys more
    // stops after two or three; the ranking below keeps the best topk of whatever is left)
    auto cut = [&]() {
        uint64_t mine[kBqFinalPerThread];
#pragma unroll
        for (int u = 0; u < kBqFinalPerThread; ++u) {
            const int i = tid + u * kBqFinalBlock;
            mine[u] = i < c ? s_keys[i] : 0ull;
        }
        const int slack = kRankDirectMax - topk > 0 ? kRankDirectMax - topk : 0;
        const uint64_t t = block_select_threshold<kBqFinalBlock, kBqFinalPerThread>(mine, topk, slack == 0, slack, s_sel);
        if (tid == 0) s_n = 0;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kBqFinalPerThread; ++u)
            if (mine[u] >= t) s_keys[atomicAdd(&s_n, 1)] = mine[u];
        __syncthreads();
        c = s_n;
        if (t > floor_key + 1) floor_key = t - 1;   // (at least topk keys >= t are kept)
    };
    // the rows [0, m) of `rows` (LDS or global), kBqFinalChunk between two cuts
    auto score_rows = [&](const uint32_t* rows, int m) {
        for (int base = 0; base < m; base += kBqFinalChunk) {   // uniform
            const int end = base + kBqFinalChunk < m ? base + kBqFinalChunk : m;
            for (int i0 = base + tid; i0 < end; i0 += kBqFinalBlock * kBqFinalAhead) {
                uint32_t row[kBqFinalAhead];
                Row rr[kBqFinalAhead];
#pragma unroll
                for (int u = 0; u < kBqFinalAhead; ++u) {
                    const int i = i0 + u * kBqFinalBlock;
                    row[u] = rows[i < end ? i : base];   // (a valid row: the load below is unconditional)
                    rr[u] = load_row(feats, static_cast<int64_t>(row[u]));
                }
#pragma unroll
                for (int u = 0; u < kBqFinalAhead; ++u) {
                    const float s = cosine_score(qv, qn, rr[u]);
                    const int64_t g = row_base + row[u];
                    const uint64_t key = pack_key(s, static_cast<uint32_t>(g));
                    if (i0 + u * kBqFinalBlock < end && g != excl && key > floor_key) s_keys[atomicAdd(&s_n, 1)] = key;
                }
            }
            __syncthreads();
            c = s_n;
            __syncthreads();   // (everybody has read the count before a cut resets it)
            // a cut leaves at most max(topk, kRankDirectMax) keys (ties cannot inflate it: keys are unique), so the next
            // chunk always fits; what the last one leaves is what the ranking below can take
            if (c > topk && c > kRankDirectMax) cut();
        }
    };
    int examined = 0;
    // Records are taken kBqFinalRecs per thread at a time (768 per workgroup), the next lot requested before this one is scored.
    // Where the rows they name fit the expansion buffer together — shuffled rows: a record names ONE row, a query's ~650
    // records are one lot — they are expanded and scored in one go: one round trip for the records, one for the rows, three
    // rows of a thread in flight (what round 4's plain row-id lists cost; a record per thread per round trip made the
    // finalize 5 us slower on shuffled rows).  Where they do not (a sorted catalogue: most of sixteen
    // rows per record) the thread's records are expanded one after the other, 4096 rows at most each time.
    constexpr int kRecs = kBqFinalRecs;
    const uint64_t* const my_recs = cand_recs + static_cast<int64_t>(q) * cand_cap;
    uint64_t rec_next[kRecs];
#pragma unroll
    for (int u = 0; u < kRecs; ++u) {
        const int i = tid + u * kBqFinalBlock;
        rec_next[u] = i < n_cand ? my_recs[i] : 0ull;
    }
    for (int rbase = 0; rbase < n_cand; rbase += kRecs * kBqFinalBlock) {   // uniform
        uint64_t rec[kRecs];
        int pc[kRecs], all = 0;
#pragma unroll
        for (int u = 0; u < kRecs; ++u) {
            rec[u] = rec_next[u];
            const int i = rbase + (kRecs + u) * kBqFinalBlock + tid;
            rec_next[u] = i < n_cand ? my_recs[i] : 0ull;
            pc[u] = __popc(static_cast<uint32_t>(rec[u] >> 32) & 0xffffu);
            all += pc[u];
        }
        // where does this thread's first row go, and how many rows are there in all (a wave scan + the waves' totals)
        auto place = [&](int mine, int& at, int& m) {
            const int incl = wave_inclusive_scan(mine);
            if (lane == 63) s_wave_rows[tid >> 6] = incl;
            __syncthreads();
            at = incl - mine;
            m = 0;
#pragma unroll
            for (int w = 0; w < kBqFinalBlock / 64; ++w) {
                const int t = s_wave_rows[w];
                at += w < (tid >> 6) ? t : 0;
                m += t;
            }
        };
        auto expand = [&](uint64_t r, int& at) {
            uint32_t mask = static_cast<uint32_t>(r >> 32) & 0xffffu;
            const uint32_t row_lo = static_cast<uint32_t>(r);
            while (mask) {
                const int i = __builtin_ctz(mask);
                mask &= mask - 1u;
                s_rows[at++] = row_lo + static_cast<uint32_t>((i & 3) + 8 * (i >> 2));
            }
        };
        int at, m;
        place(all, at, m);
        if (m <= kBqFinalRows) {   // uniform: the whole lot at once
#pragma unroll
            for (int u = 0; u < kRecs; ++u) expand(rec[u], at);
            __syncthreads();
            examined += m;
            score_rows(s_rows, m);   // (ends with a barrier: s_rows and s_wave_rows are free again)
        } else {
            __syncthreads();         // (everybody has read the waves' totals)
#pragma unroll
            for (int u = 0; u < kRecs; ++u) {
                place(pc[u], at, m);   // (<= 16 rows per thread: fits)
                expand(rec[u], at);
                __syncthreads();
                examined += m;
                score_rows(s_rows, m);
            }
        }
    }
    score_rows(special_rows, n_special);
    if (tid == 0) cand_examined[q] = examined;
    block_rank_and_store<kBqFinalBlock>(s_keys, c, s_top, topk);
    __syncthreads();
    for (int i = tid; i < topk; i += kBqFinalBlock) {
        const uint64_t k = s_top[i];
        out_keys[static_cast<int64_t>(q) * topk + i] = k;
        if (out_idx) out_idx[static_cast<int64_t>(q) * topk + i] = k ? static_cast<int64_t>(static_cast<uint32_t>(~static_cast<uint32_t>(k))) : -1;
        if (out_score) out_score[static_cast<int64_t>(q) * topk + i] = k ? ordered_to_score(static_cast<uint32_t>(k >> 32)) : 0.0f;
    }
}

}  // namespace mi355
