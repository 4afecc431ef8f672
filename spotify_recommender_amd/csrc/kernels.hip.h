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

#pragma clang fp contract(off)

namespace mi355 {

constexpr int kDim = 12;              // Song.h:12
constexpr int kBlock = 512;           // threads per workgroup (8 waves)
constexpr int kRowsPerThread = 4;     // rows in flight per lane per tile
constexpr int kTileRows = kBlock * kRowsPerThread;
constexpr int kMaxTopK = 1024;        // MI355REC_MAX_TOPN_FAST
constexpr int kCandCap = 4096;        // LDS candidate slots per workgroup
constexpr int kCandLimit = kCandCap - kTileRows;  // compact above this
constexpr int kMergeBlock = 1024;
constexpr int kMergeMaxLists = 2048;
constexpr int kMergeSurvCap = 4096;
constexpr int kMergeChunk = 16;      // keys probed per list per merge round

static_assert(kCandLimit >= kMaxTopK, "threshold needs topk survivors");

struct QueryArg {
    float q[kDim];
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

__device__ __forceinline__ Row load_row(const float* __restrict__ feats, int64_t row) {
    const float4* p = reinterpret_cast<const float4*>(feats + row * kDim);
    Row r;
    r.a = p[0];
    r.b = p[1];
    r.c = p[2];
    return r;
}

// ---- workgroup-level candidate compaction --------------------------------------
// Keeps the best min(count, topk) keys of s_cand[0..count) in s_cand[0..) sorted
// descending (rank by counting: keys are unique), and returns the new filter
// threshold (the topk-th key, or 0 while fewer than topk candidates exist).
// Must be called by every thread of the workgroup.

template <int kThreads>
__device__ inline uint64_t compact_candidates(uint64_t* s_cand, uint64_t* s_top,
                                              int* s_count, int topk) {
    __syncthreads();
    const int c = *s_count;
    for (int i = threadIdx.x; i < c; i += kThreads) {
        const uint64_t mine = s_cand[i];
        int rank = 0;
        int j = 0;
        for (; j + 4 <= c; j += 4) {
            rank += (s_cand[j] > mine) + (s_cand[j + 1] > mine) +
                    (s_cand[j + 2] > mine) + (s_cand[j + 3] > mine);
        }
        for (; j < c; ++j) rank += (s_cand[j] > mine);
        if (rank < topk) s_top[rank] = mine;
    }
    __syncthreads();
    const int kept = c < topk ? c : topk;
    for (int i = threadIdx.x; i < kept; i += kThreads) s_cand[i] = s_top[i];
    const uint64_t thr = (c >= topk) ? s_top[topk - 1] : 0ull;
    __syncthreads();
    if (threadIdx.x == 0) *s_count = kept;
    __syncthreads();
    return thr;
}

// ---- streaming scan ----------------------------------------------------------
// Workgroup b owns the contiguous rows [b*rows_per_block, (b+1)*rows_per_block)
// (rows_per_block is a multiple of 64, so every wave-level load instruction
// covers one contiguous 3 KiB span).  One lane = one row: the 12-term sums are
// sequential in-lane, which is what makes the result bit-identical to the
// reference CPU loop.  The next tile is loaded before the current one is
// scored so ~200 KiB per CU stay in flight.
//
// kScoresOnly: write the n scores (mirrors calculateSimilarities' output).
// else: filter keys against the workgroup's running topk-th key, append the
// survivors to an LDS buffer, and leave the workgroup's sorted top-k list in
// block_lists[b][0..topk).

// kDebug (development A/B only; 0 in the product): 1 = no end-of-tile barrier pair,
// 2 = ballot only, no LDS append, 4 = skip the seed compaction (threshold preset).
template <bool kQueryFromRow, bool kScoresOnly, int kDebug = 0>
__global__ __launch_bounds__(kBlock) void scan_kernel(
    const float* __restrict__ feats, int64_t n, int64_t rows_per_block, int iters,
    int64_t row_base, QueryArg qarg, int64_t query_row, int64_t exclude_global,
    int topk, uint64_t* __restrict__ block_lists, float* __restrict__ scores_out) {
    __shared__ uint64_t s_cand[kScoresOnly ? 1 : kCandCap];
    __shared__ uint64_t s_top[kScoresOnly ? 1 : kMaxTopK];
    __shared__ int s_count;

    float q[kDim];
    if constexpr (kQueryFromRow) {
        const float* qp = feats + query_row * kDim;  // wave-uniform: scalar loads
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = qp[j];
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = qarg.q[j];
    }
    const float qn = query_norm(q);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t blk_begin = static_cast<int64_t>(blockIdx.x) * rows_per_block;
    int64_t blk_end = blk_begin + rows_per_block;
    if (blk_end > n) blk_end = n;
    // rows past the block's end re-read its last row (one cached line) so the
    // prefetch can be unconditional: a conditional load would make the compiler
    // wait vmcnt(0) at the join and serialise the pipeline
    const int64_t last_row = blk_end - 1;  // the host launches only non-empty blocks

    if constexpr (!kScoresOnly) {
        if (tid == 0) s_count = 0;
        __syncthreads();
    }
    uint64_t thr = 0;
    if constexpr ((kDebug & 4) != 0) thr = pack_key(0.985f, 0u);
    // Re-tighten the threshold once ~topk NEW candidates have piled up: ranking
    // c candidates costs ~c*c/32 LDS cycles, so c must stay a small multiple of
    // topk (a stale threshold that lets 2000+ candidates through costs ~85 us).
    int compact_at = 2 * topk > 256 ? 2 * topk : 256;
    if (compact_at > kCandLimit) compact_at = kCandLimit;

    auto load_tile = [&](Row (&dst)[kRowsPerThread], int it) {
        const int64_t tile_begin = blk_begin + static_cast<int64_t>(it) * kTileRows;
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            const int64_t r = tile_begin + u * kBlock + tid;
            dst[u] = load_row(feats, r < blk_end ? r : last_row);
        }
    };

    auto process_tile = [&](const Row (&rows)[kRowsPerThread], int it) {
        const int64_t tile_begin = blk_begin + static_cast<int64_t>(it) * kTileRows;
#pragma unroll
        for (int u = 0; u < kRowsPerThread; ++u) {
            const int64_t r = tile_begin + u * kBlock + tid;
            const bool in_range = r < blk_end;
            const float s = cosine_score(q, qn, rows[u]);
            if constexpr (kScoresOnly) {
                if (in_range) scores_out[r] = s;
            } else {
                const int64_t g = row_base + r;
                uint64_t key = pack_key(s, static_cast<uint32_t>(g));
                if (!in_range || g == exclude_global) key = 0;
                const bool pass = key > thr;
                const uint64_t ballot = __ballot(pass);
                if (ballot && (kDebug & 2) == 0) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_count, __popcll(ballot));
                    base = __builtin_amdgcn_readfirstlane(base);
                    const int pos = base + __popcll(ballot & ((1ull << lane) - 1ull));
                    if (pass) s_cand[pos] = key;
                }
                if (thr == 0 && (kDebug & 4) == 0) {
                    // seed: establish the threshold from the first rows instead
                    // of letting whole unfiltered tiles pile up (thr is uniform)
                    __syncthreads();
                    const int c = s_count;
                    __syncthreads();
                    if (c >= topk) thr = compact_candidates<kBlock>(s_cand, s_top, &s_count, topk);
                }
            }
        }
        if constexpr (!kScoresOnly && (kDebug & 1) == 0) {
            // keep room for one more unfiltered tile (two barriers: every wave
            // must have read the count before any wave appends again)
            __syncthreads();
            const int c = s_count;
            __syncthreads();
            if (c >= compact_at) thr = compact_candidates<kBlock>(s_cand, s_top, &s_count, topk);
        }
    };

    Row buf_a[kRowsPerThread];
    Row buf_b[kRowsPerThread];
    load_tile(buf_a, 0);
    for (int it = 0; it < iters; it += 2) {
        load_tile(buf_b, it + 1);
        process_tile(buf_a, it);
        load_tile(buf_a, it + 2);
        process_tile(buf_b, it + 1);  // fully masked when it + 1 == iters
    }

    if constexpr (!kScoresOnly) {
        compact_candidates<kBlock>(s_cand, s_top, &s_count, topk);
        const int kept = s_count;
        uint64_t* out = block_lists + static_cast<int64_t>(blockIdx.x) * topk;
        for (int i = tid; i < topk; i += kBlock) out[i] = (i < kept) ? s_cand[i] : 0ull;
    }
}

// ---- merge of sorted candidate lists -----------------------------------------
// n_lists lists of list_len keys (each sorted descending, 0-padded) -> the best
// topk keys, sorted descending, 0-padded; optional unpack to (row, score).
// One workgroup per query (blockIdx.x = query in a batch; list / output bases
// advance by the per-query strides).
//
// Fast path: a key can only be in the global top-k if it is >= T, where T is
// the topk-th largest list HEAD (topk different lists each hold a key >= T).
// Lists are sorted, so the survivors of each list are a prefix; for
// statistically similar shards about 1.3*topk keys survive in total.  They are
// ranked by counting in LDS.  If more than kMergeSurvCap survive (adversarial
// input), an exact MSB-first radix select over all keys finds the topk-th key.

__device__ inline uint64_t merge_radix_select(const uint64_t* __restrict__ lists,
                                              int64_t total, int topk, int* s_hist,
                                              uint64_t* s_sel) {
    // returns the topk-th largest key (0 if fewer than topk non-zero keys)
    uint64_t prefix = 0, mask = 0;
    int remaining = topk;
    for (int pass = 7; pass >= 0; --pass) {
        for (int i = threadIdx.x; i < 256; i += blockDim.x) s_hist[i] = 0;
        __syncthreads();
        const int shift = pass * 8;
        for (int64_t i = threadIdx.x; i < total; i += blockDim.x) {
            const uint64_t k = lists[i];
            if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0, d = 255;
            for (; d > 0; --d) {
                if (acc + s_hist[d] >= remaining) break;
                acc += s_hist[d];
            }
            s_sel[0] = static_cast<uint64_t>(d);
            s_sel[1] = static_cast<uint64_t>(remaining - acc);
        }
        __syncthreads();
        prefix |= s_sel[0] << shift;
        mask |= 255ull << shift;
        remaining = static_cast<int>(s_sel[1]);
        __syncthreads();
    }
    return prefix;
}

__global__ __launch_bounds__(kMergeBlock) void merge_kernel(
    const uint64_t* __restrict__ lists_base, int n_lists, int list_len,
    int64_t lists_query_stride, int topk, uint64_t* __restrict__ out_keys_base,
    int64_t* __restrict__ out_idx_base, float* __restrict__ out_score_base,
    int64_t out_query_stride) {
    __shared__ uint64_t s_surv[kMergeSurvCap];
    __shared__ uint64_t s_heads[kMergeMaxLists];
    __shared__ uint64_t s_top[kMaxTopK];
    __shared__ uint64_t s_sel[2];
    __shared__ int s_hist[256];
    __shared__ int s_count;
    __shared__ int s_overflow;
    __shared__ int s_more;
    __shared__ unsigned short s_active[kMergeMaxLists];

    const int tid = threadIdx.x;
    const uint64_t* lists = lists_base + static_cast<int64_t>(blockIdx.x) * lists_query_stride;
    uint64_t* out_keys = out_keys_base + static_cast<int64_t>(blockIdx.x) * out_query_stride;

    if (tid == 0) {
        s_count = 0;
        s_overflow = 0;
        s_sel[0] = 0;
    }
    // Threshold: if m lists each hold >= j keys that are >= T, then m*j keys
    // are >= T, so T is a valid lower bound of the topk-th key when m*j >= topk.
    // Take j = 1 when there are plenty of lists (T = topk-th largest head),
    // deeper probes when there are few (e.g. 8 per-rank lists).
    int probe = 1;
    if (n_lists < 2 * topk) probe = (2 * topk + n_lists - 1) / n_lists;
    if (probe > list_len) probe = list_len;
    const int need_lists = (topk + probe - 1) / probe;
    for (int l = tid; l < n_lists; l += kMergeBlock)
        s_heads[l] = lists[static_cast<int64_t>(l) * list_len + (probe - 1)];
    __syncthreads();

    if (need_lists <= n_lists) {
        for (int l = tid; l < n_lists; l += kMergeBlock) {
            const uint64_t mine = s_heads[l];
            int rank = 0;
            for (int m = 0; m < n_lists; ++m) {
                const uint64_t o = s_heads[m];
                rank += (o > mine) || (o == mine && m < l);
            }
            if (rank == need_lists - 1) s_sel[0] = mine;
        }
    }
    __syncthreads();
    uint64_t thr = s_sel[0];
    if (thr == 0) thr = 1;  // accept every non-empty key
    __syncthreads();

    // survivors: the prefix of each list with key >= thr.  Round d looks at
    // keys [d*C, (d+1)*C) of every list that is still active (its previous
    // chunk passed entirely); all loads of a round are independent.  Typically
    // one round (a fraction of a key per list survives).
    for (int l = tid; l < n_lists; l += kMergeBlock) s_active[l] = 0;
    __syncthreads();
    for (int round = 0; round * kMergeChunk < list_len; ++round) {
        if (tid == 0) s_more = 0;
        __syncthreads();
        for (int t = tid; t < n_lists * kMergeChunk; t += kMergeBlock) {
            const int l = t / kMergeChunk;
            const int pos = round * kMergeChunk + (t % kMergeChunk);
            if (pos < list_len && s_active[l] == round) {
                const uint64_t k = lists[static_cast<int64_t>(l) * list_len + pos];
                if (k >= thr) {
                    const int slot = atomicAdd(&s_count, 1);
                    if (slot < kMergeSurvCap) s_surv[slot] = k;
                    else s_overflow = 1;
                    if ((t % kMergeChunk) == kMergeChunk - 1) s_more = 1;
                }
            }
        }
        __syncthreads();
        if (!s_more || s_overflow) break;
        // a list stays active iff the LAST key of this chunk passed
        for (int l = tid; l < n_lists; l += kMergeBlock) {
            if (s_active[l] == round) {
                const int last = (round + 1) * kMergeChunk - 1;
                if (last < list_len && lists[static_cast<int64_t>(l) * list_len + last] >= thr)
                    s_active[l] = static_cast<unsigned short>(round + 1);
            }
        }
        __syncthreads();
    }
    __syncthreads();

    if (s_overflow) {
        // exact fallback: radix-select the topk-th key over everything
        const int64_t total = static_cast<int64_t>(n_lists) * list_len;
        uint64_t kth = merge_radix_select(lists, total, topk, s_hist, s_sel);
        if (kth == 0) kth = 1;
        if (tid == 0) s_count = 0;
        __syncthreads();
        for (int64_t i0 = 0; i0 < total; i0 += kMergeBlock) {
            const int64_t i = i0 + tid;
            const uint64_t k = (i < total) ? lists[i] : 0ull;
            if (k >= kth) {
                const int pos = atomicAdd(&s_count, 1);  // exactly topk keys when unique
                if (pos < kMergeSurvCap) s_surv[pos] = k;
            }
        }
        __syncthreads();
    }

    const int c = s_count < kMergeSurvCap ? s_count : kMergeSurvCap;
    for (int i = tid; i < c; i += kMergeBlock) {
        const uint64_t mine = s_surv[i];
        int rank = 0;
        int j = 0;
        for (; j + 4 <= c; j += 4) {
            rank += (s_surv[j] > mine) + (s_surv[j + 1] > mine) +
                    (s_surv[j + 2] > mine) + (s_surv[j + 3] > mine);
        }
        for (; j < c; ++j) rank += (s_surv[j] > mine);
        if (rank < topk) s_top[rank] = mine;
    }
    __syncthreads();
    const int kept = c < topk ? c : topk;
    for (int i = tid; i < topk; i += kMergeBlock) {
        const uint64_t k = (i < kept) ? s_top[i] : 0ull;
        out_keys[i] = k;
        if (out_idx_base) {
            out_idx_base[static_cast<int64_t>(blockIdx.x) * out_query_stride + i] =
                k ? static_cast<int64_t>(static_cast<uint32_t>(~static_cast<uint32_t>(k))) : -1;
        }
        if (out_score_base) {
            out_score_base[static_cast<int64_t>(blockIdx.x) * out_query_stride + i] =
                k ? ordered_to_score(static_cast<uint32_t>(k >> 32)) : 0.0f;
        }
    }
}

// ---- read-only streaming probe (achievable-HBM ceiling) ---------------------

__global__ __launch_bounds__(kBlock) void stream_probe_kernel(
    const float4* __restrict__ data, int64_t n_vec, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        const float4 a = data[i];
        const float4 b = data[i + stride];
        const float4 c = data[i + 2 * stride];
        const float4 d = data[i + 3 * stride];
        acc ^= __float_as_uint(a.x) ^ __float_as_uint(a.y) ^ __float_as_uint(a.z) ^ __float_as_uint(a.w);
        acc ^= __float_as_uint(b.x) ^ __float_as_uint(b.y) ^ __float_as_uint(b.z) ^ __float_as_uint(b.w);
        acc ^= __float_as_uint(c.x) ^ __float_as_uint(c.y) ^ __float_as_uint(c.z) ^ __float_as_uint(c.w);
        acc ^= __float_as_uint(d.x) ^ __float_as_uint(d.y) ^ __float_as_uint(d.z) ^ __float_as_uint(d.w);
    }
    for (; i < n_vec; i += stride) {
        const float4 a = data[i];
        acc ^= __float_as_uint(a.x) ^ __float_as_uint(a.y) ^ __float_as_uint(a.z) ^ __float_as_uint(a.w);
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
