// merge.hip.h — the merge of sorted candidate lists into the global top-N (replaces the reference's host heap,
// Recommender.cu:293-315, for the per-workgroup lists of a scan and for the per-shard lists of a node), the tail that
// lets a lone query's scan launch merge its own lists, and the merge kernels.
#pragma once

#include "core.hip.h"

#pragma clang fp contract(off)

namespace mi355 {

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

// The exact fallback: the topk-th largest key among the `n_live` lists named in `live` (0 if they hold fewer than topk
// keys).  Empty slots are skipped and a wave adds the keys that share its first key's digit with ONE LDS atomic: the
// keys that end up here are close together (that is why the threshold let too many through), so in the upper byte
// passes every lane would otherwise add to the same histogram word — 77 000 serialised atomics per pass made this a
// 330 us affair when it first ran on a catalogue of few large clusters.
template <int kThreads, bool kCoherent = false>
__device__ inline uint64_t merge_global_radix_select(const uint64_t* lists, const unsigned short* live, int n_live,
                                                     int list_len, int64_t list_stride, int topk, int* s_hist, int* s_pair) {
    const int lane = threadIdx.x & 63;
    const int total = n_live * list_len;
    uint64_t prefix = 0, mask = 0;
    int remaining = topk;
    for (int pass = 7; pass >= 0; --pass) {
        for (int i = threadIdx.x; i < 256; i += kThreads) s_hist[i] = 0;
        __syncthreads();
        const int shift = pass * 8;
        for (int i0 = 0; i0 < total; i0 += kThreads) {   // uniform trip count: every lane takes part in the ballots
            const int i = i0 + static_cast<int>(threadIdx.x);
            const uint64_t k = i < total ? ld_key<kCoherent>(&lists[static_cast<int64_t>(live[i / list_len]) * list_stride + (i % list_len)]) : 0ull;
            const bool in = k != 0ull && (k & mask) == prefix;
            const int bin = static_cast<int>((k >> shift) & 255u);
            const uint64_t who = __ballot(in);
            if (who) {   // wave-uniform
                const int lead = __ffsll(static_cast<long long>(who)) - 1;
                const int b0 = __builtin_amdgcn_readlane(bin, lead);
                const uint64_t same = __ballot(in && bin == b0);
                if (lane == lead) atomicAdd(&s_hist[b0], __popcll(same));
                else if (in && bin != b0) atomicAdd(&s_hist[bin], 1);
            }
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
    int deeper;   // lists whose whole first chunk passed the threshold (they are walked)
    int n_active; // non-empty lists compacted into `active` (the one-phase path over few lists)
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
        s_pair[1] = 0;
        s_more = 0;
        sm.deeper = 0;
        sm.n_active = 0;
    }
    __syncthreads();
    int probe = 1;
    if (n_lists < 2 * topk) probe = (2 * topk + n_lists - 1) / n_lists;
    if (probe > list_len) probe = list_len;
    const int need_lists = (topk + probe - 1) / probe;
    int slack = need_lists / 8;
    uint64_t thr = 1;  // accept every non-empty key
    int first = 0;     // keys [0, first) of every list are already dealt with
    bool synced = false;   // uniform: the branch below has already put a barrier behind its survivors

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
        int local_full = 0;   // lists whose whole first chunk is there: with thr = 1 they are the ones deeper rounds would walk
#pragma unroll
        for (int u = 0; u < kFirstPer; ++u) {
            hk[u] = j == 0 ? k[u] : 0ull;
            local_nonzero += hk[u] != 0ull;
            local_full += (j == first - 1 && k[u] != 0ull) ? 1 : 0;
        }
        for (int l = tid; l < n_lists; l += kThreads) s_active[l] = 0xffff;
        {   // one LDS atomic per wave (hundreds of threads adding to the one word serialise: measured 2 us in the 8-bit
            // scan's sample selection, the same pattern)
            const int wave_nonzero = __builtin_amdgcn_readlane(wave_inclusive_scan(local_nonzero), 63);
            if ((tid & 63) == 0 && wave_nonzero) atomicAdd(&s_pair[0], wave_nonzero);
            const int wave_full = __builtin_amdgcn_readlane(wave_inclusive_scan(local_full), 63);
            if ((tid & 63) == 0 && wave_full) atomicAdd(&s_pair[1], wave_full);
        }
        __syncthreads();
        const int nonempty = s_pair[0];
        const int full_chunks = s_pair[1];
        // FEW NON-EMPTY LISTS.  With a launch-wide bound most workgroups of a scan keep nothing, and on a catalogue whose
        // similar rows lie next to each other the whole top-k sits in the lists of the two or twenty workgroups that met
        // the query's cluster: fewer heads than the threshold select needs, thr stays 1, and those lists were then walked
        // 16 keys per dependent round trip (a query alone at 10 M rows, 3000 contiguous clusters: 67 us against 47 on
        // uniform rows, all of it here).  Instead: ONE load phase over all keys of the non-empty lists, when they fit.
        // (only where some list WOULD be walked: many lists of two or three keys each — a uniform catalogue under a good bound
        // — are done after the first phase as they are, and a second load phase would only add a round trip)
        const bool sparse = nonempty < need_lists && full_chunks > 0 && static_cast<int64_t>(nonempty) * list_len <= kSurvCap;   // uniform
        if (nonempty >= need_lists)  // uniform
            thr = block_select_threshold<kThreads, kFirstPer>(hk, need_lists, false, slack, s_sel);
        if (sparse) {   // uniform
#pragma unroll
            for (int u = 0; u < kFirstPer; ++u)
                if (hk[u]) s_active[atomicAdd(&sm.n_active, 1)] = static_cast<unsigned short>((u * kThreads + tid) / kMergeFirst);
            __syncthreads();
            first = list_len;   // every key is taken here: no deeper rounds (s_more stays 0)
            const int total = sm.n_active * list_len;   // <= kSurvCap
            // all loads are requested before the first is looked at: ONE memory round trip, not one per kThreads keys
            uint64_t kk[kSurvPer];
#pragma unroll
            for (int r = 0; r < kSurvPer; ++r) {
                const int i = r * kThreads + tid;
                kk[r] = i < total ? ld_key<kCoherent>(&lists[static_cast<int64_t>(s_active[i / list_len]) * list_stride + (i % list_len)]) : 0ull;
            }
#pragma unroll
            for (int r = 0; r < kSurvPer; ++r) {   // (uniform loop: every lane takes part in the ballot)
                const uint64_t have = __ballot(kk[r] != 0ull);
                int base = 0;
                if ((tid & 63) == 0 && have) base = atomicAdd(&s_count, __popcll(have));
                base = __builtin_amdgcn_readfirstlane(base);
                if (kk[r]) s_surv[base + lanes_below(have)] = kk[r];
            }
        } else {
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
                    atomicAdd(&sm.deeper, 1);
                }
            }
        }
        // THE KEYS SIT IN FEW LISTS: the same argument one level down.  When `deep_need` = ceil(topk / first) lists passed
        // their whole first chunk, those lists hold `first` keys each at or above the smallest of their chunk-end keys — a
        // threshold that is at least the one above (every one of those chunk ends passed it).  A catalogue of few LARGE
        // clusters: 65 workgroups met the query's cluster and kept 30 ... 100 keys each, a hundred others one stray key;
        // the 100th largest head was a stray, all 4 000 keys of the 65 lists "survived", overflowed the riding merger's
        // 2048 slots and sent every third launch of a streamed fp32 scan through the exact fallback (160 us per query
        // where the scan takes 80).  On a shuffled catalogue under a launch-wide bound few lists get this far and the
        // second select is not run.
        __syncthreads();
        synced = true;
        const int deep_need = (topk + first - 1) / first;
        if (first > 1 && sm.deeper >= deep_need) {   // uniform
#pragma unroll
            for (int u = 0; u < kFirstPer; ++u) hk[u] = (j == first - 1 && k[u] >= thr) ? k[u] : 0ull;
            const uint64_t t2 = block_select_threshold<kThreads, kFirstPer>(hk, deep_need, false, deep_need / 8, s_sel);
            thr = t2 > thr ? t2 : thr;
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
    if (!synced) __syncthreads();   // uniform
    MI355REC_MPHASE(1);   // first chunk loaded, threshold selected, survivors appended

    // Deeper rounds: round d looks at the next chunk of every list that is still active (its previous chunk passed
    // entirely); chunks grow (16, 32, 64, 128 keys: a list that is still there after two rounds is a long one), the loads of
    // a round are independent and issued before any of them is consumed, and the thread that holds a chunk's LAST key is the
    // one that keeps its list active — no second, dependent look at the list per round.  (Until round 5: 16 keys per round and
    // a re-load of every active list's last key, two dependent round trips per 16 keys: 65 lists of 30 ... 100 keys — a
    // catalogue of few large clusters — took 12 us to walk.)
    // With the first-chunk phase above this loop usually does not run at all.
    int start = first;
    for (int round = 0; start < list_len && !s_overflow; ++round) {
        // s_more was raised by whoever marked a list active for this round
        if (!s_more) break;  // uniform: read after a barrier, rewritten only after the next one
        __syncthreads();
        if (tid == 0) s_more = 0;
        __syncthreads();
        const int log_chunk = 4 + (round < 3 ? round : 3);
        const int chunk = 1 << log_chunk;   // (kMergeChunk = 16 is the first)
        const int total = n_lists << log_chunk;
        for (int t0 = 0; t0 < total; t0 += kThreads * 8) {   // (a list's chunk never straddles two iterations: chunk divides kThreads * 8)
            uint64_t k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = t0 + u * kThreads + tid;
                const int l = t >> log_chunk;
                const int pos = start + (t & (chunk - 1));
                // (== round + 1: the holder of this chunk's last key, in another wave, has already promoted the list)
                const unsigned a = t < total ? s_active[l] : 0xffffu;
                const bool live = pos < list_len && (a == static_cast<unsigned>(round) || a == static_cast<unsigned>(round) + 1u);
                k[u] = live ? ld_key<kCoherent>(&lists[static_cast<int64_t>(l) * list_stride + pos]) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k[u] >= thr) {
                    const int slot = atomicAdd(&s_count, 1);
                    if (slot < kSurvCap) s_surv[slot] = k[u];
                    else s_overflow = 1;
                    const int t = t0 + u * kThreads + tid;
                    if ((t & (chunk - 1)) == chunk - 1 && start + chunk < list_len) {   // the chunk's last key passed and the list goes on
                        s_active[t >> log_chunk] = static_cast<unsigned short>(round + 1);
                        s_more = 1;
                    }
                }
            }
        }
        __syncthreads();
        start += chunk;
    }
    __syncthreads();
    if (s_overflow) {
        // Too many keys passed the threshold.  The survivor buffer is full of genuine keys, so the topk-th largest of THEM
        // is a valid, much higher threshold: take it, and gather again in one sweep over the non-empty lists.  Only if that
        // overflows too (mass ties: thousands of equal scores) does the exact radix select over those lists run.
        if (tid == 0) s_pair[1] = 0;
        __syncthreads();
        for (int l = tid; l < n_lists; l += kThreads)
            if (ld_key<kCoherent>(&lists[static_cast<int64_t>(l) * list_stride]) != 0ull) s_active[atomicAdd(&s_pair[1], 1)] = static_cast<unsigned short>(l);
        uint64_t retry_thr;
        {
            uint64_t mine[kSurvPer];
#pragma unroll
            for (int r = 0; r < kSurvPer; ++r) mine[r] = s_surv[tid + r * kThreads];
            retry_thr = block_select_threshold<kThreads, kSurvPer>(mine, topk, false, topk / 8, s_sel);   // (barriers inside)
        }
        __syncthreads();
        const int live = s_pair[1];
        const int total = live * list_len;
        if (tid == 0) {
            s_count = 0;
            s_overflow = 0;
        }
        __syncthreads();
        for (int i0 = 0; i0 < total; i0 += kThreads) {   // uniform trip count: every lane takes part in the ballot
            const int i = i0 + tid;
            const uint64_t k = i < total ? ld_key<kCoherent>(&lists[static_cast<int64_t>(s_active[i / list_len]) * list_stride + (i % list_len)]) : 0ull;
            const bool pass = k >= retry_thr;   // retry_thr >= 1
            const uint64_t who = __ballot(pass);
            int base = 0;
            if ((tid & 63) == 0 && who) base = atomicAdd(&s_count, __popcll(who));
            base = __builtin_amdgcn_readfirstlane(base);
            if (pass) {
                const int slot = base + lanes_below(who);
                if (slot < kSurvCap) s_surv[slot] = k;
                else s_overflow = 1;
            }
        }
        __syncthreads();
        if (s_overflow) {   // uniform
            __syncthreads();
            uint64_t kth = merge_global_radix_select<kThreads, kCoherent>(lists, s_active, live, list_len, list_stride, topk, s_sel.hist, s_pair);
            if (kth == 0) kth = 1;
            if (tid == 0) s_count = 0;
            __syncthreads();
            for (int i = tid; i < total; i += kThreads) {
                const uint64_t k = ld_key<kCoherent>(&lists[static_cast<int64_t>(s_active[i / list_len]) * list_stride + (i % list_len)]);
                if (k >= kth) {
                    const int pos = atomicAdd(&s_count, 1);  // exactly topk keys when unique
                    if (pos < kSurvCap) s_surv[pos] = k;
                }
            }
            __syncthreads();
        }
    }

    MI355REC_MPHASE(2);   // deeper rounds done
    int c = s_count < kSurvCap ? s_count : kSurvCap;
    __syncthreads();
    // (from kRankCountMax keys up the ranking below is a bitonic sort — 45 barrier stages for 512 slots, 8 us in this
    // 1024-thread workgroup: a survivor set of 300 keys, which a catalogue of few large clusters leaves, is cut first)
    if (c > topk && c > kRankCountMax) {  // uniform: too many to rank by counting, cut to a little over topk in O(c)
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
}  // namespace mi355
