// handoff.hip.h — the LAUNCH-WIDE BOUND of a query and how it travels between workgroups and launches.
//
// The reference scores every row for every query (Recommender.cu:184-254) and keeps the best topN on the host
// (:293-315).  A row can be among them only if its score reaches the topN-th best score of the shard; every scan of
// this engine therefore starts from a LOWER BOUND of that score, found before the scan by looking at a few rows:
//
//   * a spread SAMPLE: up to 256 evenly spaced regions, one value per 64-lane wave tile of each (the tile's best row),
//     v = the topN-th largest of those <= 2048 values: topN distinct rows score >= v.  Taken over the rows the scan
//     itself streams (fp32 rows: f32_sample_regions below; 8-bit replica: replica_q8.hip.h; fp16 replica, many queries
//     at once on the matrix core: replica_multi.hip.h);
//   * the query row's NEIGHBOURHOOD (nbhd_bound): the topN-th best EXACT score among the 2048 rows around the row the
//     query excludes — for recommendByIndex (Recommender.cu:275-318) that is the query's own row.  A catalogue that
//     is sorted by genre, artist or album keeps similar rows next to each other (DataManager.cpp:244-250,299: genre
//     ids are handed out in order of first appearance and end up in features[11]); there an evenly spaced sample sees
//     eight rows of the query's cluster, never a hundred, and its bound comes from OTHER clusters.  The neighbourhood
//     is disjoint from nothing and needs to be: it is simply a SECOND lower bound, and the scan takes the larger.
//
// Both are bounds on EXACT scores of real rows, so a scan over the fp32 rows uses them as they are and a scan over a
// replica subtracts its error bound once.  They only ever rule rows OUT; whatever a broken or missing bound lets
// through is scored by the exact chain as always.
//
// HAND-OFFS THAT FAIL SAFE.  Values cross workgroups outside the stream order of launches in two places: the sample
// values the "seed riders" of a launch leave for the rider that finishes last (same launch), and the bound that rider
// leaves for the scanners of the next launch.  Both decide which rows a scan may skip, so a reader that picked up a
// value of an EARLIER query (the buffers alternate) could place the bound above the true N-th score: a silently wrong
// top-N.  Every such value therefore carries the EPOCH of the query it belongs to in its upper word, written with the
// value by one 64-bit store; a reader that finds another epoch treats the value as absent — fewer sample values, or
// no launch-wide bound at all, both of which only LOWER the bound (slower, never wrong).  Arrival counters count up
// across launches and are never reset: the host tells each launch the count it starts from, so a missed or repeated
// reset cannot make a rider believe it is the last one.  (tests/test_gpu_replica.py poisons the buffers and drops
// stores through mi355rec_debug_handoff to check this.)
#pragma once

#include "core.hip.h"

#pragma clang fp contract(off)

namespace mi355 {

constexpr int kHalfSeedBlock = 512;            // threads of a sampling workgroup
constexpr int kHalfSeedWaves = kHalfSeedBlock / 64;
constexpr int kHalfSeedMaxGrid = 256;          // sampled regions: <= 2048 sample values per query
constexpr int kHalfSeedPerThread = 4;          // x 512 threads of a workgroup that selects from them
constexpr int kNbhdRows = 2048;                // rows around the excluded row that give the neighbourhood's bound
constexpr int kNbhdSlot = kHalfSeedMaxGrid * kHalfSeedWaves;   // where a single query's sample buffer keeps that bound ...
constexpr int kSampleSlots = kNbhdSlot + 8;                    // ... so such a buffer holds this many tagged values

__device__ __forceinline__ unsigned long long tag_value(uint32_t epoch, uint32_t v) {
    return (static_cast<unsigned long long>(epoch) << 32) | v;
}
// (Epoch 0 is never handed out: a reader that has none — a launch without a bound — and a zeroed buffer both read as absent.)
__device__ __forceinline__ uint32_t untag_value(unsigned long long t, uint32_t epoch) {   // 0 = absent
    return (epoch != 0u && static_cast<uint32_t>(t >> 32) == epoch) ? static_cast<uint32_t>(t) : 0u;
}
// A float left under an epoch (a cutoff, a bound): -inf (nothing is ruled out) when the epoch is not the reader's.
__device__ __forceinline__ float untag_cutoff(unsigned long long t, uint32_t epoch) {
    return (epoch != 0u && static_cast<uint32_t>(t >> 32) == epoch) ? __uint_as_float(static_cast<uint32_t>(t)) : -__builtin_inff();
}
// Every wave's write-through stores have reached device scope before the workgroup counts itself out.  (A
// workgroup-scope release fence does NOT wait for global stores on this target: the ISA showed
// `s_waitcnt lgkmcnt(0); s_barrier` between the sc1 stores and the counter atomic.)
__device__ __forceinline__ void wait_own_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct SeedCtl {
    unsigned done;               // riders that have stored their sample, counted up across launches, never reset
    unsigned pad;
    unsigned long long cutoff;   // tag_value(epoch, bits of a float): what the last rider made of the sample for the next launch
};

// What a streamed launch carries for the NEXT query: after the scanners and the merger come next.n_wgs "seed riders"
// that take its sample and, with next.nbhd, one more workgroup (the LAST of the grid) for its neighbourhood.  They are
// resident from the start like everybody else — the host launches that many scanners fewer.
struct NextSeed {
    float q[kDim];             // the next query (used when query_ptr is null)
    const float* query_ptr;    // ... or where its 12 floats live (a resident row, possibly of another shard)
    long long exclude_global;
    void* out;                 // its sample buffer: kSampleSlots epoch-tagged unsigned long long (fp16 single-query scan of
                               // experiment builds: plain uint32_t[], read by the NEXT launch only)
    int n_wgs;                 // seed riders in this launch (0 = none)
    int regions;
    long long stride_rows;
    // the rider that finishes LAST turns the sample into the next launch's bound, so that launch starts scanning at
    // once instead of selecting in every workgroup
    SeedCtl* ctl;
    int topk;                  // of the next query
    int exact;                 // 8-bit replica: sample values are exact scores of the waves' best rows (one margin) or approximate ones (two)
    uint32_t epoch;            // of the next query: the tag of its sample values and of its bound
    uint32_t done_base;        // ctl->done before this launch's riders arrive
    int debug_skip;            // test hook (0 in the product): the riders do NOT store regions below this one
    int nbhd;                  // 1: the grid's last workgroup computes the next query's neighbourhood bound into out[kNbhdSlot]
    const float* anchors;      // the handle's anchor table (nbhd_anchor), or null
};

// ---- selecting from <= 2048 sample values ----------------------------------------------------------------------
struct Sample {
    uint32_t v[kHalfSeedPerThread];   // this thread's share of the sample values (ordered-u32 images; 0 = empty)
};
// (request and use are two calls: a scan asks for its sample FIRST — before its first tile: loads complete in order,
// and behind the tile the sample of a 1 M-row shard was not usable before the tile was, 3 us later — and looks at it
// after the query)
struct SampleRaw {
    unsigned long long t[kHalfSeedPerThread];
};
template <int kBlock, bool kSameLaunch = false>
__device__ __forceinline__ SampleRaw sample_request(const unsigned long long* seed_vals, int n_seed) {
    SampleRaw raw;
#pragma unroll
    for (int r = 0; r < kHalfSeedPerThread; ++r) {
        const int i = static_cast<int>(threadIdx.x) + r * kBlock;
        raw.t[r] = 0ull;
        if (i < n_seed)   // kSameLaunch: written by other workgroups of THIS launch
            raw.t[r] = kSameLaunch ? __hip_atomic_load(&seed_vals[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : seed_vals[i];
    }
    return raw;
}
// Values under another epoch than the reader's count as absent.
__device__ __forceinline__ Sample sample_finish(const SampleRaw& raw, uint32_t epoch) {
    Sample s;
#pragma unroll
    for (int r = 0; r < kHalfSeedPerThread; ++r) s.v[r] = untag_value(raw.t[r], epoch);
    return s;
}
template <int kBlock, bool kSameLaunch = false>
__device__ __forceinline__ Sample sample_load(const unsigned long long* seed_vals, int n_seed, uint32_t epoch) {
    return sample_finish(sample_request<kBlock, kSameLaunch>(seed_vals, n_seed), epoch);
}

// "Any v with at least topk sample values >= v", as high as is cheap to find — not a k-th order statistic to the last
// bit.  So ONE histogram pass instead of a radix select: the values are binned linearly over [max - kSelSpan, max]
// (1024 bins of 2.4e-4: a fiftieth of the smallest margin a replica scan then subtracts), the bins are summed from the
// top, and v is the lower edge of the bin in which the count reaches topk.  Four barriers in all against two per byte
// pass of the radix select over the 64-bit (value, index) keys (measured: 3.4 us of every workgroup's prologue
// wherever the bound is not handed over by the launch before — shards below ~4 M rows and every query alone).  A
// sample whose topk-th value lies more than kSelSpan below its maximum lands in the last bin and takes the radix
// select as before.  Returns -inf when the sample holds fewer than topk values.  Called by all kBlock threads.
constexpr int kSelBins = 1024;
constexpr float kSelSpan = 0.25f;
constexpr int kSelScratch = kSelBins + 16;   // ints of LDS scratch the selection needs (bins, block max, wave totals, result)

template <int kBlock, int kVals>
__device__ __forceinline__ float kth_of_values(const uint32_t (&vals)[kVals], int n_seed, int topk, int* s_seeds /* zeroed */,
                                               SelectSmem& s_sel, int* s_bins /* kSelScratch ints nobody else is using */) {
    static_assert(kSelBins % kBlock == 0, "whole bins per thread");
    constexpr int kPer = kSelBins / kBlock;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    float v = -__builtin_inff();
    if (n_seed <= 0) return v;   // uniform
    uint32_t vmax = 0u;
    int have = 0;
#pragma unroll
    for (int r = 0; r < kVals; ++r) {
        vmax = vals[r] > vmax ? vals[r] : vmax;
        have += vals[r] != 0u;
    }
    for (int i = tid; i < kSelScratch; i += kBlock) s_bins[i] = 0;
    vmax = wave_max_u32(vmax);
    __syncthreads();   // the scratch is zero
    if (lane == 0 && vmax) atomicMax(reinterpret_cast<unsigned int*>(&s_bins[kSelBins]), vmax);
    // ONE count per wave: 512 threads adding to the one LDS word were 2 us of serialised atomics — most of what this
    // selection cost (the phase clock had the sample values back 1.4 us after the workgroup's entry and the cutoff
    // only at 5.0 us)
    const int wave_have = __builtin_amdgcn_readlane(wave_inclusive_scan(have), 63);
    if (lane == 0 && wave_have) atomicAdd(s_seeds, wave_have);
    __syncthreads();
    if (*s_seeds < topk) return v;   // uniform
    const float smax = ordered_to_score(static_cast<uint32_t>(s_bins[kSelBins]));
#pragma unroll
    for (int r = 0; r < kVals; ++r) {
        if (vals[r]) {
            const float d = (smax - ordered_to_score(vals[r])) * (static_cast<float>(kSelBins) / kSelSpan);
            int bin = static_cast<int>(d);
            bin = (d >= 0.0f && bin < kSelBins - 1) ? bin : (d >= 0.0f ? kSelBins - 1 : 0);   // (NaN: the last bin)
            if (!(d == d)) bin = kSelBins - 1;
            atomicAdd(&s_bins[bin], 1);
        }
    }
    __syncthreads();
    // thread t owns bins [t * kPer, (t + 1) * kPer): bin 0 holds the largest values
    int mine_bins[kPer];
    int c = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        mine_bins[u] = s_bins[tid * kPer + u];
        c += mine_bins[u];
    }
    int incl = wave_inclusive_scan(c);
    if (lane == 63) s_bins[kSelBins + 1 + wave] = incl;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_bins[kSelBins + 1 + w];   // (<= 7 reads, wave-uniform)
    incl += before;
    int run = incl - c;
    if (run < topk && incl >= topk) {   // exactly one thread: the count reaches topk inside its bins
        int bin = tid * kPer;
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            if (run < topk) bin = tid * kPer + u;
            run += mine_bins[u];
        }
        s_bins[kSelBins + 12] = bin + 1;
    }
    __syncthreads();
    const int found = s_bins[kSelBins + 12] - 1;
    if (found >= 0 && found < kSelBins - 1) {   // uniform
        // every value of bins 0 .. found is >= this edge (2e-6: the rounding of the bin arithmetic)
        v = smax - static_cast<float>(found + 1) * (kSelSpan / static_cast<float>(kSelBins)) - 2.0e-6f;
    } else {   // the topk-th value lies far below the maximum (or the values are not what they should be): exact
        uint64_t keys[kVals];
#pragma unroll
        for (int r = 0; r < kVals; ++r) {
            const int i = tid + r * kBlock;
            keys[r] = vals[r] ? (static_cast<uint64_t>(vals[r]) << 32) | static_cast<uint32_t>(i + 1) : 0ull;
        }
        const uint64_t t = block_select_threshold<kBlock, kVals>(keys, topk, false, topk / 8 + 2, s_sel);
        v = ordered_to_score(static_cast<uint32_t>(t >> 32));
    }
    return v;
}

template <int kBlock>
__device__ __forceinline__ float sample_kth_value(const Sample& sample, int n_seed, int topk, int* s_seeds /* zeroed */,
                                                  SelectSmem& s_sel, int* s_bins /* kSelScratch ints nobody else is using */) {
    return kth_of_values<kBlock, kHalfSeedPerThread>(sample.v, n_seed, topk, s_seeds, s_sel, s_bins);
}

// ---- the neighbourhood of the excluded row ------------------------------------------------------------------
// Called by ALL kThreads threads of a workgroup (one barrier pair inside).  Scores up to kRows rows around local row
// (exclude_global - row_base) with the exact chain and returns the ordered-u32 image of a score v such that at least
// `topk` of them — the excluded row left out — score >= v; 0 when the excluded row is not a row of this shard (or the shard
// is smaller than a neighbourhood).  Two halves, so that a caller can REQUEST the rows before it loads its query (which
// sits behind two dependent scalar loads and a norm: this work is on the critical path of a query alone).
// How many rows: 16 x topk, in whole rounds of kThreads, at most kRows — 512 rows at top-10, all 2048 from top-100 up (the
// bound wants the topk-th best of a few times topk rows of the query's cluster, not of every row a workgroup can hold:
// at 1 M rows x top-10 the neighbourhood workgroup is the last one out of the sample launch of a query alone).
// Selection without workgroup barriers: every WAVE takes the ceil(topk / waves)-th largest of ITS rows (the wave-level
// radix select, digits relative to the wave's own range: in a tight cluster all scores lie within 1e-4 of each other and
// a fixed-width histogram cannot tell them apart — tried, and the fp32 scan's bound fell below the whole cluster), and
// v = the smallest of those: each wave holds that many rows at or above its value, so their union holds topk at or above v.
template <int kThreads, int kRows = kNbhdRows>
struct Nbhd {
    static_assert(kRows % kThreads == 0, "whole rows per thread");
    static constexpr int kPer = kRows / kThreads;
    static constexpr int kWaves = kThreads / 64;
    static constexpr int kScratch = kWaves * 256 + 4;   // ints of LDS scratch nbhd_finish needs
    Row rows[kPer];
    int64_t lo, center;
    int per;     // uniform: rows per thread in use
    bool have;   // uniform
};

template <int kThreads, int kRows = kNbhdRows>
__device__ __forceinline__ Nbhd<kThreads, kRows> nbhd_request(const float* __restrict__ feats, int64_t n, int64_t row_base,
                                                              int64_t exclude_global, int topk) {
    Nbhd<kThreads, kRows> nb;
    nb.center = exclude_global - row_base;
    int per = (16 * topk + kThreads - 1) / kThreads;
    per = per < 1 ? 1 : (per > nb.kPer ? nb.kPer : per);
    nb.per = per;
    const int64_t rows = static_cast<int64_t>(per) * kThreads;
    nb.have = nb.center >= 0 && nb.center < n && n >= rows && topk >= 1;
    int64_t lo = nb.center - rows / 2;
    if (lo > n - rows) lo = n - rows;
    if (lo < 0) lo = 0;
    nb.lo = lo;
    if (nb.have) {   // uniform
#pragma unroll
        for (int u = 0; u < nb.kPer; ++u) {
            if (u < per)   // uniform
                nb.rows[u] = load_row(feats, lo + u * kThreads + static_cast<int>(threadIdx.x));
        }
    }
    return nb;
}

// The selection: `keys` = this thread's rows (0 = none), rows_per_wave = how many rows a wave holds among them.
// `s_scratch`: Nbhd::kScratch ints of LDS nobody else is using.
template <int kThreads, int kKeys>
__device__ inline uint32_t nbhd_select(const uint64_t (&keys)[kKeys], int rows_per_wave, int topk, int* s_scratch) {
    constexpr int kWaves = kThreads / 64;
    const int tid = threadIdx.x;
    unsigned* const s_min = reinterpret_cast<unsigned*>(&s_scratch[kWaves * 256]);
    if (tid == 0) *s_min = ~0u;
    __syncthreads();
    // a wave holds rows_per_wave rows, one of which may be the excluded one
    int need = (topk + kWaves - 1) / kWaves;
    if (need > rows_per_wave - 1) need = rows_per_wave - 1;   // (then fewer than topk rows stand behind the bound: checked below)
    const uint64_t t = wave_select_threshold<kKeys>(keys, need, false, need / 8 + 1, s_scratch + (tid >> 6) * 256);
    if ((tid & 63) == 0) atomicMin(s_min, static_cast<unsigned>(t >> 32));
    __syncthreads();
    return need * kWaves >= topk ? *s_min : 0u;   // uniform
}

// ---- a query that excludes no row of this shard: where do ITS neighbours lie? ---------------------------------------
// The neighbourhood above is taken around the row a query excludes — its own, for recommendByIndex (Recommender.cu:275-318).
// A query by VALUE (mi355rec_query_topn with exclude = -1: a new track), or one whose excluded row lives on another shard,
// has no such row, and on a catalogue sorted by genre ran with the spread sample's bound alone (2-11x the by-row times,
// VERDICT r5 "missing" 3).  Its ANCHOR: kAnchorRows rows spread evenly over the shard — one every n / 4096 rows: a cluster
// of a few thousand neighbouring rows holds one or more of them — scored exactly; the best of them stands in for the
// excluded row.  On shuffled rows the anchor is an arbitrary good row and the bound around it is merely valid, as the
// neighbourhood of an excluded row is there.
// The anchors' rows are read from the handle's ANCHOR TABLE (`anchors`: a contiguous copy of those 4096 rows, 196 KB, made at
// create by one strided device-to-device copy: engine_state.hip.h, build_anchors) — taken from the matrix itself they are 4096 reads a page apart, every lane of every load
// on a line and a TLB entry of its own: the one workgroup that does this held a streamed launch up by 7 us (31.8 instead of
// 24.4 us at 10 M rows) and the thousand of a batch cost 0.6 ms.  The table only CHOOSES the centre (a stale table — the
// caller overwrote a borrowed matrix and has not rebuilt yet — still gives a valid bound: that is computed from the rows).
// `anchors` null: read from the matrix.  Called by ALL kThreads threads (one barrier pair); kAhead rows of a thread in
// flight at a time; `s_scratch`: 3 * kThreads / 64 + 1 ints of LDS nobody else is using.  Returns a LOCAL row in [0, n).
constexpr int kAnchorRows = 4096;

__host__ __device__ inline int64_t anchor_row(int64_t n, int i) {   // which row anchor i is
    const int64_t stride = n >= kAnchorRows ? n / kAnchorRows : 1;
    const int64_t r = static_cast<int64_t>(i) * stride + (stride >> 1);
    return r < n ? r : n - 1;
}

template <int kThreads, int kAhead = 4>
__device__ inline int64_t nbhd_anchor(const float* __restrict__ feats, const float* __restrict__ anchors, int64_t n,
                                      const float (&q)[kDim], float qn, int* s_scratch) {
    constexpr int kWaves = kThreads / 64;
    constexpr int kPer = kAnchorRows / kThreads;
    static_assert(kAnchorRows % kThreads == 0 && kPer % kAhead == 0, "whole rounds of kAhead rows per thread");
    const int tid = threadIdx.x;
    uint32_t best = 0u;
    int best_i = tid;
#pragma unroll 1
    for (int k0 = 0; k0 < kPer; k0 += kAhead) {
        Row rows[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int i = (k0 + u) * kThreads + tid;
            rows[u] = anchors ? load_row(anchors, static_cast<int64_t>(i)) : load_row(feats, anchor_row(n, i));
        }
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const uint32_t o = score_to_ordered(cosine_score(q, qn, rows[u]));
            if (o > best) {
                best = o;
                best_i = (k0 + u) * kThreads + tid;
            }
        }
    }
    const uint32_t top = wave_max_u32(best);
    const int holder = static_cast<int>(__builtin_ctzll(__ballot(best == top)));   // (never empty)
    const int wave_i = __builtin_amdgcn_readlane(best_i, holder);
    if ((tid & 63) == 0) {
        s_scratch[2 * (tid >> 6)] = static_cast<int>(top);
        s_scratch[2 * (tid >> 6) + 1] = wave_i;
    }
    __syncthreads();
    uint32_t all = 0u;
    int which = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {   // (uniform: every thread reads the same entries)
        const uint32_t t = static_cast<uint32_t>(s_scratch[2 * w]);
        if (w == 0 || t > all) {
            all = t;
            which = s_scratch[2 * w + 1];
        }
    }
    __syncthreads();   // (the scratch is free again)
    return anchor_row(n, which);
}

// Is the row a query excludes a row of this shard?  (Otherwise its neighbourhood is taken around its anchor.)
__device__ __forceinline__ bool nbhd_has_center(int64_t n, int64_t row_base, int64_t exclude_global) {
    return exclude_global >= row_base && exclude_global - row_base < n;
}

template <int kThreads, int kRows = kNbhdRows>
__device__ inline uint32_t nbhd_finish(const Nbhd<kThreads, kRows>& nb, const float (&q)[kDim], float qn, int topk, int* s_scratch) {
    constexpr int kPer = Nbhd<kThreads, kRows>::kPer;
    const int tid = threadIdx.x;
    if (!nb.have) return 0u;   // uniform
    uint64_t keys[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        keys[u] = 0ull;
        if (u < nb.per) {   // uniform
            const int64_t r = nb.lo + u * kThreads + tid;
            const float s = cosine_score(q, qn, nb.rows[u]);
            // unique keys: the score's image in the high word, the position in the low word
            if (r != nb.center) keys[u] = (static_cast<uint64_t>(score_to_ordered(s)) << 32) | static_cast<uint32_t>(u * kThreads + tid + 1);
        }
    }
    return nbhd_select<kThreads, kPer>(keys, 64 * nb.per, topk, s_scratch);
}

// The same bound from kRounds x kPerRound x kThreads rows with only kPerRound rows of a thread in flight at a time: for a
// workgroup that has TIME but few registers — the neighbourhood rider of a streamed fp32 scan shares the kernel's 80-VGPR
// budget, and its launch lasts 80 us.  (How tight the bound is decides what the next launch costs on a catalogue of few
// large clusters: at 300 contiguous clusters of 33 k rows a streamed fp32 query took 164 us with 1024 rows behind the
// bound — about the 14th percentile of the cluster — against 89 us with 2048.)
template <int kThreads, int kPerRound, int kRounds>
__device__ inline uint32_t nbhd_bound_rounds(const float* __restrict__ feats, int64_t n, int64_t row_base, int64_t exclude_global,
                                             const float (&q)[kDim], float qn, int topk, int* s_scratch,
                                             const float* __restrict__ anchors = nullptr) {
    constexpr int kKeys = kPerRound * kRounds;
    constexpr int64_t kRows = static_cast<int64_t>(kKeys) * kThreads;
    const int tid = threadIdx.x;
    if (!(n >= kRows && topk >= 1)) return 0u;   // uniform
    // (the query excludes no row of this shard: around its anchor — found with kPerRound rows in flight, this workgroup's budget)
    const int64_t center = nbhd_has_center(n, row_base, exclude_global) ? exclude_global - row_base
                                                                        : nbhd_anchor<kThreads, kPerRound>(feats, anchors, n, q, qn, s_scratch);
    int64_t lo = center - kRows / 2;
    if (lo > n - kRows) lo = n - kRows;
    if (lo < 0) lo = 0;
    uint64_t keys[kKeys];
#pragma unroll
    for (int round = 0; round < kRounds; ++round) {
        Row rows[kPerRound];
#pragma unroll
        for (int u = 0; u < kPerRound; ++u) rows[u] = load_row(feats, lo + (round * kPerRound + u) * kThreads + tid);
#pragma unroll
        for (int u = 0; u < kPerRound; ++u) {
            const int slot = round * kPerRound + u;
            const int64_t r = lo + slot * kThreads + tid;
            const float s = cosine_score(q, qn, rows[u]);
            keys[slot] = r != center ? (static_cast<uint64_t>(score_to_ordered(s)) << 32) | static_cast<uint32_t>(slot * kThreads + tid + 1) : 0ull;
        }
        asm volatile("" ::: "memory");   // the next round's loads stay behind this round's scores
    }
    return nbhd_select<kThreads, kKeys>(keys, 64 * kKeys, topk, s_scratch);
}

template <int kThreads, int kRows = kNbhdRows>
__device__ inline uint32_t nbhd_bound(const float* __restrict__ feats, int64_t n, int64_t row_base, int64_t exclude_global,
                                      const float (&q)[kDim], float qn, int topk, int* s_scratch,
                                      const float* __restrict__ anchors = nullptr) {
    if (!nbhd_has_center(n, row_base, exclude_global) && n >= kRows)   // uniform: around the query's anchor
        exclude_global = row_base + nbhd_anchor<kThreads>(feats, anchors, n, q, qn, s_scratch);
    const Nbhd<kThreads, kRows> nb = nbhd_request<kThreads, kRows>(feats, n, row_base, exclude_global, topk);
    return nbhd_finish<kThreads, kRows>(nb, q, qn, topk, s_scratch);
}

// The neighbourhood workgroup of a sample launch or of a streamed launch's riders: the bound goes, under the query's
// epoch, to slot kNbhdSlot of the query's sample buffer — read by the NEXT launch on the stream (plain store: a
// kernel boundary lies between).  Always stores, so that a slot never keeps an older query's value under a live epoch.
template <int kThreads, int kRows = kNbhdRows>
__device__ inline void nbhd_to_slot(const float* __restrict__ feats, int64_t n, int64_t row_base, const float* query_ptr,
                                    const float (&by_value)[kDim], int64_t exclude_global, int topk, uint32_t epoch,
                                    unsigned long long* __restrict__ sample_buf, int* s_scratch /* Nbhd<kThreads, kRows>::kScratch ints */,
                                    const float* __restrict__ anchors = nullptr) {
    float q[kDim];
    auto load_query = [&]() {
        if (query_ptr) {
#pragma unroll
            for (int j = 0; j < kDim; ++j) q[j] = query_ptr[j];
        } else {
#pragma unroll
            for (int j = 0; j < kDim; ++j) q[j] = by_value[j];
        }
    };
    uint32_t v;
    if (nbhd_has_center(n, row_base, exclude_global) || n < kRows) {   // uniform
        const Nbhd<kThreads, kRows> nb = nbhd_request<kThreads, kRows>(feats, n, row_base, exclude_global, topk);   // the rows first ...
        load_query();                                                                                               // ... then the query
        v = nbhd_finish<kThreads, kRows>(nb, q, query_norm(q), topk, s_scratch);
    } else {   // no excluded row here: the query first, its anchor, the rows around that
        load_query();
        const float qn = query_norm(q);
        const int64_t anchor = nbhd_anchor<kThreads>(feats, anchors, n, q, qn, s_scratch);
        const Nbhd<kThreads, kRows> nb = nbhd_request<kThreads, kRows>(feats, n, row_base, row_base + anchor, topk);
        v = nbhd_finish<kThreads, kRows>(nb, q, qn, topk, s_scratch);
    }
    if (threadIdx.x == 0) sample_buf[kNbhdSlot] = tag_value(epoch, v);
}

// The end of a seed rider (or of a workgroup of a sample launch) whose launch also SELECTS: it arrives, and the one
// that finds itself last loads the whole sample past its L2 and selects.  No device-wide fence: on this part a release
// / acquire pair at agent scope writes back and invalidates the whole L2 under the scanners (measured: the launch took
// 43 us instead of 28).  Instead the values are stored and loaded as device-scope atomics (write-through stores,
// L2-bypassing loads); each wave waits for ITS stores to complete (s_waitcnt vmcnt(0): a workgroup-scope fence does
// not), one thread counts, and the last workgroup's loads are issued after its counter value came back.
// Returns true in the last workgroup, with `v` = the sample's topk-th value (-inf: the sample cannot say).
// Called by all kBlock threads; s_flag / s_seeds are LDS ints, s_bins kSelScratch ints of LDS scratch.
template <int kBlock>
__device__ __forceinline__ bool sample_arrive_and_select(SeedCtl* ctl, unsigned done_base, unsigned n_arrivals,
                                                         const unsigned long long* seed_vals, int n_seed, int topk, uint32_t epoch,
                                                         int* s_flag, int* s_seeds, SelectSmem& s_sel, int* s_bins, float& v) {
    wait_own_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
        *s_seeds = 0;
        *s_flag = __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == done_base + n_arrivals;
    }
    __syncthreads();
    if (!*s_flag) return false;   // uniform
    const Sample all = sample_load<kBlock, true>(seed_vals, n_seed, epoch);
    v = sample_kth_value<kBlock>(all, n_seed, topk, s_seeds, s_sel, s_bins);
    return true;
}

// ---- the sample over the fp32 rows ---------------------------------------------------------------------------------
// A REGION is kHalfSeedBlock rows from row g * stride_rows on (stride_rows >= kHalfSeedBlock: no row is in two regions),
// one row per lane: every wave leaves the best EXACT score of its 64 rows (the excluded row left out) at
// out[g * 8 + wave], under the query's epoch, written through to device scope (the last rider of the same launch
// reads it).  Rider `rider` of `next.n_wgs` takes regions rider, rider + n_wgs, ...; kAhead regions per memory round
// trip (4 in the sample launch of a query alone, where the round trips are the launch; 2 for the riders of a streamed
// scan, which have the whole launch and share the scanners' register budget).  Called by all kHalfSeedBlock threads.
template <int kAhead = 4>
__device__ __forceinline__ void f32_sample_regions(const float* __restrict__ feats, int64_t n, int64_t row_base, const NextSeed& next,
                                                   int rider) {
    const int tid = threadIdx.x;
    unsigned long long* const out = static_cast<unsigned long long*>(next.out);
    Row rows[kAhead];
    auto load_round = [&](int g0) {
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int g = g0 + u * next.n_wgs;
            const int64_t r = static_cast<int64_t>(g < next.regions ? g : rider) * next.stride_rows + tid;
            rows[u] = load_row(feats, r < n ? r : n - 1);
        }
    };
    if (rider < next.regions) load_round(rider);   // (before the query: its 12 floats sit behind two dependent scalar loads)
    float q[kDim];
    if (next.query_ptr) {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = next.query_ptr[j];
    } else {
#pragma unroll
        for (int j = 0; j < kDim; ++j) q[j] = next.q[j];
    }
    const float qn = query_norm(q);
    for (int g0 = rider; g0 < next.regions; g0 += kAhead * next.n_wgs) {
        uint32_t v[kAhead];
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int g = g0 + u * next.n_wgs;
            const int64_t r = static_cast<int64_t>(g < next.regions ? g : rider) * next.stride_rows + tid;
            const float s = cosine_score(q, qn, rows[u]);
            const bool use = r < n && row_base + r != next.exclude_global;
            v[u] = wave_max_u32(use ? score_to_ordered(s) : 0u);
        }
        if (g0 + kAhead * next.n_wgs < next.regions) load_round(g0 + kAhead * next.n_wgs);   // uniform
#pragma unroll
        for (int u = 0; u < kAhead; ++u) {
            const int g = g0 + u * next.n_wgs;
            if (g < next.regions && g >= next.debug_skip && (tid & 63) == 0)
                __hip_atomic_store(&out[g * kHalfSeedWaves + (tid >> 6)], tag_value(next.epoch, v[u]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace mi355
