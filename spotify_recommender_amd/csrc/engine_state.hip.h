// engine_state.hip.h — the single-device handle behind the C-ABI (include/mi355rec.h): what it owns on the device, how its
// launch geometries are planned, how it is created and how its launches are timed.  Part of ONE translation unit
// (mi355rec.hip includes engine_state, engine_single, engine_batch in this order); no CPU fallback anywhere in it.
//
// Host side of the drop-in boundary: owns the device-resident catalogue shard (replaces Recommender::initialize's
// cudaMalloc / cudaMemcpy, Recommender.cu:155-168).
#pragma once

// (an MI355REC_EXPERIMENTS build is a test build too: its child-process tests break the hand-offs on purpose)
#if defined(MI355REC_EXPERIMENTS) && !defined(MI355REC_TEST_HOOKS)
#define MI355REC_TEST_HOOKS 1
#endif
#include "mi355rec_diag.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <new>
#include <cstring>
#include <string>
#include <vector>

#include "batched.hip.h"
#include "kernels.hip.h"
#include "replica.hip.h"
#include "replica_q8.hip.h"
#include "replica_multi.hip.h"

using namespace mi355;

namespace {

thread_local std::string g_last_error;

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

constexpr int kTimingPairs = 8192;
constexpr int kDirectResultSlots = 2048;   // results up to this many slots are stored straight into pinned host memory

using ScanConfig = DefaultScanCfg;
using MultiConfig = DefaultMultiCfg;
using HalfConfig = DefaultHalfCfg;
using Q8Config = DefaultQ8Cfg;
constexpr int kRideTopnMax = 640;   // largest topN whose merge rides in the next fp32 scan launch
constexpr int64_t kReplicaMinRows = 65536;      // smaller shards are created without a replica (built on demand by set_replica(ON))
constexpr int64_t kHalfAutoMinRows = 1000000;   // below this a query is launch-bound either way (measured: 11.8 vs
                                                // 13.5 us per streamed query at 1 M rows, equal at 300 k)
constexpr int kScanBlock = ScanConfig::kBlock;
constexpr int kScanTileRows = ScanConfig::kTileRows;
constexpr int64_t kF32SampleMinRows = 2000000;  // below this a fp32 scan is a dozen microseconds: no sample (the neighbourhood still rides)
constexpr int kFp32 = 0, kFp16 = 1, kQ8 = 2;    // which rows a single-query scan streams (mi355rec::Stashed::kind)
const float* const kNoQueryPtr = nullptr;   // kernel argument of the variants that take the query by value

// Launch geometry of a single-query scan over a replica of the catalogue (fp16: replica.hip.h,
// 8-bit: replica_q8.hip.h).
// The sample of the fp32 rows (handoff.hip.h, f32_sample_regions) that gives the fp32 scan its launch-wide bound, and
// the geometry of a streamed fp32 launch that carries the NEXT query's seed riders and neighbourhood workgroup.
struct F32Geom {
    int seed_grid = 0;                  // sampled regions of kHalfSeedBlock rows (0: the shard is too small to be worth a sample) ...
    int64_t seed_stride = 0;            // ... and the rows between their starts
    int riders = 0;                     // seed riders of a streamed launch (0: none)
    int nbhd = 0;                       // 1: a streamed launch also carries the next query's neighbourhood workgroup
    int r_scan = 0, r_iters = 0;        // its scanners and their tiles
};

struct ReplicaGeom {
    int grid = 0, iters = 0;            // plain launch
    int sgrid = 0, siters = 0;          // streamed launch without seed riders (one more workgroup is the merger)
    int seed_grid = 0;                  // sampled regions ...
    int64_t seed_stride = 0;            // ... and the rows between their starts
    int riders = 0;                     // seed riders of a streamed launch
    int r_scan = 0, r_iters = 0;        // its scanners and their tiles
};

}  // namespace

struct mi355rec {
    int device = 0;
    int64_t n = 0;
    int64_t row_base = 0;
    const float* d_feats = nullptr;
    float* owned_feats = nullptr;
    // LANES (mi355rec_create_lane): further handles over the SAME rows and replicas, each with its own stream state.  Once a
    // lane exists the rows the first handle owned and both replicas belong to the group: freed by whoever is destroyed last.
    struct SharedRows {
        std::atomic<int> refs{1};
        void* owned_feats = nullptr;
        void* d_half = nullptr;
        void* d_q8 = nullptr;
        float margin_mix = 0.0f, margin_mfma = 0.0f;   // of the replicas above (half_selfcheck_kernel's verdict)
    };
    SharedRows* shared = nullptr;
    bool is_lane = false;
    int lane_stream_attempts = 0;   // streams mi355rec_create_lane went through until one overlapped the parent's (0: not tested)
    int lane_overlaps = -1;         // 1: the lane's stream and its parent's run kernels side by side; 0: no such stream was found; -1: not tested

    int cus = 0;
    int grid = 0;
    int64_t rows_per_block = 0;
    int iters = 0;
    // geometry of the multi-query pass (scan_multi_kernel)
    int mgrid = 0;
    int64_t mrows_per_block = 0;
    int miters = 0;

    uint64_t* d_block_lists = nullptr;  // grid x kMaxTopK
    // streamed single queries (mi355rec_enqueue_*_streamed): the merge of query k rides in
    // the scan launch of query k + 1; two more list buffers alternate
    uint64_t* d_stream_lists[2] = {nullptr, nullptr};
    int sgrid = 0, siters = 0;          // scanning workgroups of a streamed launch (one slot is the merger's)
    F32Geom fg;                         // the fp32 scan's sample and riders
    bool streamed_ready = false;        // both list buffers exist
    bool pending = false;               // a streamed query's lists wait for their merge
    int pending_buf = 0, pending_topn = 0;
    uint64_t* pending_out = nullptr;
    // fp16 replica of the catalogue (replica.hip.h) and the geometry of the scan over it
    uint4* d_half = nullptr;            // ((n + 1) / 2) pairs of rows x 48 B
    // sample maxima: 8 bytes per entry — epoch-tagged values (8-bit scan, multi-query pass: replica.hip.h, "hand-offs
    // that fail safe"); the fp16 single-query scan uses the same buffers as plain uint32_t[]
    unsigned long long* d_half_seed = nullptr;    // kSampleSlots tagged values: the sample of the query in flight + its neighbourhood's bound
                                                  // (every handle has it and d_stream_seed / d_stream_ctl: the fp32 scan takes a bound too)
    SeedCtl* d_lone_ctl = nullptr;                // arrival counter and bound of the fp32 sample launch of a query alone ...
    unsigned lone_ctl_done = 0;                   // ... which counts up from here (never reset)
    unsigned long long* d_half_mseed = nullptr;   // kHmQueries x that: the sample of a multi-query pass over the replica
    unsigned long long* d_half_mcuts = nullptr;   // [kHmQueries] tagged cutoffs the sample launch of such a pass leaves (its last workgroup) ...
    SeedCtl* d_half_mctl = nullptr;               // ... and its arrival counter, which counts up from ...
    unsigned half_mctl_done = 0;                  // ... here (never reset)
    uint32_t epoch_ctr = 0;             // the last epoch handed out (one per query / batch whose sample or cutoff crosses workgroups; never 0)
    unsigned ctl_done[2] = {0u, 0u};    // what d_stream_ctl[i].done holds (the riders' arrival counters are never reset)
    unsigned mctl_done[2] = {0u, 0u};   // ... and d_mstream_ctl[i].done
    unsigned lone_base[9] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};   // ... and d_lone_ctr[0..8]
    // test hooks (mi355rec_debug_handoff): the next rider launch drops the sample stores of regions below this one /
    // is told a wrong arrival count, so that none of its riders is the last
    int dbg_skip_regions = 0;
    bool dbg_no_last = false;
    unsigned long long* d_half_rescored = nullptr;   // [kRideMaxLists] rows sent to the exact chain, per workgroup slot
    unsigned* d_lone_ctr = nullptr;     // [9] arrival counters of a lone query's launch (merge.hip.h, LoneTail): they count up across
                                        // launches and are never reset; lone_base is what they hold
    int64_t lone_fused = 0;             // lone queries served by one launch (scan + merge + completion word)
    int64_t half_scans = 0;             // replica scans enqueued since create ...
    int64_t q8_scans = 0;               // ... of which over the 8-bit replica
    // which route every launch of a query took (mi355rec_stats_t::route_*), since create
    struct Routes {
        int64_t fp32 = 0, fp16 = 0, q8 = 0, q8_lone = 0, multi_fp32 = 0, multi_fp16 = 0, multi_q8 = 0, mfma_two_pass = 0;
    } routes;
    ReplicaGeom hg;                     // geometry of the scan over the fp16 replica ...
    uint4* d_q8 = nullptr;              // 8-bit replica (replica_q8.hip.h): ((n + 3) / 4) quads of rows x 48 B
    float* d_anchor = nullptr;          // the anchor table (handoff.hip.h, nbhd_anchor): kAnchorRows rows x 48 B, a copy made at create
    ReplicaGeom qg;                     // ... and over the 8-bit one
    int replica_mode = 0;               // MI355REC_REPLICA_AUTO / _OFF / _ON
    bool replica_allowed = true;        // false: created with MI355REC_CREATE_NO_REPLICA
    float replica_build_ms = 0.f;
    float margin_mix = kBqMarginFlush;   // error bound the single-query replica scan may claim (v_fma_mix_f32) ...
    float margin_mfma = kBqMarginFlush;  // ... and the multi-query pass (matrix core): 1.0e-3 where the device check passes
    int pending_lists = 0;              // lists of the streamed query that waits for its merge
    // Streamed queries over the replica run ONE CALL BEHIND: query k is launched by call k + 1 (or
    // by the flush), so that its launch can carry the sample of query k + 1 (seed riders) instead
    // of a seed launch per query.
    struct Stashed {
        bool has = false;
        const float* qptr = nullptr;    // where the query's 12 floats live on a device, or null: q holds the vector
        float q[kDim] = {0};
        int64_t exclude = -1;
        int topn = 0;
        uint64_t* out = nullptr;
        int seed_buf = 0;               // which of d_stream_seed holds ITS sample maxima
        uint32_t epoch = 0;             // the tag of its sample values and of its cutoff
        int kind = 0;                   // which rows its scan streams — and its sample was taken over: kFp32, kFp16 (experiment builds), kQ8
        bool cutoff_ready = false;      // ... by riders, whose last one left the launch-wide cutoff / bound in d_stream_ctl
    } stashed;
    unsigned long long* d_stream_seed[2] = {nullptr, nullptr};
    SeedCtl* d_stream_ctl = nullptr;    // [2]: rider count and finished cutoff beside each of d_stream_seed (8-bit replica)
    // a STREAM of batches over the replica (mi355rec_enqueue_batch_keys_streamed)
    bool mstream_ready = false;
    uint64_t* d_mstream_lists[2] = {nullptr, nullptr};   // [kHmQueries][hgrid][kMultiMaxTopK], alternating
    unsigned long long* d_mstream_seed[2] = {nullptr, nullptr};    // [kHmQueries][regions * 8] tagged sample maxima, alternating
    unsigned long long* d_mstream_cuts = nullptr;    // [2][kHmQueries] tagged cutoffs the last seed rider left for the batch whose sample it completed
    SeedCtl* d_mstream_ctl = nullptr;   // [2] the riders' arrival counters
    struct MStash {
        bool has = false;
        HalfMultiArg arg;
        int nq = 0, topn = 0;
        uint64_t* out = nullptr;
        int seed_buf = 0;
        uint32_t epoch = 0;             // the tag of its sample values and of its cutoffs
        bool cuts_ready = false;        // its cutoffs were selected by the riders that took its sample (d_mstream_cuts[seed_buf])
    } mstash;
    struct MPending {
        bool has = false;
        int buf = 0, nq = 0, topn = 0, n_lists = 0;
        uint64_t* out = nullptr;
    } mpending;
    uint32_t* d_seed_vals = nullptr;    // kMultiChain x (mgrid * kSeedWaves) sample maxima
    uint64_t* d_seed_keys = nullptr;    // kMultiChain x kMultiMaxTopK: sample top-k per query of a multi-query chain

    // Every enqueue uses the handle's scratch (block lists, seed buffers) on the
    // caller's stream: consecutive uses on DIFFERENT streams are ordered with an
    // event (order_stream), so results never depend on a sync the caller forgot.
    hipStream_t last_stream = nullptr;
    bool has_last_stream = false;
    hipEvent_t order_ev = nullptr;

    // resources of the synchronous host API
    hipStream_t stream = nullptr;
    size_t slot_cap = 0;
    uint64_t* d_keys = nullptr;
    int64_t* d_idx = nullptr;
    float* d_score = nullptr;
    int64_t* h_idx = nullptr;   // pinned, mapped into the device's address space
    float* h_score = nullptr;   // pinned, mapped
    int64_t* hd_idx = nullptr;  // device-side addresses of the two pinned buffers: small results are
    float* hd_score = nullptr;  // written there by the merge kernel itself (no D2H copy launch)
    uint32_t* h_done = nullptr;   // pinned, mapped: the completion word of a synchronous single query
    uint32_t* hd_done = nullptr;
    uint32_t done_seq = 0;
    float* d_scores_full = nullptr;

    // batched path (batched.hip.h): allocated by the first batched call
    struct Batched {
        bool ready = false;
        int grid = 0;                 // workgroups of pass 1 (= groups / 2 of the threshold select)
        int grid2 = 0;                // workgroups of pass 2
        int occ1 = 0, occ2 = 0;
        float margin = kBqMarginFlush; // error bound of the fp16 pre-filter (set by the device self-check)
        int step1 = 4;                // pass 1 looks at every step1-th tile (tuning knob MI355REC_BQ_STEP1)
        int qgrid = 0, qiters = 0;    // geometry of the queued exact scan
        uint32_t* bfrag = nullptr;    // [32][64][4]
        float* qnorm = nullptr;
        float* qthr = nullptr;
        uint32_t* qflags = nullptr;
        int* cand_count = nullptr;
        uint64_t* cand_rows = nullptr;   // [1024][cand_cap] candidate records (batched.hip.h: up to 16 rows each)
        int* cand_examined = nullptr;    // [1024] rows the finalize expanded each query's records to (diagnostics)
        int cand_cap = 0;                // candidate records kept per query, also in counters[6] for the passes
        int* counters = nullptr;         // [4]
        uint32_t* special_rows = nullptr;
        uint32_t* nb_vals = nullptr;     // [1024] the queries' neighbourhood bounds (bq_prepare_kernel), ordered-u32, 0 = none
        // Does the neighbourhood bound ever beat pass 1's own threshold on THIS catalogue?  It does where similar rows lie next to
        // each other (a CSV grouped by genre) and never on shuffled rows, where its thousand workgroups are 8 us per chunk for
        // nothing.  bq_select_kernel counts the queries it won for (counters[7], cumulative); the finalize leaves that count and the
        // number of chunks that computed it in mapped host memory; the host reads them WITHOUT synchronising when it plans a chunk
        // (a stale value only delays the decision): after kBqNbProbeChunks chunks without a win the bound is computed for every
        // kBqNbProbeEvery-th chunk only, and one win switches it back on for good.  Exact either way.
        volatile int* h_nb_report = nullptr;   // mapped host memory: [0] wins, [1] chunks finished that computed the bound
        int* d_nb_report = nullptr;            // its device address
        int nb_skipped = 0;                    // chunks since the last one that computed it (host)
        bool nb_sparse = false;                // true: only every kBqNbProbeEvery-th chunk computes it
        bool nbhd_this_chunk = false;          // the chunk being enqueued computes it (launch_bq_passes -> the finalize launch)
        float* gmax = nullptr;           // [grid][32][64]
        // pass 1's per-lane maxima of the tiles it looked at, for pass 2 to skip what they rule out (batched.hip.h,
        // kTileMax): [visited tile][4][64] uint4 = 4 KiB per visited 64-row tile, 16 B per catalogue row at step 4
        uint4* tile_max = nullptr;
        int64_t tile_max_tiles = 0;      // visited tiles it has room for
                                         // (MI355REC_BATCH_MFMA_NOSKIP runs the passes without it: A/B, tests)
        int* queue = nullptr;            // [1024]
        uint64_t* qlists = nullptr;      // [1024][qgrid][kMultiMaxTopK]
        float* d_queries = nullptr;      // device copies of host queries / excludes (one chunk)
        long long* d_exclude = nullptr;
        // pinned staging ring for host queries (slot reused after its copy has completed)
        static constexpr int kSlots = 4;
        float* h_queries[kSlots] = {nullptr, nullptr, nullptr, nullptr};
        long long* h_exclude[kSlots] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t slot_ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
        bool slot_used[kSlots] = {false, false, false, false};
        int next_slot = 0;
        int launches = 0;                // chunks enqueued (stats)
        int last_count = 0;              // queries of the last chunk (what the diagnostics cover)
    } bq;
    int batch_path = 0;               // MI355REC_BATCH_AUTO / _MULTI / _MFMA

    // optional HIP-event timing of the enqueued kernels
    bool timing = false;
    int timing_stride = 1;      // time every stride-th launch of each kind
    int scan_launches = 0, merge_launches = 0;
    std::vector<hipEvent_t> ev_scan, ev_merge, ev_pass;  // (start, stop) pairs
    int n_scan_pairs = 0, n_merge_pairs = 0, n_pass_pairs = 0;
    int pass_launches = 0;
    float last_scan_ms = 0.f, last_merge_ms = 0.f, last_pass_ms = 0.f;

    std::string err;
};

namespace {

int fail(mi355rec* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    g_last_error = buf;
    return code;
}

// One epoch per query / batch whose sample values or cutoff are handed from workgroup to workgroup (never 0: a
// zeroed buffer holds no valid tag).
uint32_t next_epoch(mi355rec* h) {
    if (++h->epoch_ctr == 0u) ++h->epoch_ctr;
    return h->epoch_ctr;
}

#define HIP_TRY(h, expr)                                                          \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess)                                                     \
            return fail((h), e_ == hipErrorOutOfMemory ? MI355REC_ERR_OUT_OF_MEMORY \
                                                       : MI355REC_ERR_HIP,        \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                      \
    } while (0)

// The handle's scratch is shared by all its launches.  When the stream changes
// between two calls, the new stream first waits for everything the handle has
// enqueued on the previous one (one event record + one stream wait; nothing when
// the stream stays the same, which is the serving-loop case).
int order_stream(mi355rec* h, hipStream_t s) {
    if (h->has_last_stream && h->last_stream != s) {
        if (hipEventRecord(h->order_ev, h->last_stream) == hipSuccess) {
            HIP_TRY(h, hipStreamWaitEvent(s, h->order_ev, 0));
        } else {
            (void)hipGetLastError();  // the previous stream no longer exists: nothing left to order against
        }
    }
    h->last_stream = s;
    h->has_last_stream = true;
    return MI355REC_OK;
}

// Synchronous host API: runs on the handle's private stream, after any
// asynchronous work the caller enqueued through this handle.
int sync_api_begin(mi355rec* h) { return order_stream(h, h->stream); }

// Single-query scan: tiles of kScanTileRows rows are dealt round-robin over the
// resident workgroups (rows_per_block = 0 selects that mapping in the kernel), so
// the chip reads one moving window of the matrix — 3 % faster than a contiguous
// block of rows per workgroup (measured, tools/kbench.hip).
void plan_grid(mi355rec* h, int blocks_per_cu) {
    int64_t max_blocks = static_cast<int64_t>(h->cus) * blocks_per_cu;
    if (max_blocks > kMergeMaxLists) max_blocks = kMergeMaxLists;
    MI355REC_EXP_INT(max_blocks, "MI355REC_EXP_FP32_GRID", 1, max_blocks - 1);   // (tools/lat_exp.sh: fewer lists for the merge)
    const int64_t tiles = (h->n + kScanTileRows - 1) / kScanTileRows;
    h->grid = static_cast<int>(tiles < max_blocks ? tiles : max_blocks);
    h->rows_per_block = 0;
    h->iters = static_cast<int>((tiles + h->grid - 1) / h->grid);
    // a streamed launch: one workgroup is the merger of the query before ...
    int g = h->grid > 1 ? h->grid - 1 : 1;
    if (g > kRideMaxLists - 1) g = kRideMaxLists - 1;
    MI355REC_EXP_INT(g, "MI355REC_EXP_SGRID", 1, g - 1);
    if (tiles < g) g = static_cast<int>(tiles);
    h->sgrid = g;
    h->siters = static_cast<int>((tiles + g - 1) / g);
    // ... and, where the launch can spare them, a few are the NEXT query's seed riders and its neighbourhood workgroup
    // (handoff.hip.h): a rider takes two regions per memory round trip (~2.5 us) and should be done well before the
    // scanners (~3 us per tile each) are; sized as if it took four, which still leaves it under half of the launch
    // (10 M rows: 11 riders x 12 round trips = 30 of 80 us).
    F32Geom& f = h->fg;
    f = F32Geom();
    f.r_scan = h->sgrid;
    f.r_iters = h->siters;
    int64_t sg = h->n / kHalfSeedBlock;
    if (sg > kHalfSeedMaxGrid) sg = kHalfSeedMaxGrid;
    if (h->n >= kF32SampleMinRows && sg >= 64) {
        f.seed_grid = static_cast<int>(sg);
        f.seed_stride = h->n / sg;
    }
    if (h->grid >= 16 && h->grid == max_blocks && h->n >= kNbhdRows) {
        f.nbhd = 1;
        if (f.seed_grid > 0) {
            int rounds = static_cast<int>(h->siters * 3.0 / 12.0);
            if (rounds < 1) rounds = 1;
            int riders = (f.seed_grid + 4 * rounds - 1) / (4 * rounds);
            if (riders > h->grid / 16) riders = h->grid / 16;
            MI355REC_EXP_INT(riders, "MI355REC_EXP_F32_RIDERS", 0, h->grid / 4);
            f.riders = riders;
        }
        f.r_scan = h->grid - 1 - f.riders - f.nbhd;
        if (f.r_scan > kRideMaxLists - 1) f.r_scan = kRideMaxLists - 1;
        f.r_iters = static_cast<int>((tiles + f.r_scan - 1) / f.r_scan);
    }
}

// Multi-query pass: same round-robin tile mapping for the full pass; the seed
// kernel samples the first 512 rows of `mgrid` evenly spaced regions (mrows_per_block rows
// apart) so that the sample also represents catalogues that are ordered.
void plan_multi_grid(mi355rec* h, int blocks_per_cu) {
    int64_t max_blocks = static_cast<int64_t>(h->cus) * blocks_per_cu;
    if (max_blocks > kMergeMaxLists) max_blocks = kMergeMaxLists;
    const int64_t tiles = (h->n + MultiConfig::kTileRows - 1) / MultiConfig::kTileRows;
    h->mgrid = static_cast<int>(tiles < max_blocks ? tiles : max_blocks);
    h->miters = static_cast<int>((tiles + h->mgrid - 1) / h->mgrid);
    int64_t stride = h->n / h->mgrid;
    stride = stride / 64 * 64;
    if (stride < MultiConfig::kTileRows) stride = MultiConfig::kTileRows;
    h->mrows_per_block = stride;  // seed kernel only: distance between sampled regions
}

// Scan over a replica: tiles of `tile_rows` rows dealt round-robin; the seed kernel samples
// `tile_rows` rows of up to 256 evenly spaced regions (>= tile_rows apart, so no row is sampled
// twice; starts are multiples of `align` rows, the replica's packing unit).
ReplicaGeom plan_replica(const mi355rec* h, int occ, int tile_rows, int align, double us_per_tile) {
    ReplicaGeom g;
    if (occ < 1) occ = 1;
    if (occ > 3) occ = 3;
    int64_t max_blocks = static_cast<int64_t>(h->cus) * occ;
    if (max_blocks > kRideMaxLists) max_blocks = kRideMaxLists;
    MI355REC_EXP_INT(max_blocks, "MI355REC_EXP_REPLICA_GRID", 1, max_blocks - 1);
    const int64_t tiles = (h->n + tile_rows - 1) / tile_rows;
    g.grid = static_cast<int>(tiles < max_blocks ? tiles : max_blocks);
    g.iters = static_cast<int>((tiles + g.grid - 1) / g.grid);
    g.sgrid = g.grid > 1 ? g.grid - 1 : 1;
    g.siters = static_cast<int>((tiles + g.sgrid - 1) / g.sgrid);
    int64_t sg = h->n / tile_rows;
    if (sg > kHalfSeedMaxGrid) sg = kHalfSeedMaxGrid;
    g.seed_grid = static_cast<int>(sg);
    g.seed_stride = sg > 0 ? (h->n / sg) / align * align : 0;
    // seed riders of a streamed launch: each takes four regions per memory round trip (~2 us) and
    // should be done well before the scanners (us_per_tile each) are
    g.riders = 0;
    g.r_scan = g.sgrid;
    g.r_iters = g.siters;
    if (sg > 0 && g.grid >= 16) {
        int rounds = static_cast<int>(g.siters * us_per_tile / 12.0);
        if (rounds < 1) rounds = 1;
        int riders = static_cast<int>((sg + 4 * rounds - 1) / (4 * rounds));
        if (riders > g.grid / 8) riders = g.grid / 8;
        MI355REC_EXP_INT(riders, "MI355REC_EXP_RIDERS", 0, g.grid / 2);
        if (riders > 0) {
            g.riders = riders;
            g.r_scan = g.grid - 2 - riders;   // (the merger, and the next query's neighbourhood workgroup)
            g.r_iters = static_cast<int>((tiles + g.r_scan - 1) / g.r_scan);
            // Whole rounds only: where the cap above binds (a 1 M-row shard: 61 riders for 256 regions) the last few
            // regions would cost every rider's launch one more round trip — on a shard that small the riders are the
            // last workgroups out (phase clock: 9.5 us of a 9.5 us launch) and its sample is half its rows anyway.
            const int64_t whole = static_cast<int64_t>(riders) * 4 * rounds;
            if (whole < sg && whole >= 64) {
                sg = whole;
                g.seed_grid = static_cast<int>(sg);
                g.seed_stride = (h->n / sg) / align * align;
            }
        }
    }
    return g;
}

void plan_half_grid(mi355rec* h) {
    int occ = 0;
    // the fp16 replica: the multi-query pass's workgroups and sampled regions (experiment builds: also the single-query scan's)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_half_multi_kernel<false, false>, kHmBlock, 0) != hipSuccess) occ = 1;
    MI355REC_EXP_INT(occ, "MI355REC_EXP_HOCC", 1, 4);
    h->hg = plan_replica(h, occ, HalfConfig::kTileRows, 2, 2.1);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_q8_kernel<Q8Config, true, false>, Q8Config::kBlock, 0) != hipSuccess) occ = 1;
    h->qg = plan_replica(h, occ, Q8Config::kTileRows, 4, 2.1);
}

void free_replica(mi355rec* h) {
    // (the replicas themselves stay while a group of lanes shares them: mi355rec_destroy releases the group's reference)
    // (... the GROUP's: a replica this member was still building when the build failed is its own)
    const bool groups_half = h->shared && h->shared->d_half == h->d_half, groups_q8 = h->shared && h->shared->d_q8 == h->d_q8;
    void* bufs[] = {groups_half ? nullptr : h->d_half, groups_q8 ? nullptr : h->d_q8, h->d_half_mseed, h->d_half_rescored, h->d_half_mcuts,
                    h->d_half_mctl};
    h->d_half_mcuts = nullptr;
    h->d_half_mctl = nullptr;
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    h->d_half = nullptr;
    h->d_q8 = nullptr;
    h->d_half_mseed = nullptr;
    h->d_half_rescored = nullptr;
}

// The per-handle state that goes with a replica (sample values, cutoffs and arrival counters of the multi-query pass,
// the "rows re-scored" slots): a lane has its own, over replicas it shares.
int alloc_replica_state(mi355rec* h);
int alloc_replica(mi355rec* h, int64_t n_padded) {
    HIP_TRY(h, hipMalloc(&h->d_half, static_cast<size_t>(n_padded) * 24));
    HIP_TRY(h, hipMalloc(&h->d_q8, static_cast<size_t>((h->n + 3) / 4) * 48));
    return alloc_replica_state(h);
}

int alloc_replica_state(mi355rec* h) {
    HIP_TRY(h, hipMalloc(&h->d_half_mseed, sizeof(unsigned long long) * kHmSampleSlots));
    HIP_TRY(h, hipMalloc(&h->d_half_rescored, sizeof(unsigned long long) * kRideMaxLists));
    HIP_TRY(h, hipMalloc(&h->d_half_mcuts, sizeof(unsigned long long) * kHmQueries));
    HIP_TRY(h, hipMemsetAsync(h->d_half_mcuts, 0, sizeof(unsigned long long) * kHmQueries, h->stream));
    HIP_TRY(h, hipMalloc(&h->d_half_mctl, sizeof(SeedCtl)));
    HIP_TRY(h, hipMemsetAsync(h->d_half_mctl, 0, sizeof(SeedCtl), h->stream));
    h->half_mctl_done = 0;
    HIP_TRY(h, hipMemsetAsync(h->d_half_rescored, 0, sizeof(unsigned long long) * kRideMaxLists, h->stream));
    return MI355REC_OK;
}

// (Re)builds the replica from the fp32 rows on the handle's stream and waits for it.  All or
// nothing: after a failure the handle has NO replica (d_half and everything keyed on it is null)
// and keeps serving from the fp32 rows.
int build_replica_inner(mi355rec* h);
int build_replica(mi355rec* h) {
    const int rc = build_replica_inner(h);
    if (rc != MI355REC_OK) {
        (void)hipStreamSynchronize(h->stream);
        free_replica(h);
        if (h->replica_mode == MI355REC_REPLICA_ON || h->replica_mode == MI355REC_REPLICA_FP16) h->replica_mode = MI355REC_REPLICA_AUTO;
    }
    return rc;
}

int build_replica_inner(mi355rec* h) {
    const int64_t n_padded = (h->n + 1) & ~static_cast<int64_t>(1);
    if (!h->d_half) {
        const int rc = alloc_replica(h, n_padded);
        if (rc != MI355REC_OK) return rc;
    }
    hipEvent_t a = nullptr, b = nullptr;
    const bool timed = hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess;
    if (timed) (void)hipEventRecord(a, h->stream);
    hipLaunchKernelGGL(replica_build_kernel, dim3(static_cast<unsigned>((n_padded + 255) / 256)), dim3(256), 0, h->stream,
                       h->d_feats, h->n, n_padded, reinterpret_cast<uint2*>(h->d_half));
    const int64_t n_quads4 = (h->n + 3) / 4 * 4;
    hipLaunchKernelGGL(q8_build_kernel, dim3(static_cast<unsigned>((n_quads4 + 255) / 256)), dim3(256), 0, h->stream,
                       h->d_feats, h->n, n_quads4, reinterpret_cast<uint32_t*>(h->d_q8));
    if (timed) (void)hipEventRecord(b, h->stream);
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (timed && e == hipSuccess) (void)hipEventElapsedTime(&h->replica_build_ms, a, b);
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    HIP_TRY(h, e);
    HIP_TRY(h, hipGetLastError());
    // which error bound the pre-filters may claim on this device (replica.hip.h, half_selfcheck_kernel)
    float* scratch = reinterpret_cast<float*>(h->d_half_seed);   // (any scratch of >= 16 bytes will do)
    hipLaunchKernelGGL(half_selfcheck_kernel, dim3(1), dim3(64), 0, h->stream, scratch);
    float chk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    HIP_TRY(h, hipMemcpyAsync(chk, scratch, sizeof chk, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const bool cvt_kept = chk[1] > 2.9e-6f && chk[1] < 3.1e-6f;
    h->margin_mfma = (cvt_kept && chk[0] == 9.5367431640625e-07f) ? kBqMargin : kBqMarginFlush;
    h->margin_mix = (cvt_kept && chk[2] == 9.5367431640625e-07f && chk[3] == 9.5367431640625e-07f) ? kBqMargin : kBqMarginFlush;
    return MI355REC_OK;
}

// The anchor table: a contiguous copy of kAnchorRows rows spread evenly over the shard (handoff.hip.h).  A snapshot, like the
// replicas: rebuilt by mi355rec_rebuild_replica; a stale one only picks a poorer centre (the bound is computed from the rows).
int build_anchors(mi355rec* h) {
    if (!h->d_anchor) return MI355REC_OK;
    // anchor i = row i * stride + stride / 2 (anchor_row): one strided 2-D copy, 48 B per row; a shard with fewer rows than
    // anchors (stride 1) fills what it has and leaves the rest zero (a zero row scores 0: never the best of a useful search)
    const int64_t stride = h->n >= kAnchorRows ? h->n / kAnchorRows : 1;
    const size_t rows = h->n >= kAnchorRows ? kAnchorRows : static_cast<size_t>(h->n);
    HIP_TRY(h, hipMemsetAsync(h->d_anchor, 0, sizeof(float) * kDim * kAnchorRows, h->stream));
    HIP_TRY(h, hipMemcpy2DAsync(h->d_anchor, sizeof(float) * kDim, h->d_feats + (stride >> 1) * kDim, sizeof(float) * kDim * static_cast<size_t>(stride),
                                sizeof(float) * kDim, rows, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MI355REC_OK;
}

int create_common(const float* feats, bool on_device, int64_t n, int dim, int device,
                  int64_t row_base, int flags, mi355rec_t** out) {
    if (out) *out = nullptr;
    if (!out || (!feats && n != 0)) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null argument");
    if (flags & ~MI355REC_CREATE_NO_REPLICA) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "unknown create flags 0x%x", flags);
    if (dim != kDim) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "dim must be %d, got %d", kDim, dim);
    // n == 0 is an EMPTY SHARD (a rank of a row-sharded catalogue with more ranks
    // than rows): every query answers with an all-empty list, merges work as usual.
    if (n < 0 || row_base < 0 || n + row_base > 0xfffffffell)
        return fail(nullptr, MI355REC_ERR_INVALID_ARG, "rows %lld (base %lld) out of range",
                    (long long)n, (long long)row_base);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(nullptr, MI355REC_ERR_NO_DEVICE,
                    "no HIP device visible: the MI355X engine has no CPU fallback");
    if (device < 0 || device >= count)
        return fail(nullptr, MI355REC_ERR_INVALID_ARG, "device %d not in [0,%d)", device, count);

    DeviceGuard guard(device);
    if (!guard.ok) return fail(nullptr, MI355REC_ERR_HIP, "hipSetDevice(%d) failed", device);

    mi355rec* h = new mi355rec();
    h->device = device;
    h->n = n;
    h->row_base = row_base;

    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        delete h;
        return fail(nullptr, MI355REC_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    }
    h->cus = prop.multiProcessorCount;

    if (n > 0) {
        int occ = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, scan_kernel<ScanConfig, true, false>, kScanBlock, 0);
        if (e != hipSuccess || occ < 1) occ = 1;
        if (occ > 4) occ = 4;
        plan_grid(h, occ);
        int mocc = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&mocc, scan_multi_kernel<MultiConfig>, MultiConfig::kBlock, 0);
        if (e != hipSuccess || mocc < 1) mocc = 1;
        if (mocc > 4) mocc = 4;
        plan_multi_grid(h, mocc);
        plan_half_grid(h);
    } else {
        h->grid = h->mgrid = 1;  // sizes the (unused) scratch; no scan is ever launched
    }

    int rc = MI355REC_OK;
    auto cleanup = [&](int code, const char* what, hipError_t he) {
        rc = fail(nullptr, code, "%s: %s", what, hipGetErrorString(he));
        mi355rec_destroy(h);
        return rc;
    };

    if (n == 0) {
        h->d_feats = nullptr;
    } else if (on_device) {
        if (reinterpret_cast<uintptr_t>(feats) & 15) {
            delete h;
            return fail(nullptr, MI355REC_ERR_INVALID_ARG, "device matrix must be 16-byte aligned");
        }
        h->d_feats = feats;
    } else {
        const size_t bytes = static_cast<size_t>(n) * kDim * sizeof(float);
        if ((e = hipMalloc(&h->owned_feats, bytes)) != hipSuccess)
            return cleanup(MI355REC_ERR_OUT_OF_MEMORY, "hipMalloc(catalogue)", e);
        if ((e = hipMemcpy(h->owned_feats, feats, bytes, hipMemcpyHostToDevice)) != hipSuccess)
            return cleanup(MI355REC_ERR_HIP, "hipMemcpy(catalogue H2D)", e);
        h->d_feats = h->owned_feats;
    }

    int single_lists = h->grid > h->hg.grid ? h->grid : h->hg.grid;
    if (h->qg.grid > single_lists) single_lists = h->qg.grid;
    size_t list_words = static_cast<size_t>(single_lists) * kMaxTopK;
    const size_t multi_words = static_cast<size_t>(h->mgrid > h->hg.grid ? h->mgrid : h->hg.grid) * kMultiChain * kMultiMaxTopK;
    if (multi_words > list_words) list_words = multi_words;
    if ((e = hipMalloc(&h->d_block_lists, sizeof(uint64_t) * list_words)) != hipSuccess)
        return cleanup(MI355REC_ERR_OUT_OF_MEMORY, "hipMalloc(block lists)", e);
    if ((e = hipMalloc(&h->d_lone_ctr, sizeof(unsigned) * 16)) != hipSuccess || (e = hipMemset(h->d_lone_ctr, 0, sizeof(unsigned) * 16)) != hipSuccess)
        return cleanup(MI355REC_ERR_OUT_OF_MEMORY, "hipMalloc(lone counters)", e);
    // the sample / bound buffers of single queries (handoff.hip.h): every scan takes a launch-wide bound, over whichever rows
    {
        unsigned long long** seeds[] = {&h->d_half_seed, &h->d_stream_seed[0], &h->d_stream_seed[1]};
        for (unsigned long long** b : seeds)
            if ((e = hipMalloc(b, sizeof(unsigned long long) * kSampleSlots)) != hipSuccess ||
                (e = hipMemset(*b, 0, sizeof(unsigned long long) * kSampleSlots)) != hipSuccess)
                return cleanup(MI355REC_ERR_OUT_OF_MEMORY, "hipMalloc(sample values)", e);
        if ((e = hipMalloc(&h->d_stream_ctl, sizeof(SeedCtl) * 2)) != hipSuccess || (e = hipMemset(h->d_stream_ctl, 0, sizeof(SeedCtl) * 2)) != hipSuccess ||
            (e = hipMalloc(&h->d_lone_ctl, sizeof(SeedCtl))) != hipSuccess || (e = hipMemset(h->d_lone_ctl, 0, sizeof(SeedCtl))) != hipSuccess)
            return cleanup(MI355REC_ERR_OUT_OF_MEMORY, "hipMalloc(sample control)", e);
    }
    if ((e = hipMalloc(&h->d_seed_vals, sizeof(uint32_t) * kMultiChain * static_cast<size_t>(h->mgrid) * kSeedWaves)) != hipSuccess)
        return cleanup(MI355REC_ERR_OUT_OF_MEMORY, "hipMalloc(seed values)", e);
    if ((e = hipMalloc(&h->d_seed_keys, sizeof(uint64_t) * kMultiChain * kMultiMaxTopK)) != hipSuccess)
        return cleanup(MI355REC_ERR_OUT_OF_MEMORY, "hipMalloc(seed keys)", e);
    if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess)
        return cleanup(MI355REC_ERR_HIP, "hipStreamCreate", e);
    if (n >= kNbhdRows && (e = hipMalloc(&h->d_anchor, sizeof(float) * kDim * kAnchorRows)) != hipSuccess)
        return cleanup(MI355REC_ERR_OUT_OF_MEMORY, "hipMalloc(anchor table)", e);
    if ((e = hipEventCreateWithFlags(&h->order_ev, hipEventDisableTiming)) != hipSuccess)
        return cleanup(MI355REC_ERR_HIP, "hipEventCreate", e);
    if (on_device) {
        // A borrowed matrix may still be being written by a kernel on some caller
        // stream (e.g. a torch generator): wait once, here, so that no query can
        // scan it half-written.  Later writes to it are the caller's to order.
        if ((e = hipDeviceSynchronize()) != hipSuccess)
            return cleanup(MI355REC_ERR_HIP, "hipDeviceSynchronize", e);
    }
    // The replicas (fp16 + 8-bit: +75 % device memory, one pass over the rows) unless the caller asked for a handle
    // without them (MI355REC_CREATE_NO_REPLICA: 48 B per row resident instead of 84).
    // Shards below kReplicaMinRows get none: no AUTO path reads it there (single queries switch over at 1 M
    // rows, batches at 65536); mi355rec_set_replica(ON) builds it on demand.  If the +75 % cannot be had the
    // handle degrades to fp32-only (same results, 48 B/row) and says so in mi355rec_last_error.
    if (h->d_anchor) {
        const int arc = build_anchors(h);
        if (arc != MI355REC_OK) {
            rc = arc;
            mi355rec_destroy(h);
            return rc;
        }
    }
    h->replica_allowed = (flags & MI355REC_CREATE_NO_REPLICA) == 0;
    if (n >= kReplicaMinRows && h->replica_allowed) {
        const int brc = build_replica(h);
        if (brc != MI355REC_OK) {
            (void)hipGetLastError();
            h->err = "fp16 replica not built (" + h->err + "): this handle serves from the fp32 rows only";
        }
    }

    *out = h;
    return MI355REC_OK;
}

// Result slots of the synchronous host API (device + pinned host mirrors).
int ensure_slots(mi355rec* h, size_t slots) {
    if (slots <= h->slot_cap) return MI355REC_OK;
    size_t cap = h->slot_cap ? h->slot_cap : 1024;
    while (cap < slots) cap *= 2;
    if (h->d_keys) (void)hipFree(h->d_keys);
    if (h->d_idx) (void)hipFree(h->d_idx);
    if (h->d_score) (void)hipFree(h->d_score);
    if (h->h_idx) (void)hipHostFree(h->h_idx);
    if (h->h_score) (void)hipHostFree(h->h_score);
    h->d_keys = nullptr; h->d_idx = nullptr; h->d_score = nullptr;
    h->h_idx = nullptr; h->h_score = nullptr;
    h->slot_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_keys, cap * sizeof(uint64_t)));
    HIP_TRY(h, hipMalloc(&h->d_idx, cap * sizeof(int64_t)));
    HIP_TRY(h, hipMalloc(&h->d_score, cap * sizeof(float)));
    HIP_TRY(h, hipHostMalloc(&h->h_idx, cap * sizeof(int64_t), hipHostMallocMapped));
    HIP_TRY(h, hipHostMalloc(&h->h_score, cap * sizeof(float), hipHostMallocMapped));
    HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hd_idx), h->h_idx, 0));
    HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hd_score), h->h_score, 0));
    if (!h->h_done) {
        HIP_TRY(h, hipHostMalloc(&h->h_done, sizeof(uint32_t), hipHostMallocMapped));
        *h->h_done = 0u;
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hd_done), h->h_done, 0));
    }
    h->slot_cap = cap;
    return MI355REC_OK;
}

int timing_begin(mi355rec* h, std::vector<hipEvent_t>& evs, int& pairs, int& launches, hipStream_t s) {
    if (!h->timing) return -1;
    if ((launches++ % h->timing_stride) != 0) return -1;
    if (pairs >= kTimingPairs) return -1;
    if (static_cast<int>(evs.size()) < 2 * (pairs + 1)) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return -1;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1; }
        evs.push_back(a);
        evs.push_back(b);
    }
    (void)hipEventRecord(evs[2 * pairs], s);
    return pairs;
}

void timing_end(mi355rec* h, std::vector<hipEvent_t>& evs, int& pairs, int slot, hipStream_t s) {
    (void)h;
    if (slot < 0) return;
    (void)hipEventRecord(evs[2 * slot + 1], s);
    pairs = slot + 1;
}

// The event pair of the next timed launch, NOT recorded: LAUNCH_TIMED hands it to the dispatch
// itself (hipExtLaunchKernelGGL), so it stamps the kernel's own start and end — the same
// interval rocprofv3 reports — instead of two extra stream commands around the launch (those
// bracket the dispatch too: +3 us on a 40 us kernel).
int timing_slot(mi355rec* h, std::vector<hipEvent_t>& evs, int& pairs, int& launches) {
    if (!h->timing) return -1;
    if ((launches++ % h->timing_stride) != 0) return -1;
    if (pairs >= kTimingPairs) return -1;
    if (static_cast<int>(evs.size()) < 2 * (pairs + 1)) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return -1;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return -1; }
        evs.push_back(a);
        evs.push_back(b);
    }
    return pairs;
}

#define LAUNCH_TIMED(h, evs, pairs, launches, kernel, grid, block, s, ...)                                  \
    do {                                                                                                    \
        const int slot_ = timing_slot((h), (evs), (pairs), (launches));                                     \
        if (slot_ >= 0) {                                                                                   \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, s, (evs)[2 * slot_], (evs)[2 * slot_ + 1], 0,      \
                                  __VA_ARGS__);                                                             \
            (pairs) = slot_ + 1;                                                                            \
        } else {                                                                                            \
            hipLaunchKernelGGL(kernel, grid, block, 0, s, __VA_ARGS__);                                     \
        }                                                                                                   \
    } while (0)

}  // namespace
