// mi355rec.hip — C-ABI (include/mi355rec.h) over the gfx950 kernels.
//
// Host side of the drop-in boundary: owns the device-resident catalogue shard
// (replaces Recommender::initialize's cudaMalloc/cudaMemcpy,
// Recommender.cu:155-168), launches the fused scan + merge (replaces
// calculateSimilarities + the host heap, Recommender.cu:184-254,293-315) and
// hands back indices/scores.  No CPU fallback anywhere in this file.
#include "mi355rec.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
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
        uint32_t* cand_rows = nullptr;   // [1024][cand_cap]
        int cand_cap = 0;                // candidate rows kept per query, also in counters[6] for the passes
        int* counters = nullptr;         // [4]
        uint32_t* special_rows = nullptr;
        uint32_t* nb_vals = nullptr;     // [1024] the queries' neighbourhood bounds (bq_prepare_kernel), ordered-u32, 0 = none
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
    // (handoff.hip.h): a rider takes four regions per memory round trip (~2.5 us) and should be done well before the
    // scanners (~3 us per tile each) are.
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
    void* bufs[] = {h->d_half, h->d_q8, h->d_half_mseed, h->d_half_rescored, h->d_half_mcuts, h->d_half_mctl};
    h->d_half_mcuts = nullptr;
    h->d_half_mctl = nullptr;
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    h->d_half = nullptr;
    h->d_q8 = nullptr;
    h->d_half_mseed = nullptr;
    h->d_half_rescored = nullptr;
}

int alloc_replica(mi355rec* h, int64_t n_padded) {
    HIP_TRY(h, hipMalloc(&h->d_half, static_cast<size_t>(n_padded) * 24));
    HIP_TRY(h, hipMalloc(&h->d_q8, static_cast<size_t>((h->n + 3) / 4) * 48));
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

bool use_half(const mi355rec* h, const uint64_t* upper_dev) {
    if (!h->d_half || upper_dev || h->replica_mode == MI355REC_REPLICA_OFF) return false;
    return h->replica_mode == MI355REC_REPLICA_ON || h->replica_mode == MI355REC_REPLICA_FP16 || h->n >= kHalfAutoMinRows;
}

// Single queries stream the 8-bit replica (half the fp16 one's bytes per row); experiment builds can keep them on the
// fp16 one (MI355REC_REPLICA_FP16: A/B).
bool use_q8(const mi355rec* h) { return h->d_q8 && h->replica_mode != MI355REC_REPLICA_FP16; }

// Which rows the next single query on this handle streams.
int single_kind(const mi355rec* h, const uint64_t* upper_dev) {
    if (!use_half(h, upper_dev)) return kFp32;
    return use_q8(h) ? kQ8 : kFp16;
}

// Streamed launches over the 8-bit replica: the last seed rider out turns the sample into the next launch's
// cutoff (saves a ~4 us select in every workgroup of that launch).  The riders then take sample + select
// (~10 us) in all, so only where the scanners run longer than that.
bool q8_hoists(const mi355rec* h) { return h->qg.riders > 0 && h->qg.r_iters >= 5; }
// The sample holds EXACT scores of its rows (one margin in the cutoff instead of two: a third of the candidates)
// where the extra fetch per sampled wave is not on the launch's critical path.
bool q8_exact_sample(const mi355rec* h) { return h->qg.iters >= 3; }

// Is the row a query excludes a row of THIS shard?  Then its neighbourhood gives the scan a bound (handoff.hip.h).
bool nbhd_applies(const mi355rec* h, int64_t exclude_global) {
    return exclude_global >= h->row_base && exclude_global < h->row_base + h->n && h->n >= kNbhdRows;
}

// The sample launch of a query ALONE over a replica (the first query of a stream as well): the sampled regions and,
// when the excluded row is a row of this shard, one more workgroup for its neighbourhood.  The values are tagged with
// `epoch`, which the scan that reads them is given as well.
void enqueue_half_seed(mi355rec* h, int kind, const float* qptr, const QueryArg& qa, int64_t exclude_global, int topn,
                       unsigned long long* seed_buf, uint32_t epoch, hipStream_t s) {
    if (kind == kQ8) {
        const int extra = nbhd_applies(h, exclude_global) ? 1 : 0;
        if (h->qg.seed_grid + extra <= 0) return;
#define SEED_Q8(EXACT)                                                                                                     \
    hipLaunchKernelGGL((seed_q8_kernel<EXACT>), dim3(h->qg.seed_grid + extra), dim3(kHalfSeedBlock), 0, s, h->d_feats, h->d_q8, \
                       h->n, h->qg.seed_stride, h->row_base, qa, qptr, exclude_global, seed_buf, epoch, h->qg.seed_grid, topn)
        if (q8_exact_sample(h)) SEED_Q8(true);
        else SEED_Q8(false);
#undef SEED_Q8
        return;
    }
#ifdef MI355REC_EXPERIMENTS
    if (h->hg.seed_grid <= 0) return;
    uint32_t* const seed_out = reinterpret_cast<uint32_t*>(seed_buf);   // the fp16 scan's plain values
    if (qptr) {
        hipLaunchKernelGGL((seed_half_kernel<true>), dim3(h->hg.seed_grid), dim3(kHalfSeedBlock), 0, s, h->d_feats, h->d_half,
                           h->n, h->hg.seed_stride, h->row_base, qa, qptr, exclude_global, seed_out);
    } else {
        hipLaunchKernelGGL((seed_half_kernel<false>), dim3(h->hg.seed_grid), dim3(kHalfSeedBlock), 0, s, h->d_feats, h->d_half,
                           h->n, h->hg.seed_stride, h->row_base, qa, kNoQueryPtr, exclude_global, seed_out);
    }
#endif
}

// The same for a query alone over the fp32 rows (kernels.hip.h, seed_f32_kernel): the regions' last workgroup leaves the
// bound in `ctl`, the neighbourhood workgroup its own in seed_buf[kNbhdSlot].  `*ctl_done` is what ctl->done holds (the
// counter is never reset).  Returns whether a sample (hence a bound in `ctl`) was enqueued.
bool enqueue_f32_seed(mi355rec* h, const float* qptr, const float* query12, int64_t exclude_global, int topn,
                      unsigned long long* seed_buf, SeedCtl* ctl, unsigned* ctl_done, uint32_t epoch, hipStream_t s) {
    NextSeed sd;
    std::memset(&sd, 0, sizeof sd);
    sd.query_ptr = qptr;
    if (!qptr) std::memcpy(sd.q, query12, sizeof sd.q);
    sd.exclude_global = exclude_global;
    sd.out = seed_buf;
    sd.regions = h->fg.seed_grid;
    sd.n_wgs = h->fg.seed_grid;
    sd.stride_rows = h->fg.seed_stride;
    sd.ctl = sd.regions > 0 ? ctl : nullptr;
    sd.topk = topn;
    sd.epoch = epoch;
    sd.done_base = *ctl_done + (h->dbg_no_last ? 0x40000000u : 0u);
    sd.debug_skip = h->dbg_skip_regions;
    sd.nbhd = nbhd_applies(h, exclude_global) ? 1 : 0;
    if (sd.regions + sd.nbhd <= 0) return false;
    hipLaunchKernelGGL(seed_f32_kernel, dim3(sd.regions + sd.nbhd), dim3(kHalfSeedBlock), 0, s, h->d_feats, h->n, h->row_base, sd);
    if (sd.regions > 0) {
        *ctl_done += static_cast<unsigned>(sd.regions);
        h->dbg_no_last = false;   // (test hooks of mi355rec_debug_handoff: they apply to ONE sampling launch)
        h->dbg_skip_regions = 0;
    }
    return sd.regions > 0;
}

// Enqueue the scan for one query.  qptr != null: the kernel reads the query's 12 floats from there
// (a resident row, or any other device-readable address).
// *n_lists = per-workgroup lists it leaves in d_block_lists.
// lone != null (a lone query whose caller waits on the host): over the 8-bit replica of a large shard the launch
// also merges its own lists into lone's buffers (merge.hip.h, lone_tail) and *fused is set.
constexpr int64_t kLoneFusedMinRows = 4000000;
// A query alone over the fp32 rows gets a sample launch of its own (~5 us) from here up: below, the scan is a dozen
// microseconds and launch-bound.
constexpr int64_t kF32LoneSeedMinRows = 4000000;
// (Round 4 had a LONE synchronous query below 1.5 M rows read the fp32 rows — two launches against the replica's three
// were worth more than the bytes: 27.3 against 29.7 us at 1 M rows.  Once the 8-bit scan's prologue had been fixed —
// sample requested before the first tile, one LDS atomic per wave in its selection — the replica won from 1 M rows up
// again (tools/route_thresholds.sh: 25.7 against 27.4 us at 1 M, 27.1 against 30.4 at 1.4 M, 28.3 against 39.9 at 3 M;
// 28.0 against 24.3 at 0.7 M), which is where single queries take it anyway: the rule is gone.)
int enqueue_scan(mi355rec* h, const float* qptr, const float* query12,
                 int64_t exclude_global, int topn, const uint64_t* upper_dev, hipStream_t s, int* n_lists,
                 const LoneTail* lone = nullptr, bool* fused = nullptr) {
    QueryArg qa;
    std::memset(&qa, 0, sizeof qa);
    qa.margin = h->margin_mix;
    if (!qptr) std::memcpy(qa.q, query12, sizeof qa.q);
    const PrevMerge none{nullptr, 0, 0, nullptr};
    const LoneTail no_tail{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};
    NextSeed no_next;
    std::memset(&no_next, 0, sizeof no_next);
    if (fused) *fused = false;
    const int kind = single_kind(h, upper_dev);
    if (kind == kQ8) {
        ++h->half_scans;
        *n_lists = h->qg.grid;
        ++h->q8_scans;
        const uint32_t epoch = next_epoch(h);
        enqueue_half_seed(h, kQ8, qptr, qa, exclude_global, topn, h->d_half_seed, epoch, s);
        const int q8_seeds = (q8_exact_sample(h) ? -1 : 1) * h->qg.seed_grid * kHalfSeedWaves;   // (negative: exact values)
        const unsigned long long* const no_cutoff = nullptr;
        if (lone && h->n >= kLoneFusedMinRows) {
            ++h->routes.q8_lone;
            // the arrival counters of the launch's tail count up and are never reset: this launch starts from ...
            LoneTail tail = *lone;
            const unsigned grid = static_cast<unsigned>(h->qg.grid);
            for (unsigned g = 0; g < 8u; ++g) tail.base[g] = h->lone_base[g];
            tail.base[8] = h->lone_base[8];
            if (qptr) {
                LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, true, false, true>),
                             dim3(h->qg.grid), dim3(Q8Config::kBlock), s,
                             h->d_feats, h->d_q8, h->n, h->qg.iters, h->row_base, qa, qptr, exclude_global, topn,
                             h->d_block_lists, h->d_half_seed, q8_seeds, h->d_half_rescored, none, no_next,
                             no_cutoff, tail, epoch);
            } else {
                LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, false, false, true>),
                             dim3(h->qg.grid), dim3(Q8Config::kBlock), s,
                             h->d_feats, h->d_q8, h->n, h->qg.iters, h->row_base, qa, kNoQueryPtr, exclude_global, topn,
                             h->d_block_lists, h->d_half_seed, q8_seeds, h->d_half_rescored, none, no_next,
                             no_cutoff, tail, epoch);
            }
            HIP_TRY(h, hipGetLastError());
            // (the books move only once the launch is known to have been accepted: a refused launch leaves host and
            // device counters in step)
            for (unsigned g = 0; g < 8u; ++g) h->lone_base[g] += lone_tail_members(grid, g);
            h->lone_base[8] += lone_tail_groups(grid);
            *fused = true;
            return MI355REC_OK;
        }
        ++h->routes.q8;
        if (qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, true, false>),
                         dim3(h->qg.grid), dim3(Q8Config::kBlock), s,
                         h->d_feats, h->d_q8, h->n, h->qg.iters, h->row_base, qa, qptr, exclude_global, topn,
                         h->d_block_lists, h->d_half_seed, q8_seeds, h->d_half_rescored, none, no_next,
                         no_cutoff, no_tail, epoch);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, false, false>),
                         dim3(h->qg.grid), dim3(Q8Config::kBlock), s,
                         h->d_feats, h->d_q8, h->n, h->qg.iters, h->row_base, qa, kNoQueryPtr, exclude_global, topn,
                         h->d_block_lists, h->d_half_seed, q8_seeds, h->d_half_rescored, none, no_next,
                         no_cutoff, no_tail, epoch);
        }
        HIP_TRY(h, hipGetLastError());
        return MI355REC_OK;
    }
#ifdef MI355REC_EXPERIMENTS
    if (kind == kFp16) {
        ++h->half_scans;
        *n_lists = h->hg.grid;
        ++h->routes.fp16;
        uint32_t* const half_seed = reinterpret_cast<uint32_t*>(h->d_half_seed);   // (the fp16 scan's plain sample values)
        enqueue_half_seed(h, kFp16, qptr, qa, exclude_global, topn, h->d_half_seed, 0u, s);
        if (qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_kernel<HalfConfig, true, false>),
                         dim3(h->hg.grid), dim3(HalfConfig::kBlock), s,
                         h->d_feats, h->d_half, h->n, h->hg.iters, h->row_base, qa, qptr, exclude_global, topn,
                         h->d_block_lists, half_seed, h->hg.seed_grid * kHalfSeedWaves, h->d_half_rescored, none, no_next);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_kernel<HalfConfig, false, false>),
                         dim3(h->hg.grid), dim3(HalfConfig::kBlock), s,
                         h->d_feats, h->d_half, h->n, h->hg.iters, h->row_base, qa, kNoQueryPtr, exclude_global,
                         topn, h->d_block_lists, half_seed, h->hg.seed_grid * kHalfSeedWaves, h->d_half_rescored, none, no_next);
        }
        HIP_TRY(h, hipGetLastError());
        return MI355REC_OK;
    }
#endif
    *n_lists = h->grid;
    ++h->routes.fp32;
    // The launch-wide bound (kernels.hip.h): on shards where ~5 us are worth it, and never for the later rounds of
    // topn > 1024 (they look for keys BELOW the round before: a lower bound on the best keys says nothing there).
    const unsigned long long* bound = nullptr;
    const unsigned long long* sample = nullptr;
    uint32_t epoch = 0u;
    if (!upper_dev && h->n >= kF32LoneSeedMinRows) {
        epoch = next_epoch(h);
        if (enqueue_f32_seed(h, qptr, query12, exclude_global, topn, h->d_half_seed, h->d_lone_ctl, &h->lone_ctl_done, epoch, s))
            bound = &h->d_lone_ctl->cutoff;
        sample = h->d_half_seed;
    }
    if (qptr) {
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_kernel<ScanConfig, true, false>),
                     dim3(h->grid), dim3(kScanBlock), s,
                     h->d_feats, h->n, h->rows_per_block, h->iters, h->row_base, qa,
                     qptr, exclude_global, topn, h->d_block_lists,
                     static_cast<float*>(nullptr), upper_dev, none, bound, sample, epoch, no_next);
    } else {
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_kernel<ScanConfig, false, false>),
                     dim3(h->grid), dim3(kScanBlock), s,
                     h->d_feats, h->n, h->rows_per_block, h->iters, h->row_base, qa,
                     kNoQueryPtr, exclude_global, topn, h->d_block_lists,
                     static_cast<float*>(nullptr), upper_dev, none, bound, sample, epoch, no_next);
    }
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

int enqueue_merge(mi355rec* h, const uint64_t* lists, int n_lists, int list_len, int topn,
                  uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s, uint32_t notify = 0) {
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    if (notify) {   // the host polls h->h_done for this value (mi355rec_query_row_topn)
        hipLaunchKernelGGL(merge_notify_kernel, dim3(1), dim3(kMergeBlock), 0, s, lists, n_lists, list_len,
                           static_cast<int64_t>(list_len), topn, out_keys, out_idx, out_score, h->hd_done, notify);
    } else {
        hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(kMergeBlock), 0, s, lists, n_lists, list_len,
                           static_cast<int64_t>(list_len), static_cast<int64_t>(0), topn, out_keys, out_idx, out_score,
                           static_cast<int64_t>(0));
    }
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

// Multi-query passes for up to kMultiChain queries: ONE cheap seed (approximate
// scores of a spread ~2.6 % sample -> a chip-wide starting threshold per query),
// then per group of kMultiQueries the full pass (the catalogue is streamed once per
// group), then ONE merge launch with a workgroup per query.  topn <= kMultiMaxTopK.
int enqueue_multi(mi355rec* h, const float* queries, const int64_t* exclude, int count, int topn,
                  uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    MultiQueryArg qa[kMultiChain / kMultiQueries];
    const int groups = (count + kMultiQueries - 1) / kMultiQueries;
    for (int g = 0; g < groups; ++g) {
        std::memset(&qa[g], 0, sizeof qa[g]);
        for (int q = 0; q < kMultiQueries; ++q) {
            const int src = g * kMultiQueries + q;
            qa[g].exclude[q] = -1;
            if (src < count) {
                std::memcpy(qa[g].q[q], queries + static_cast<size_t>(src) * kDim, sizeof(float) * kDim);
                if (exclude) qa[g].exclude[q] = exclude[src];
            }
        }
    }
    const int64_t list_stride = static_cast<int64_t>(h->mgrid) * topn;
    const int seed_count = h->mgrid * kSeedWaves;
    const bool seeded = h->miters >= 3 && seed_count >= topn && seed_count <= kMergeBlock * kSeedSelectPerThread;
    if (seeded) {
        // one cheap launch for the whole chain: approximate scores of a spread
        // 2.6 % sample, then the per-query bound (kernels.hip.h, "seed")
        SeedQueryArg sq;
        std::memset(&sq, 0, sizeof sq);
        for (int q = 0; q < kMultiChain; ++q) sq.exclude[q] = -1;
        for (int q = 0; q < count; ++q) {
            std::memcpy(sq.q[q], queries + static_cast<size_t>(q) * kDim, sizeof(float) * kDim);
            if (exclude) sq.exclude[q] = exclude[q];
        }
        hipLaunchKernelGGL(seed_multi_kernel, dim3(h->mgrid), dim3(kSeedBlock), 0, s, h->d_feats, h->n,
                           h->mrows_per_block, h->row_base, sq, count, h->d_seed_vals);
        hipLaunchKernelGGL(seed_select_kernel, dim3(count), dim3(kMergeBlock), 0, s, h->d_seed_vals, seed_count,
                           topn, h->d_seed_keys);
    }
    for (int g = 0; g < groups; ++g) {
        const int nq = count - g * kMultiQueries < kMultiQueries ? count - g * kMultiQueries : kMultiQueries;
        ++h->routes.multi_fp32;
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_multi_kernel<MultiConfig>),
                     dim3(h->mgrid), dim3(MultiConfig::kBlock), s,
                     h->d_feats, h->n, static_cast<int64_t>(0), static_cast<int64_t>(0), h->miters, h->row_base,
                     qa[g], nq, g * kMultiQueries, topn, h->d_block_lists,
                     seeded ? h->d_seed_keys : static_cast<const uint64_t*>(nullptr));
    }
    HIP_TRY(h, hipGetLastError());
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(kMergeBlock), 0, s, h->d_block_lists, h->mgrid, topn,
                       static_cast<int64_t>(topn), list_stride, topn, out_keys, out_idx, out_score, static_cast<int64_t>(topn));
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

// Multi-query passes over the fp16 replica (replica_multi.hip.h): per group of up to kHmQueries
// queries ONE sample launch + ONE pass over the 24 B/row replica, then one merge launch with a
// workgroup per query for the whole chain.  queries[i] by value, or qptrs[i] != null: where its 12
// floats live in device-readable memory.  topn <= kMultiMaxTopK, count <= kMultiChain.
// Which replica a multi-query pass streams.  The 8-bit front end (12 B/row, integer matrix core, fp16 re-check of
// its candidates) moves half the bytes but its bound is 25x the fp16 one: ~1 % of the (row, query) pairs come
// back as candidates, 3.7 us per query of a pass against 0.85 us (measured, 10 M rows: 1 query 36.9 vs 44.1 us,
// 2: 41.9 vs 44.7, 12: 82 vs 53, 32: 152 vs 71).  So: passes of one or two queries, or when forced.
// Round 5: the front end is an A/B route of experiment builds (its one AUTO cell, passes of two queries, was worth
// 2.6 us per call and a second instantiation of the pass kernel to keep bit-identical).
bool multi_front_q8(const mi355rec* h, int nq) {
#ifdef MI355REC_EXPERIMENTS
    if (!use_q8(h) || h->batch_path == MI355REC_BATCH_HALF) return false;
    return h->batch_path == MI355REC_BATCH_Q8 || nq <= 2;
#else
    (void)h;
    (void)nq;
    return false;
#endif
}

bool half_multi_ok(const mi355rec* h, int topn) {
    return h->d_half && h->replica_mode != MI355REC_REPLICA_OFF && topn <= kMultiMaxTopK && h->hg.seed_grid > 0 &&
           h->hg.seed_grid * kHalfSeedWaves >= topn;
}

void fill_half_multi_arg(HalfMultiArg& arg, float margin, const float* queries, const float* const* qptrs, const int64_t* exclude,
                         int g0, int nq) {
    std::memset(&arg, 0, sizeof arg);
    arg.margin = margin;
    for (int q = 0; q < kHmQueries; ++q) {
        arg.exclude[q] = -1;
        if (q >= nq) continue;
        if (qptrs && qptrs[g0 + q]) {
            hm_set_pointer(arg, q, qptrs[g0 + q]);
        } else if (queries) {
            std::memcpy(arg.q[q], queries + static_cast<size_t>(g0 + q) * kDim, sizeof(float) * kDim);
        }
        if (exclude) arg.exclude[q] = exclude[g0 + q];
    }
}

// How much of the shard a batch of nq queries samples for its cutoffs (replica_multi.hip.h, hm_sample_regions):
// regions of 1024 << l rows.  The sample is paid once per batch, the candidates its cutoff lets through once per
// query: 2.6 % of 10 M rows leave ~5 100 candidates per query, 5 % ~2 700, 10 % ~1 400.  A sample launch of its own
// is over in a few us whatever it reads; seed riders (`riding`) share the memory system with the pass they ride in,
// row for row, so a streamed batch samples at most 5 % (measured, tools/hm_riders.sh, 10 M rows x 12 queries: launches
// of 45.1 / 43.6 / 45.1 us at 2.6 / 5 / 10 %; x 32 queries: 52.3 / 49.7 / 50.2).  Regions must not overlap.
int hm_sample_log2(const mi355rec* h, int nq, bool riding) {
    int l = nq >= 12 ? 2 : nq >= 5 ? 1 : 0;
    if (riding && l > 1) l = 1;
    MI355REC_EXP_INT(l, "MI355REC_EXP_SAMPLE_LOG2", 0, 3);
    while (l > 0 && (static_cast<int64_t>(1024) << l) > h->hg.seed_stride) --l;
    return l;
}

int enqueue_half_multi(mi355rec* h, const float* queries, const float* const* qptrs, const int64_t* exclude, int count,
                       int topn, uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    const int n_seed = h->hg.seed_grid * kHalfSeedWaves;
    HmRide no_ride;
    std::memset(&no_ride, 0, sizeof no_ride);
    HalfMultiArg arg;
    for (int g0 = 0; g0 < count; g0 += kHmQueries) {
        const int nq = count - g0 < kHmQueries ? count - g0 : kHmQueries;
        fill_half_multi_arg(arg, h->margin_mfma, queries, qptrs, exclude, g0, nq);
        const uint32_t epoch = next_epoch(h);
        // the sample launch's last workgroup selects the cutoffs; the pass reads them (stream order)
        const unsigned long long* const cuts = h->d_half_mcuts;
        // (+ one workgroup per query for its neighbourhood's bound: handoff.hip.h)
        hipLaunchKernelGGL(seed_half_multi_kernel, dim3(h->hg.seed_grid + nq), dim3(kHmBlock), 0, s, h->d_feats, h->d_half, h->n, h->row_base,
                           h->hg.seed_stride, arg, nq, h->hg.seed_grid, h->d_half_mseed, epoch, hm_sample_log2(h, nq, false), h->d_half_mctl,
                           h->half_mctl_done + (h->dbg_no_last ? 0x40000000u : 0u), h->d_half_mcuts, topn, h->dbg_skip_regions);
        HIP_TRY(h, hipGetLastError());
        h->half_mctl_done += static_cast<unsigned>(h->hg.seed_grid);
        h->dbg_no_last = false;   // (test hooks of mi355rec_debug_handoff: they apply to ONE sampling launch)
        h->dbg_skip_regions = 0;
        ++h->half_scans;
#ifdef MI355REC_EXPERIMENTS
        if (multi_front_q8(h, nq)) {   // rows from the 8-bit replica through the integer matrix core (replica_multi.hip.h)
            ++h->q8_scans;
            ++h->routes.multi_q8;
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_multi_kernel<false, true>), dim3(h->hg.grid),
                         dim3(kHmBlock), s, h->d_feats, h->d_half, reinterpret_cast<const uint32_t*>(h->d_q8), h->n, h->row_base, arg, nq,
                         g0, topn, h->d_block_lists, h->d_half_mseed, n_seed, h->d_half_rescored, no_ride, arg, cuts, epoch);
        } else
#endif
        {
            ++h->routes.multi_fp16;
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_multi_kernel<false, false>), dim3(h->hg.grid),
                         dim3(kHmBlock), s, h->d_feats, h->d_half, static_cast<const uint32_t*>(nullptr), h->n, h->row_base, arg, nq,
                         g0, topn, h->d_block_lists, h->d_half_mseed, n_seed, h->d_half_rescored, no_ride, arg, cuts, epoch);
        }
    }
    HIP_TRY(h, hipGetLastError());
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    hipLaunchKernelGGL(merge_kernel, dim3(count), dim3(kMergeBlock), 0, s, h->d_block_lists, h->hg.grid, topn,
                       static_cast<int64_t>(topn), static_cast<int64_t>(h->hg.grid) * topn, topn, out_keys, out_idx, out_score,
                       static_cast<int64_t>(topn));
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

// ---- a STREAM of batches over the replica (mi355rec_enqueue_batch_keys_streamed) ---------------------
// The single-query stream's scheme (enqueue_streamed), one level up: the stream runs one call behind —
// call k + 1 LAUNCHES batch k — and that launch carries, beside its scanners, a merging workgroup per three
// queries of batch k - 1 and a few seed riders that take the sample of batch k + 1.  A stream of K batches
// costs K launches + one sample launch at its head + one merge launch at its tail (the flush).
// Workgroups of a streamed launch that do not scan take a scanner's place among the resident ones (measured at 10 M rows:
// 32 mergers + 64 riders of 512 made a 41 us pass 57 us), so they are as few as can still finish inside the pass:
constexpr int kHmRiders = 16;       // seed riders per 1024 rows of a sampled region: a rider's wave gets through a 128-row
                                    // chunk every ~2 us beside a pass (as a scanner's does), 16 (or 32) of them take 33 us
constexpr int kHmMergesPerWg = 3;   // queries of the previous batch one merging workgroup takes, one after the other (~10 us each)
constexpr int kHmNbhdPerWg = 4;     // queries of the next batch one neighbourhood workgroup takes, one after the other (~4 us each)

int ensure_mstream(mi355rec* h) {
    if (h->mstream_ready) return MI355REC_OK;
    const size_t list_bytes = sizeof(uint64_t) * static_cast<size_t>(kHmQueries) * h->hg.grid * kMultiMaxTopK;
    const size_t seed_bytes = sizeof(unsigned long long) * static_cast<size_t>(kHmSampleSlots);
    hipError_t e = hipSuccess;
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc(&h->d_mstream_lists[i], list_bytes);
        if (e == hipSuccess) e = hipMalloc(&h->d_mstream_seed[i], seed_bytes);
    }
    if (e == hipSuccess) e = hipMalloc(&h->d_mstream_cuts, sizeof(unsigned long long) * 2 * kHmQueries);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_mstream_cuts, 0, sizeof(unsigned long long) * 2 * kHmQueries, h->stream);
    if (e == hipSuccess) e = hipMalloc(&h->d_mstream_ctl, sizeof(SeedCtl) * 2);
    if (e == hipSuccess) e = hipMemsetAsync(h->d_mstream_ctl, 0, sizeof(SeedCtl) * 2, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {   // all or nothing
        if (h->d_mstream_cuts) (void)hipFree(h->d_mstream_cuts);
        if (h->d_mstream_ctl) (void)hipFree(h->d_mstream_ctl);
        h->d_mstream_cuts = nullptr;
        h->d_mstream_ctl = nullptr;
        for (int i = 0; i < 2; ++i) {
            if (h->d_mstream_lists[i]) (void)hipFree(h->d_mstream_lists[i]);
            if (h->d_mstream_seed[i]) (void)hipFree(h->d_mstream_seed[i]);
            h->d_mstream_lists[i] = nullptr;
            h->d_mstream_seed[i] = nullptr;
        }
        return fail(h, e == hipErrorOutOfMemory ? MI355REC_ERR_OUT_OF_MEMORY : MI355REC_ERR_HIP, "hipMalloc(batch stream): %s",
                    hipGetErrorString(e));
    }
    h->mctl_done[0] = h->mctl_done[1] = 0u;
    h->mstream_ready = true;
    return MI355REC_OK;
}

// Launches the stashed batch: scanners + the mergers of the batch before it + (next != null) the seed
// riders of the batch after it.
int launch_mstash(mi355rec* h, hipStream_t s, const HalfMultiArg* next, int next_nq, int next_topn, int next_buf, uint32_t next_epoch_tag) {
    auto& st = h->mstash;
    const int buf = h->mpending.has ? 1 - h->mpending.buf : 0;
    HmRide ride;
    std::memset(&ride, 0, sizeof ride);
    if (h->mpending.has) {
        ride.prev_lists = h->d_mstream_lists[h->mpending.buf];
        ride.prev_out = h->mpending.out;
        ride.prev_queries = h->mpending.nq;
        ride.merge_wgs = (h->mpending.nq + kHmMergesPerWg - 1) / kHmMergesPerWg;
        ride.prev_n_lists = h->mpending.n_lists;
        ride.prev_topk = h->mpending.topn;
    }
    if (next) {
        ride.sample_log2 = hm_sample_log2(h, next_nq, true);
        ride.seed_wgs = kHmRiders << ride.sample_log2;
        MI355REC_EXP_INT(ride.seed_wgs, "MI355REC_EXP_RIDERS", 1, 512);
        if (ride.seed_wgs > h->hg.seed_grid) ride.seed_wgs = h->hg.seed_grid;
        ride.nb_wgs = (next_nq + kHmNbhdPerWg - 1) / kHmNbhdPerWg;
        ride.next_queries = next_nq;
        ride.regions = h->hg.seed_grid;
        ride.stride_rows = h->hg.seed_stride;
        ride.next_seed_vals = h->d_mstream_seed[next_buf];
        ride.next_ctl = h->d_mstream_ctl + next_buf;
        ride.next_cuts = h->d_mstream_cuts + next_buf * kHmQueries;
        ride.next_topk = next_topn;
        ride.next_epoch = next_epoch_tag;
        // the riders' arrival counter counts up and is never reset: this launch's riders start from ...
        ride.done_base = h->mctl_done[next_buf] + (h->dbg_no_last ? 0x40000000u : 0u);
        ride.debug_skip = h->dbg_skip_regions;
        h->dbg_no_last = false;
        h->dbg_skip_regions = 0;
    }
    const unsigned long long* cuts_ready = st.cuts_ready ? h->d_mstream_cuts + st.seed_buf * kHmQueries : nullptr;
    // the launch stays within one resident wave of workgroups: the riders and mergers take scanner slots
    const int others = ride.merge_wgs + ride.seed_wgs + ride.nb_wgs;
    int scanners = h->hg.grid - others;
    if (scanners < 1) scanners = 1;
    ++h->half_scans;
#ifdef MI355REC_EXPERIMENTS
    if (multi_front_q8(h, st.nq)) {
        ++h->q8_scans;
        ++h->routes.multi_q8;
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_multi_kernel<true, true>),
                     dim3(scanners + others), dim3(kHmBlock), s, h->d_feats, h->d_half,
                     reinterpret_cast<const uint32_t*>(h->d_q8), h->n, h->row_base,
                     st.arg, st.nq, 0, st.topn, h->d_mstream_lists[buf], h->d_mstream_seed[st.seed_buf],
                     h->hg.seed_grid * kHalfSeedWaves, h->d_half_rescored, ride, next ? *next : st.arg, cuts_ready, st.epoch);
    } else
#endif
    {
        ++h->routes.multi_fp16;
        LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_multi_kernel<true, false>),
                     dim3(scanners + others), dim3(kHmBlock), s, h->d_feats, h->d_half,
                     static_cast<const uint32_t*>(nullptr), h->n, h->row_base,
                     st.arg, st.nq, 0, st.topn, h->d_mstream_lists[buf], h->d_mstream_seed[st.seed_buf],
                     h->hg.seed_grid * kHalfSeedWaves, h->d_half_rescored, ride, next ? *next : st.arg, cuts_ready, st.epoch);
    }
    HIP_TRY(h, hipGetLastError());
    if (next) h->mctl_done[next_buf] += static_cast<unsigned>(ride.seed_wgs);   // (the books move once the launch has been accepted)
    h->mpending.has = true;
    h->mpending.buf = buf;
    h->mpending.nq = st.nq;
    h->mpending.topn = st.topn;
    h->mpending.n_lists = scanners;
    h->mpending.out = st.out;
    st.has = false;
    return MI355REC_OK;
}

int flush_mstream(mi355rec* h, hipStream_t s) {
    if (h->mstash.has) {
        const int rc = launch_mstash(h, s, nullptr, 0, 0, 0, 0u);
        if (rc) return rc;
    }
    if (!h->mpending.has) return MI355REC_OK;
    const auto& p = h->mpending;
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    hipLaunchKernelGGL(merge_kernel, dim3(p.nq), dim3(kMergeBlock), 0, s, h->d_mstream_lists[p.buf], p.n_lists, p.topn,
                       static_cast<int64_t>(p.topn), static_cast<int64_t>(p.n_lists) * p.topn, p.topn, p.out,
                       static_cast<int64_t*>(nullptr), static_cast<float*>(nullptr), static_cast<int64_t>(p.topn));
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    h->mpending.has = false;
    return MI355REC_OK;
}

// One batch of <= kHmQueries queries joins the stream.
int enqueue_mstream(mi355rec* h, const float* queries, const float* const* qptrs, const int64_t* exclude, int g0, int nq, int topn,
                    uint64_t* out_keys, hipStream_t s) {
    int rc = ensure_mstream(h);
    if (rc) return rc;
    HalfMultiArg arg;
    fill_half_multi_arg(arg, h->margin_mfma, queries, qptrs, exclude, g0, nq);
    int seed_buf = 0;
    bool cuts_ready = false;
    const uint32_t epoch = next_epoch(h);   // the tag of this batch's sample values and cutoffs
    if (h->mstash.has) {
        seed_buf = 1 - h->mstash.seed_buf;
        rc = launch_mstash(h, s, &arg, nq, topn, seed_buf, epoch);   // its riders take THIS batch's sample (and select its cutoffs)
        if (rc) return rc;
        cuts_ready = h->hg.seed_grid > 0;   // (launch_mstash gave the launch seed riders)
    } else {   // the head of a stream: a sample launch of its own
        hipLaunchKernelGGL(seed_half_multi_kernel, dim3(h->hg.seed_grid + nq), dim3(kHmBlock), 0, s, h->d_feats, h->d_half, h->n, h->row_base,
                           h->hg.seed_stride, arg, nq, h->hg.seed_grid, h->d_mstream_seed[seed_buf], epoch, hm_sample_log2(h, nq, false),
                           h->d_mstream_ctl + seed_buf, h->mctl_done[seed_buf] + (h->dbg_no_last ? 0x40000000u : 0u),
                           h->d_mstream_cuts + seed_buf * kHmQueries, topn, h->dbg_skip_regions);
        HIP_TRY(h, hipGetLastError());
        h->mctl_done[seed_buf] += static_cast<unsigned>(h->hg.seed_grid);
        h->dbg_no_last = false;
        h->dbg_skip_regions = 0;
        cuts_ready = h->hg.seed_grid > 0;
    }
    auto& st = h->mstash;
    st.has = true;
    st.arg = arg;
    st.nq = nq;
    st.topn = topn;
    st.out = out_keys;
    st.seed_buf = seed_buf;
    st.epoch = epoch;
    st.cuts_ready = cuts_ready;
    return MI355REC_OK;
}

int check_topn(mi355rec* h, int topn, bool allow_rounds) {
    if (topn <= 0)
        return fail(h, MI355REC_ERR_INVALID_ARG, "topn must be positive, got %d", topn);
    if (!allow_rounds && topn > kMaxTopK)
        return fail(h, MI355REC_ERR_INVALID_ARG, "topn %d > %d is not supported by this call", topn, kMaxTopK);
    return MI355REC_OK;
}

// One query end to end on stream `s`: scan + merge, in rounds of kMaxTopK when
// topn is larger (round r only sees keys below the last key of round r-1, read
// from device memory, so the rounds are enqueued back to back without a sync).
int enqueue_query(mi355rec* h, const float* qptr, const float* query12, int64_t exclude_global,
                  int topn_asked, uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s, uint32_t notify = 0) {
    // A shard of n rows has at most n results (the reference's heap never grows
    // past N-1, Recommender.cu:300): run only the rounds that can produce keys and
    // pad the rest, so an absurd topn costs a memset, not topn/1024 catalogue scans.
    const int topn = static_cast<int64_t>(topn_asked) < h->n ? topn_asked : static_cast<int>(h->n);
    if (topn < topn_asked) {
        const size_t pad = static_cast<size_t>(topn_asked - topn);
        HIP_TRY(h, hipMemsetAsync(out_keys + topn, 0, pad * sizeof(uint64_t), s));
        if (out_idx) HIP_TRY(h, hipMemsetAsync(out_idx + topn, 0xff, pad * sizeof(int64_t), s));
        if (out_score) HIP_TRY(h, hipMemsetAsync(out_score + topn, 0, pad * sizeof(float), s));
    }
    for (int done = 0; done < topn; done += kMaxTopK) {
        const int k = topn - done < kMaxTopK ? topn - done : kMaxTopK;
        const uint64_t* upper = done ? out_keys + done - 1 : nullptr;
        int lists = 0;
        // a notifying query (single round, its caller polls the completion word): scan, merge and the word in ONE launch
        LoneTail lone{h->d_lone_ctr, out_keys, out_idx, out_score, h->hd_done, notify, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};   // (bases: enqueue_scan)
        bool fused = false;
        int rc = enqueue_scan(h, qptr, query12, exclude_global, k, upper, s, &lists, (notify && h->d_lone_ctr) ? &lone : nullptr, &fused);
        if (rc) return rc;
        if (fused) {
            ++h->lone_fused;
            continue;
        }
        // (a notifying merge is only asked for single-round queries: it is the last launch of the call)
        rc = enqueue_merge(h, h->d_block_lists, lists, k, k, out_keys + done,
                           out_idx ? out_idx + done : nullptr, out_score ? out_score + done : nullptr, s, notify);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// Waits for the completion word of a notifying merge (a relaxed spin on pinned host memory); the stream
// is asked now and then so that a failed launch cannot hang the caller.
int wait_done(mi355rec* h, uint32_t want) {
    for (uint64_t spins = 1;; ++spins) {
        if (__atomic_load_n(h->h_done, __ATOMIC_ACQUIRE) == want) return MI355REC_OK;
        __builtin_ia32_pause();
        if ((spins & 0x3ffff) == 0) {   // every ~1 ms
            const hipError_t e = hipStreamQuery(h->stream);
            if (e == hipSuccess) {
                if (__atomic_load_n(h->h_done, __ATOMIC_ACQUIRE) == want) return MI355REC_OK;
                return fail(h, MI355REC_ERR_HIP, "the query's stream drained without its completion word");
            }
            if (e != hipErrorNotReady) return fail(h, MI355REC_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(e));
        }
    }
}

// ---- streamed single queries -------------------------------------------------------
// A stream of single queries runs ONE CALL BEHIND: query k is launched by call k + 1 (or by the flush), and its launch
// carries, beside the scanners, the merger of query k - 1's lists (one workgroup) and — where the launch can spare
// them — the seed riders and the neighbourhood workgroup of query k + 1 (handoff.hip.h), so that every launch starts
// from a launch-wide bound without a sample launch of its own.  That holds for all three kinds of rows a scan can
// stream (fp32, 8-bit replica; fp16 replica in experiment builds).  One scanning workgroup fewer than the plain scan
// uses per non-scanning one, so the launch still fits the chip in one wave of workgroups.
int ensure_streamed_alloc(mi355rec* h);
int ensure_streamed(mi355rec* h) {
    if (h->streamed_ready) return MI355REC_OK;
    const int rc = ensure_streamed_alloc(h);
    if (rc != MI355REC_OK) {   // all or nothing: no half-allocated state survives a failure
        for (int i = 0; i < 2; ++i) {
            if (h->d_stream_lists[i]) (void)hipFree(h->d_stream_lists[i]);
            h->d_stream_lists[i] = nullptr;
        }
        return rc;
    }
    h->streamed_ready = true;
    return MI355REC_OK;
}

int ensure_streamed_alloc(mi355rec* h) {
    int most = h->sgrid > h->hg.sgrid ? h->sgrid : h->hg.sgrid;
    if (h->qg.sgrid > most) most = h->qg.sgrid;
    for (int i = 0; i < 2; ++i)
        HIP_TRY(h, hipMalloc(&h->d_stream_lists[i], sizeof(uint64_t) * static_cast<size_t>(most) * kMaxTopK));
    return MI355REC_OK;
}

int launch_stashed(mi355rec* h, hipStream_t s, bool with_next, const float* next_ptr, const float* next_q,
                   int64_t next_exclude, int next_topn, int next_buf, uint32_t next_epoch_tag);

int flush_streamed(mi355rec* h, hipStream_t s) {
    if (h->stashed.has) {
        const int rc = launch_stashed(h, s, false, nullptr, nullptr, -1, 0, 0, 0u);
        if (rc) return rc;
    }
    if (!h->pending) return MI355REC_OK;
    h->pending = false;
    return enqueue_merge(h, h->d_stream_lists[h->pending_buf], h->pending_lists, h->pending_topn, h->pending_topn,
                         h->pending_out, nullptr, nullptr, s);
}

// How many seed riders a streamed launch over `kind` rows carries for the NEXT query, and whether their last one
// leaves that query's bound (cutoff) in d_stream_ctl.
int stream_riders(const mi355rec* h, int kind) { return kind == kFp32 ? h->fg.riders : (kind == kQ8 ? h->qg.riders : h->hg.riders); }
bool stream_hoists(const mi355rec* h, int kind) {
    return kind == kFp32 ? h->fg.riders > 0 : (kind == kQ8 ? q8_hoists(h) : false);
}
// ... and whether the launch has a workgroup for the next query's neighbourhood at all.
bool stream_nbhd(const mi355rec* h, int kind) { return kind == kFp32 ? h->fg.nbhd != 0 : (kind == kQ8 ? h->qg.riders > 0 : false); }

// Launches the stashed streamed query: scanners + the riding merger of the query before it + (with_next) the seed
// riders and the neighbourhood workgroup of the query after it.
int launch_stashed(mi355rec* h, hipStream_t s, bool with_next, const float* next_ptr, const float* next_q,
                   int64_t next_exclude, int next_topn, int next_buf, uint32_t next_epoch_tag) {
    auto& st = h->stashed;
    // The fp32 scan's riding merger keeps 2048 survivors; with ~770 lists and topN near 1000 about
    // 2.2 topN keys survive its first cut, and an overflow drops into the exact radix select over all
    // keys in global memory (correct, ~1 ms).  Such a query's merge gets its own launch instead.
    if (st.kind == kFp32 && h->pending && h->pending_topn > kRideTopnMax) {
        h->pending = false;
        const int rc = enqueue_merge(h, h->d_stream_lists[h->pending_buf], h->pending_lists, h->pending_topn, h->pending_topn,
                                     h->pending_out, nullptr, nullptr, s);
        if (rc) return rc;
    }
    const int buf = h->pending ? 1 - h->pending_buf : 0;
    PrevMerge prev{nullptr, 0, 0, nullptr};
    if (h->pending) prev = PrevMerge{h->d_stream_lists[h->pending_buf], h->pending_lists, h->pending_topn, h->pending_out};
    NextSeed next;
    std::memset(&next, 0, sizeof next);
    next.query_ptr = nullptr;
    next.exclude_global = -1;
    int scanners, iters;
    if (st.kind == kFp32) {
        scanners = h->sgrid;
        iters = h->siters;
    } else {
        const ReplicaGeom& g = st.kind == kQ8 ? h->qg : h->hg;
        scanners = g.sgrid;
        iters = g.siters;
    }
    unsigned riders_arriving = 0u;
    if (with_next && (stream_riders(h, st.kind) > 0 || stream_nbhd(h, st.kind))) {
        next.query_ptr = next_ptr;
        if (!next_ptr) std::memcpy(next.q, next_q, sizeof next.q);
        next.exclude_global = next_exclude;
        next.out = h->d_stream_seed[next_buf];
        next.n_wgs = stream_riders(h, st.kind);
        next.nbhd = stream_nbhd(h, st.kind) ? 1 : 0;   // (it stores its slot even when the excluded row is not of this shard)
        if (st.kind == kFp32) {
            next.regions = h->fg.seed_grid;
            next.stride_rows = h->fg.seed_stride;
            scanners = h->fg.r_scan;
            iters = h->fg.r_iters;
        } else {
            const ReplicaGeom& g = st.kind == kQ8 ? h->qg : h->hg;
            next.regions = g.seed_grid;
            next.stride_rows = g.seed_stride;
            scanners = g.r_scan;
            iters = g.r_iters;
        }
        next.ctl = (next.n_wgs > 0 && stream_hoists(h, st.kind)) ? h->d_stream_ctl + next_buf : nullptr;
        next.topk = next_topn;
        next.exact = st.kind == kQ8 && q8_exact_sample(h);
        next.epoch = next_epoch_tag;
        if (next.ctl) {   // the riders' arrival counter counts up and is never reset: this launch's riders start from ...
            next.done_base = h->ctl_done[next_buf] + (h->dbg_no_last ? 0x40000000u : 0u);
            riders_arriving = static_cast<unsigned>(next.n_wgs);
        }
        next.debug_skip = h->dbg_skip_regions;
        h->dbg_no_last = false;
        h->dbg_skip_regions = 0;
    }
    QueryArg qa;
    std::memset(&qa, 0, sizeof qa);
    qa.margin = h->margin_mix;
    if (!st.qptr) std::memcpy(qa.q, st.q, sizeof qa.q);
    const dim3 grid(static_cast<unsigned>(scanners + 1 + next.n_wgs + next.nbhd));
    unsigned long long* const my_seed = h->d_stream_seed[st.seed_buf];
    const unsigned long long* ready = st.cutoff_ready ? &h->d_stream_ctl[st.seed_buf].cutoff : nullptr;
    if (st.kind == kQ8) {
        ++h->half_scans;
        ++h->q8_scans;
        ++h->routes.q8;
        const int n_seed = h->qg.seed_grid * kHalfSeedWaves;
        const int q8_seeds = q8_exact_sample(h) ? -n_seed : n_seed;   // (negative: exact values)
        const LoneTail no_tail{nullptr, nullptr, nullptr, nullptr, nullptr, 0u, {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}};
        if (st.qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, true, true>),
                         grid, dim3(Q8Config::kBlock), s,
                         h->d_feats, h->d_q8, h->n, iters, h->row_base, qa, st.qptr, st.exclude, st.topn,
                         h->d_stream_lists[buf], my_seed, q8_seeds, h->d_half_rescored, prev, next, ready, no_tail, st.epoch);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_q8_kernel<Q8Config, false, true>),
                         grid, dim3(Q8Config::kBlock), s,
                         h->d_feats, h->d_q8, h->n, iters, h->row_base, qa, kNoQueryPtr, st.exclude, st.topn,
                         h->d_stream_lists[buf], my_seed, q8_seeds, h->d_half_rescored, prev, next, ready, no_tail, st.epoch);
        }
#ifdef MI355REC_EXPERIMENTS
    } else if (st.kind == kFp16) {
        ++h->half_scans;
        ++h->routes.fp16;
        const int n_seed = h->hg.seed_grid * kHalfSeedWaves;
        if (st.qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_kernel<HalfConfig, true, true>),
                         grid, dim3(HalfConfig::kBlock), s,
                         h->d_feats, h->d_half, h->n, iters, h->row_base, qa, st.qptr, st.exclude, st.topn,
                         h->d_stream_lists[buf], reinterpret_cast<uint32_t*>(my_seed), n_seed, h->d_half_rescored, prev, next);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_half_kernel<HalfConfig, false, true>),
                         grid, dim3(HalfConfig::kBlock), s,
                         h->d_feats, h->d_half, h->n, iters, h->row_base, qa, kNoQueryPtr, st.exclude, st.topn,
                         h->d_stream_lists[buf], reinterpret_cast<uint32_t*>(my_seed), n_seed, h->d_half_rescored, prev, next);
        }
#endif
    } else {
        ++h->routes.fp32;
        if (st.qptr) {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_kernel<ScanConfig, true, false, 0, true>),
                         grid, dim3(kScanBlock), s,
                         h->d_feats, h->n, static_cast<int64_t>(0), iters, h->row_base, qa, st.qptr,
                         st.exclude, st.topn, h->d_stream_lists[buf], static_cast<float*>(nullptr),
                         static_cast<const uint64_t*>(nullptr), prev, ready, my_seed, st.epoch, next);
        } else {
            LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, (scan_kernel<ScanConfig, false, false, 0, true>),
                         grid, dim3(kScanBlock), s,
                         h->d_feats, h->n, static_cast<int64_t>(0), iters, h->row_base, qa, kNoQueryPtr,
                         st.exclude, st.topn, h->d_stream_lists[buf], static_cast<float*>(nullptr),
                         static_cast<const uint64_t*>(nullptr), prev, ready, my_seed, st.epoch, next);
        }
    }
    HIP_TRY(h, hipGetLastError());
    // (the books move only once the launch is known to have been accepted)
    if (riders_arriving) h->ctl_done[next_buf] += riders_arriving;
    h->pending = true;
    h->pending_buf = buf;
    h->pending_topn = st.topn;
    h->pending_out = st.out;
    h->pending_lists = scanners;
    st.has = false;
    return MI355REC_OK;
}

int flush_mstream(mi355rec* h, hipStream_t s);

int enqueue_streamed(mi355rec* h, const float* qptr, const float* query12, int64_t exclude_global, int topn,
                     uint64_t* out_keys, hipStream_t s) {
    int rc = ensure_streamed(h);
    if (rc) return rc;
    rc = flush_mstream(h, s);   // a stream of BATCHES on this handle is closed first
    if (rc) return rc;
    if (MI355REC_EXP_FLAG("MI355REC_EXP_RIDE_NOMERGE") && h->pending) {
        rc = flush_streamed(h, s);
        if (rc) return rc;
    }
    // One call behind: the query of the PREVIOUS call is launched now, and its launch takes the sample and the
    // neighbourhood of this one.  The first query of a stream needs a sample launch of its own.
    const int kind = single_kind(h, nullptr);
    int seed_buf = 0;
    bool sampled = false, nbhd_taken = false;
    const uint32_t epoch = next_epoch(h);   // the tag of this query's sample values and bound
    if (h->stashed.has) {
        seed_buf = 1 - h->stashed.seed_buf;
        // the riders of a launch sample the rows that launch scans: a change of rows (mi355rec_set_replica) between two
        // calls costs the next query a sample launch of its own
        const bool same = h->stashed.kind == kind;
        sampled = same && stream_riders(h, kind) > 0;
        nbhd_taken = same && stream_nbhd(h, kind);
        rc = launch_stashed(h, s, same, qptr, query12, exclude_global, topn, seed_buf, epoch);
        if (rc) return rc;
    }
    bool bound_ready = sampled && stream_hoists(h, kind);
    if (!sampled) {   // first query of a stream, or a shard too small to spare riders
        if (kind == kFp32) {
            if (h->n >= kF32LoneSeedMinRows)
                bound_ready = enqueue_f32_seed(h, qptr, query12, exclude_global, topn, h->d_stream_seed[seed_buf], h->d_stream_ctl + seed_buf,
                                               &h->ctl_done[seed_buf], epoch, s);
        } else if (!nbhd_taken) {
            QueryArg qa;
            std::memset(&qa, 0, sizeof qa);
            qa.margin = h->margin_mix;
            if (!qptr) std::memcpy(qa.q, query12, sizeof qa.q);
            enqueue_half_seed(h, kind, qptr, qa, exclude_global, topn, h->d_stream_seed[seed_buf], epoch, s);
        }
        HIP_TRY(h, hipGetLastError());
    }
    auto& st = h->stashed;
    st.has = true;
    st.qptr = qptr;
    if (!qptr) std::memcpy(st.q, query12, sizeof st.q);
    st.exclude = exclude_global;
    st.topn = topn;
    st.out = out_keys;
    st.seed_buf = seed_buf;
    st.epoch = epoch;
    st.kind = kind;
    st.cutoff_ready = bound_ready;
    return MI355REC_OK;
}

// ---- batched path (batched.hip.h) -----------------------------------------------

constexpr int64_t kBqMinRows = 65536;   // below this the launch count, not the arithmetic, decides
constexpr int kBqMinBatch = 13;          // without a replica: up to 12 queries are ONE exact multi-query pass (141 us at 10 M rows)
constexpr int kHmAutoMax = 32;           // up to here a batch goes in ONE multi-query pass over the replica (measured at 10 M
                                         // rows x top-100, round 4: 70 / 76 / 78 / 81 us per call for 2 / 12 / 16 / 32 queries, the
                                         // matrix-core path 88-93 for any chunk of <= 32)
constexpr int kBqMinBatchReplica = 3;    // with one, the passes cost ~92 us for any chunk of <= 32 queries (two single
                                         // replica scans cost 88): measured at 10 M rows, tools/run_batched.py

void free_bq(mi355rec* h);

int ensure_bq_alloc(mi355rec* h);

// Pass 1 looks at every step-th 64-row tile (a threshold from ANY subset of the rows is valid): 4 once each wave still
// gets a couple of dozen tiles, less on small shards.  A power of two.
int bq_step1(const mi355rec* h, int64_t n_tiles) {
    const auto& b = h->bq;
    int step1 = n_tiles >= static_cast<int64_t>(b.grid) * (kBqPassBlock / 64) * 16 ? b.step1 : 1;
    while (step1 > 1 && n_tiles < static_cast<int64_t>(b.grid) * (kBqPassBlock / 64) * 8 * step1) step1 /= 2;
    return step1;
}

// First batched call on a handle: allocate the path's scratch (all or nothing).
int ensure_bq(mi355rec* h) {
    if (h->bq.ready) return MI355REC_OK;
    const int rc = ensure_bq_alloc(h);
    if (rc != MI355REC_OK) free_bq(h);   // no half-allocated state survives a failure
    return rc;
}

int ensure_bq_alloc(mi355rec* h) {
    auto& b = h->bq;
    // workgroups of a pass: what the 1024-query kernels can keep resident (LDS: 32 KiB of B
    // fragments per workgroup; registers: 4 resp. 5 waves per SIMD), the same for both passes
    int occ1 = 0, occ2 = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, bq_pass_kernel<kBqMaxBlocks, false>, kBqPassBlock, 0) != hipSuccess || occ1 < 1) occ1 = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ2, bq_pass_kernel<kBqMaxBlocks, true>, kBqPassBlock, 0) != hipSuccess || occ2 < 1) occ2 = 1;
    b.occ1 = occ1 < 5 ? occ1 : 5;
    b.occ2 = occ2 < 5 ? occ2 : 5;
    int grid = h->cus * b.occ1;
    if (grid > kBqMaxPassGrid) grid = kBqMaxPassGrid;
    b.grid = grid;
    b.grid2 = h->cus * b.occ2;
    {
        int v = b.step1;
        MI355REC_EXP_INT(v, "MI355REC_BQ_STEP1", 1, 8);
        if (v == 1 || v == 2 || v == 4 || v == 8) b.step1 = v;
    }
    b.qgrid = h->cus < 1024 ? h->cus : 1024;   // (the queued scan's last workgroup merges up to 1024 lists per query)
    const int64_t tiles = (h->n + MultiConfig::kTileRows - 1) / MultiConfig::kTileRows;
    if (tiles < b.qgrid) b.qgrid = static_cast<int>(tiles);
    b.qiters = static_cast<int>((tiles + b.qgrid - 1) / b.qgrid);
    HIP_TRY(h, hipMalloc(&b.bfrag, sizeof(uint32_t) * kBqMaxBlocks * 64 * 4));
    HIP_TRY(h, hipMalloc(&b.qnorm, sizeof(float) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.qthr, sizeof(float) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.qflags, sizeof(uint32_t) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.cand_count, sizeof(int) * kBqMaxQueries * kBqCountStride));
    HIP_TRY(h, hipMemsetAsync(b.cand_count, 0, sizeof(int) * kBqMaxQueries * kBqCountStride, h->stream));
    // A query keeps about rows / 64 candidates at most (a power of two in [2048, 65536]): uniform rows need ~650 at 10 M,
    // rows that cluster a whole cluster's worth (profiles/r04_clustered.jsonl); past it the query goes to the exact queue.
    b.cand_cap = kBqCapMin;
    while (b.cand_cap < kBqCapMax && static_cast<int64_t>(b.cand_cap) * 64 < h->n) b.cand_cap *= 2;
    HIP_TRY(h, hipMalloc(&b.cand_rows, sizeof(uint32_t) * static_cast<size_t>(kBqMaxQueries) * b.cand_cap));
    HIP_TRY(h, hipMalloc(&b.counters, sizeof(int) * 8));   // [0..3]: batched.hip.h; [4]: the queued scan's arrival counter; [6]: cand_cap
    HIP_TRY(h, hipMemsetAsync(b.counters, 0, sizeof(int) * 8, h->stream));
    HIP_TRY(h, hipMemcpyAsync(b.counters + 6, &b.cand_cap, sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMalloc(&b.special_rows, sizeof(uint32_t) * kBqSpecialCap));
    HIP_TRY(h, hipMalloc(&b.nb_vals, sizeof(uint32_t) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.gmax, sizeof(float) * static_cast<size_t>(grid) * kBqMaxBlocks * 64));
    // Room for the tile maxima of pass 1 (rows from the replica only).  Optional: without it pass 2 looks at every
    // (tile, query block) pair, as before.
    if (h->d_half) {
        const int64_t n_tiles = (h->n + 63) / 64;
        const int64_t visited = (n_tiles + bq_step1(h, n_tiles) - 1) / bq_step1(h, n_tiles);
        if (hipMalloc(&b.tile_max, sizeof(uint4) * static_cast<size_t>(visited) * (kBqMaxBlocks / 8) * 64) == hipSuccess) {
            b.tile_max_tiles = visited;
        } else {
            (void)hipGetLastError();
            b.tile_max = nullptr;
        }
    }
    HIP_TRY(h, hipMalloc(&b.queue, sizeof(int) * kBqMaxQueries));
    HIP_TRY(h, hipMalloc(&b.qlists, sizeof(uint64_t) * static_cast<size_t>(kBqMaxQueries) * b.qgrid * kMultiMaxTopK));
    HIP_TRY(h, hipMalloc(&b.d_queries, sizeof(float) * kBqMaxQueries * kDim));
    HIP_TRY(h, hipMalloc(&b.d_exclude, sizeof(long long) * kBqMaxQueries));
    for (int i = 0; i < mi355rec::Batched::kSlots; ++i) {
        HIP_TRY(h, hipHostMalloc(&b.h_queries[i], sizeof(float) * kBqMaxQueries * kDim, hipHostMallocDefault));
        HIP_TRY(h, hipHostMalloc(&b.h_exclude[i], sizeof(long long) * kBqMaxQueries, hipHostMallocDefault));
        HIP_TRY(h, hipEventCreateWithFlags(&b.slot_ev[i], hipEventDisableTiming));
    }
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(bq_select_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   static_cast<int>(sizeof(float) * grid * 2 * 5 + sizeof(int) * (kBqSelectBlock / 64) * 256)));
    // the tighter bound is only claimed where fp16 subnormals are demonstrably kept
    hipLaunchKernelGGL(bq_selfcheck_kernel, dim3(1), dim3(64), 0, h->stream, b.qnorm);
    float chk[2] = {0.0f, 0.0f};
    HIP_TRY(h, hipMemcpyAsync(chk, b.qnorm, sizeof chk, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const bool kept = chk[0] == 9.5367431640625e-07f && chk[1] > 2.9e-6f && chk[1] < 3.1e-6f;
    b.margin = kept ? kBqMargin : kBqMarginFlush;
    b.ready = true;
    return MI355REC_OK;
}

void free_bq(mi355rec* h) {
    auto& b = h->bq;
    void* dev[] = {b.bfrag, b.qnorm, b.qthr, b.qflags, b.cand_count, b.cand_rows, b.counters, b.special_rows, b.nb_vals,
                   b.gmax, b.queue, b.qlists, b.d_queries, b.d_exclude, b.tile_max};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    for (int i = 0; i < mi355rec::Batched::kSlots; ++i) {
        if (b.h_queries[i]) (void)hipHostFree(b.h_queries[i]);
        if (b.h_exclude[i]) (void)hipHostFree(b.h_exclude[i]);
        if (b.slot_ev[i]) (void)hipEventDestroy(b.slot_ev[i]);
    }
    b = mi355rec::Batched();
}

template <int NB, bool kFromReplica, bool kTileMax>
void launch_bq_passes(mi355rec* h, const float* d_queries, const long long* d_exclude, int count, int topn, hipStream_t s) {
    auto& b = h->bq;
    const int64_t n_tiles = (h->n + 63) / 64;   // a wave handles 64 rows (two 32-row MFMA tiles) at a time
    const int step1 = bq_step1(h, n_tiles);
    const size_t smem = sizeof(float) * b.grid * 2 * 5 + sizeof(int) * (kBqSelectBlock / 64) * 256;
    const uint2* half = reinterpret_cast<const uint2*>(h->d_half);
    // The queries are prepared by a launch of their own.  Folding it into pass 1's prologue (every workgroup builds
    // the fragments from the raw queries itself; bq_pass_kernel still can: prep_queries) was built and measured: the
    // launch it saves takes 4.4 us, the prologue it adds to each of pass 1's 1024 workgroups made pass 1 13 us slower
    // (10 M rows x 1024 queries: 102.5 instead of 89.7 us).
    // The same launch takes every query's NEIGHBOURHOOD bound (one workgroup each: batched.hip.h) when the queries exclude
    // rows — their own, for recommendByIndex — so that bq_select has it beside pass 1's group maxima.
    const int prep_blocks = (NB * 32 + 255) / 256;
    const bool nbhd = d_exclude != nullptr && h->n >= kNbhdRows;
    hipLaunchKernelGGL(bq_prepare_kernel, dim3(prep_blocks + (nbhd ? count : 0)), dim3(256), 0, s, d_queries, count, NB, b.bfrag,
                       b.qnorm, b.qflags, b.cand_count, b.counters, prep_blocks, h->d_feats, h->n, h->row_base, d_exclude, topn, b.nb_vals);
    d_queries = nullptr;
    int slot = timing_begin(h, h->ev_pass, h->n_pass_pairs, h->pass_launches, s);
    hipLaunchKernelGGL((bq_pass_kernel<NB, false, 0, kFromReplica, kTileMax>), dim3(b.grid), dim3(kBqPassBlock), 0, s, h->d_feats, h->n,
                       n_tiles, step1, b.bfrag, b.gmax, b.cand_count, b.cand_rows, b.counters, b.special_rows, half,
                       b.tile_max, step1, static_cast<const float*>(b.qthr), static_cast<const uint32_t*>(b.qflags),
                       d_queries, count, b.qnorm, b.qflags);
    timing_end(h, h->ev_pass, h->n_pass_pairs, slot, s);
    hipLaunchKernelGGL(bq_select_kernel, dim3(NB * 8), dim3(kBqSelectBlock), smem, s, b.gmax, b.grid, NB, topn, b.margin, b.bfrag,
                       b.qflags, b.qthr, nbhd ? static_cast<const uint32_t*>(b.nb_vals) : static_cast<const uint32_t*>(nullptr), count);
    int skip_step = step1;
    {   // experiment builds only: where does pass 2's time go (tools/bq_ab.sh)
        int v = 0;
        MI355REC_EXP_INT(v, "MI355REC_BQ_EXP", 1, 2);
        if (v == 1) skip_step = 1 << 30;   // no tile counts as visited: the new loop over ALL blocks of every tile
        if (v == 2 && kTileMax) {          // every visited tile skips ALL its blocks: what a tile costs without any
            static std::vector<float> inf(kBqMaxQueries, __builtin_inff());
            (void)hipMemcpyAsync(b.qthr, inf.data(), sizeof(float) * kBqMaxQueries, hipMemcpyHostToDevice, s);
        }
    }
    slot = timing_begin(h, h->ev_pass, h->n_pass_pairs, h->pass_launches, s);
    hipLaunchKernelGGL((bq_pass_kernel<NB, true, 0, kFromReplica, kTileMax>), dim3(b.grid2), dim3(kBqPassBlock), 0, s, h->d_feats, h->n,
                       n_tiles, 1, b.bfrag, b.gmax, b.cand_count, b.cand_rows, b.counters, b.special_rows, half,
                       b.tile_max, skip_step, static_cast<const float*>(b.qthr), static_cast<const uint32_t*>(b.qflags),
                       static_cast<const float*>(nullptr), 0, static_cast<float*>(nullptr), static_cast<uint32_t*>(nullptr));
    timing_end(h, h->ev_pass, h->n_pass_pairs, slot, s);
}

template <int NB>
void launch_bq_passes(mi355rec* h, const float* d_queries, const long long* d_exclude, int count, int topn, hipStream_t s) {
    // the passes read the fp16 replica when the handle has one (it holds their A operand ready-made)
    if (h->d_half && h->replica_mode != MI355REC_REPLICA_OFF) {
        // 512 queries and more: pass 1 also leaves the maxima of the tiles it looked at, pass 2 skips what they rule out
        if constexpr (NB >= 16) {
            const int64_t n_tiles = (h->n + 63) / 64;
            const int step1 = bq_step1(h, n_tiles);
            if (h->batch_path != MI355REC_BATCH_MFMA_NOSKIP && h->bq.tile_max && (n_tiles + step1 - 1) / step1 <= h->bq.tile_max_tiles) {
                launch_bq_passes<NB, true, true>(h, d_queries, d_exclude, count, topn, s);
                return;
            }
        }
        launch_bq_passes<NB, true, false>(h, d_queries, d_exclude, count, topn, s);
    } else {
        launch_bq_passes<NB, false, false>(h, d_queries, d_exclude, count, topn, s);
    }
}

// One chunk of up to kBqMaxQueries queries that are already in device memory.
int enqueue_bq_chunk(mi355rec* h, const float* d_queries, const long long* d_exclude, int count, int topn,
                     uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    auto& b = h->bq;
    const int blocks = (count + 31) / 32;
    int nb = 1;
    while (nb < blocks) nb *= 2;
    switch (nb) {
        case 1: launch_bq_passes<1>(h, d_queries, d_exclude, count, topn, s); break;
        case 2: launch_bq_passes<2>(h, d_queries, d_exclude, count, topn, s); break;
        case 4: launch_bq_passes<4>(h, d_queries, d_exclude, count, topn, s); break;
        case 8: launch_bq_passes<8>(h, d_queries, d_exclude, count, topn, s); break;
        case 16: launch_bq_passes<16>(h, d_queries, d_exclude, count, topn, s); break;
        default: launch_bq_passes<32>(h, d_queries, d_exclude, count, topn, s); break;
    }
    hipLaunchKernelGGL(bq_finalize_kernel, dim3(count), dim3(kBqFinalBlock), 0, s, h->d_feats, h->row_base, d_queries,
                       d_exclude, count, topn, b.qflags, b.cand_count, b.cand_rows, b.cand_cap, b.counters, b.special_rows, b.queue,
                       out_keys, out_idx, out_score,
                       (d_exclude != nullptr && h->n >= kNbhdRows) ? static_cast<const uint32_t*>(b.nb_vals) : static_cast<const uint32_t*>(nullptr));
    // The exact multi-query scan for whatever the bound could not be claimed for, its merge included (usually
    // nothing: the launch exits at once on an empty queue).
    hipLaunchKernelGGL((scan_multi_queued_kernel<MultiConfig>), dim3(b.qgrid), dim3(MultiConfig::kBlock), 0, s,
                       h->d_feats, h->n, b.qiters, h->row_base, d_queries, d_exclude, b.queue, b.counters + 1, topn,
                       b.qlists, reinterpret_cast<unsigned*>(b.counters + 4), out_keys, out_idx, out_score,
                       static_cast<int64_t>(topn));
    HIP_TRY(h, hipGetLastError());
    ++b.launches;
    ++h->routes.mfma_two_pass;
    b.last_count = count;
    return MI355REC_OK;
}

// Host queries: through a pinned staging slot into the handle's device buffers.
int stage_queries(mi355rec* h, const float* queries, const int64_t* exclude, int count, hipStream_t s) {
    auto& b = h->bq;
    const int slot = b.next_slot;
    b.next_slot = (slot + 1) % mi355rec::Batched::kSlots;
    if (b.slot_used[slot]) HIP_TRY(h, hipEventSynchronize(b.slot_ev[slot]));  // its previous copy has long finished
    std::memcpy(b.h_queries[slot], queries, sizeof(float) * static_cast<size_t>(count) * kDim);
    for (int i = 0; i < count; ++i) b.h_exclude[slot][i] = exclude ? static_cast<long long>(exclude[i]) : -1ll;
    HIP_TRY(h, hipMemcpyAsync(b.d_queries, b.h_queries[slot], sizeof(float) * static_cast<size_t>(count) * kDim,
                              hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(b.d_exclude, b.h_exclude[slot], sizeof(long long) * static_cast<size_t>(count),
                              hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipEventRecord(b.slot_ev[slot], s));
    b.slot_used[slot] = true;
    return MI355REC_OK;
}

bool use_bq(const mi355rec* h, int batch, int topn) {
    if (topn > kMultiMaxTopK || h->n < 1) return false;
    if (h->batch_path == MI355REC_BATCH_MULTI || h->batch_path == MI355REC_BATCH_HALF || h->batch_path == MI355REC_BATCH_Q8) return false;
    if (h->batch_path == MI355REC_BATCH_MFMA || h->batch_path == MI355REC_BATCH_MFMA_NOSKIP) return true;
    const bool replica = h->d_half && h->replica_mode != MI355REC_REPLICA_OFF;
    return batch >= (replica ? kBqMinBatchReplica : kBqMinBatch) && h->n >= kBqMinRows;
}

int enqueue_bq_host(mi355rec* h, const float* queries, const int64_t* exclude, int batch, int topn,
                    uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    int rc = ensure_bq(h);
    if (rc) return rc;
    for (int b0 = 0; b0 < batch; b0 += kBqMaxQueries) {
        const int count = batch - b0 < kBqMaxQueries ? batch - b0 : kBqMaxQueries;
        rc = stage_queries(h, queries + static_cast<size_t>(b0) * kDim, exclude ? exclude + b0 : nullptr, count, s);
        if (rc) return rc;
        const size_t off = static_cast<size_t>(b0) * topn;
        rc = enqueue_bq_chunk(h, h->bq.d_queries, h->bq.d_exclude, count, topn, out_keys + off,
                              out_idx ? out_idx + off : nullptr, out_score ? out_score + off : nullptr, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// `batch` queries on stream `s`: multi-query passes where they apply (topn <=
// kMultiMaxTopK), otherwise one scan per query.  Outputs are batch x topn.
int enqueue_batch(mi355rec* h, const float* queries, const int64_t* exclude_global, int batch, int topn,
                  uint64_t* out_keys, int64_t* out_idx, float* out_score, hipStream_t s) {
    // 2 ... kHmAutoMax queries on a shard with a replica: multi-query passes over the replica (24 B/row,
    // one pass per 12 queries); more: the matrix-core path (two passes whatever the count up to 1024)
    const bool half_multi = half_multi_ok(h, topn) && h->n >= kBqMinRows &&
                            (h->batch_path == MI355REC_BATCH_HALF || h->batch_path == MI355REC_BATCH_Q8 ||
                             (h->batch_path == MI355REC_BATCH_AUTO && batch >= 2 && batch <= kHmAutoMax));
    if (half_multi) {
        for (int b = 0; b < batch; b += kMultiChain) {
            const int count = batch - b < kMultiChain ? batch - b : kMultiChain;
            const size_t off = static_cast<size_t>(b) * topn;
            const int rc = enqueue_half_multi(h, queries + static_cast<size_t>(b) * kDim, nullptr,
                                              exclude_global ? exclude_global + b : nullptr, count, topn, out_keys + off,
                                              out_idx ? out_idx + off : nullptr, out_score ? out_score + off : nullptr, s);
            if (rc) return rc;
        }
        return MI355REC_OK;
    }
    if (use_bq(h, batch, topn))
        return enqueue_bq_host(h, queries, exclude_global, batch, topn, out_keys, out_idx, out_score, s);
    if (batch > 1 && topn <= kMultiMaxTopK && h->n > 0) {
        for (int b = 0; b < batch; b += kMultiChain) {
            const int count = batch - b < kMultiChain ? batch - b : kMultiChain;
            const size_t off = static_cast<size_t>(b) * topn;
            const int rc = enqueue_multi(h, queries + static_cast<size_t>(b) * kDim,
                                         exclude_global ? exclude_global + b : nullptr, count, topn, out_keys + off,
                                         out_idx ? out_idx + off : nullptr, out_score ? out_score + off : nullptr, s);
            if (rc) return rc;
        }
        return MI355REC_OK;
    }
    for (int b = 0; b < batch; ++b) {
        const size_t off = static_cast<size_t>(b) * topn;
        const int rc = enqueue_query(h, nullptr, queries + static_cast<size_t>(b) * kDim,
                                     exclude_global ? exclude_global[b] : -1, topn, out_keys + off,
                                     out_idx ? out_idx + off : nullptr, out_score ? out_score + off : nullptr, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

}  // namespace

extern "C" {

int mi355rec_build_flags(void) {
    int flags = 0;
#ifdef MI355REC_EXPERIMENTS
    flags |= MI355REC_BUILD_EXPERIMENTS;
#endif
#ifdef MI355REC_PHASE_CLOCK
    flags |= MI355REC_BUILD_PHASE_CLOCK;
#endif
    return flags;
}

int mi355rec_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

const char* mi355rec_last_global_error(void) { return g_last_error.c_str(); }

int mi355rec_create(const float* feats_host, int64_t n, int dim, int device, int64_t row_base,
                    mi355rec_t** out) {
    return create_common(feats_host, false, n, dim, device, row_base, 0, out);
}

int mi355rec_create_device(const float* feats_dev, int64_t n, int dim, int device,
                           int64_t row_base, mi355rec_t** out) {
    return create_common(feats_dev, true, n, dim, device, row_base, 0, out);
}

int mi355rec_create_ex(const float* feats_host, int64_t n, int dim, int device, int64_t row_base, int flags,
                       mi355rec_t** out) {
    return create_common(feats_host, false, n, dim, device, row_base, flags, out);
}

int mi355rec_create_device_ex(const float* feats_dev, int64_t n, int dim, int device, int64_t row_base, int flags,
                              mi355rec_t** out) {
    return create_common(feats_dev, true, n, dim, device, row_base, flags, out);
}

void mi355rec_destroy(mi355rec_t* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipEvent_t e : h->ev_scan) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_merge) (void)hipEventDestroy(e);
    free_bq(h);
    for (int i = 0; i < 2; ++i) {
        if (h->d_mstream_lists[i]) (void)hipFree(h->d_mstream_lists[i]);
        if (h->d_mstream_seed[i]) (void)hipFree(h->d_mstream_seed[i]);
    }
    if (h->d_mstream_cuts) (void)hipFree(h->d_mstream_cuts);
    if (h->d_mstream_ctl) (void)hipFree(h->d_mstream_ctl);
    for (hipEvent_t e : h->ev_pass) (void)hipEventDestroy(e);
    if (h->owned_feats) (void)hipFree(h->owned_feats);
    if (h->d_block_lists) (void)hipFree(h->d_block_lists);
    if (h->d_lone_ctr) (void)hipFree(h->d_lone_ctr);
    {
        void* bufs[] = {h->d_half_seed, h->d_stream_seed[0], h->d_stream_seed[1], h->d_stream_ctl, h->d_lone_ctl};
        for (void* b : bufs)
            if (b) (void)hipFree(b);
    }
    if (h->d_stream_lists[0]) (void)hipFree(h->d_stream_lists[0]);
    if (h->d_stream_lists[1]) (void)hipFree(h->d_stream_lists[1]);
    if (h->d_seed_keys) (void)hipFree(h->d_seed_keys);
    if (h->d_seed_vals) (void)hipFree(h->d_seed_vals);
    free_replica(h);
    if (h->d_keys) (void)hipFree(h->d_keys);
    if (h->d_idx) (void)hipFree(h->d_idx);
    if (h->d_score) (void)hipFree(h->d_score);
    if (h->d_scores_full) (void)hipFree(h->d_scores_full);
    if (h->h_idx) (void)hipHostFree(h->h_idx);
    if (h->h_score) (void)hipHostFree(h->h_score);
    if (h->h_done) (void)hipHostFree(h->h_done);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->order_ev) (void)hipEventDestroy(h->order_ev);
    delete h;
}

const char* mi355rec_last_error(const mi355rec_t* h) {
    return h ? h->err.c_str() : g_last_error.c_str();
}

int mi355rec_set_timing(mi355rec_t* h, int enabled) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    h->timing = enabled != 0;
    h->timing_stride = enabled > 1 ? enabled : 1;
    h->n_scan_pairs = 0;
    h->n_merge_pairs = 0;
    h->n_pass_pairs = 0;
    h->scan_launches = 0;
    h->merge_launches = 0;
    h->pass_launches = 0;
    return MI355REC_OK;
}

int mi355rec_stats(const mi355rec_t* hc, mi355rec_stats_t* out) {
    mi355rec_t* h = const_cast<mi355rec_t*>(hc);
    if (!h || !out) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(h->device);
    // average over the pairs recorded since set_timing(1)
    auto avg = [&](std::vector<hipEvent_t>& evs, int pairs, float& last) {
        if (pairs <= 0) return;
        double sum = 0.0;
        int good = 0;
        for (int i = 0; i < pairs; ++i) {
            float ms = 0.f;
            if (hipEventSynchronize(evs[2 * i + 1]) == hipSuccess &&
                hipEventElapsedTime(&ms, evs[2 * i], evs[2 * i + 1]) == hipSuccess) {
                sum += ms;
                ++good;
            }
        }
        if (good) last = static_cast<float>(sum / good);
    };
    avg(h->ev_scan, h->n_scan_pairs, h->last_scan_ms);
    avg(h->ev_merge, h->n_merge_pairs, h->last_merge_ms);
    avg(h->ev_pass, h->n_pass_pairs, h->last_pass_ms);
    out->rows = h->n;
    out->row_base = h->row_base;
    out->device = h->device;
    out->compute_units = h->cus;
    out->grid_blocks = h->grid;
    out->block_threads = kScanBlock;
    out->bytes_per_query = h->n * kDim * static_cast<int64_t>(sizeof(float));
    out->last_scan_ms = h->last_scan_ms;
    out->last_merge_ms = h->last_merge_ms;
    out->last_pass_ms = h->last_pass_ms;
    out->batched_grid_blocks = h->bq.ready ? h->bq.grid : 0;
    out->batched_margin = h->bq.ready ? h->bq.margin : 0.0f;
    out->replica_bytes_per_query = h->d_half ? ((h->n + 1) / 2) * 48 : 0;
    out->replica_active = use_half(h, nullptr) ? 1 : 0;
    out->replica_grid_blocks = h->d_half ? (use_q8(h) ? h->qg.grid : h->hg.grid) : 0;
    out->replica_build_ms = h->replica_build_ms;
    out->replica_margin_single = h->d_half ? h->margin_mix : 0.0f;
    out->replica_margin_multi = h->d_half ? h->margin_mfma : 0.0f;
    out->replica_single_row_bytes = !h->d_half ? 0 : (use_q8(h) ? 12 : 24);
    out->replica_single_bytes_per_query = !h->d_half ? 0 : (use_q8(h) ? ((h->n + 3) / 4) * 48 : ((h->n + 1) / 2) * 48);
    out->lone_fused_queries = static_cast<int32_t>(h->lone_fused & 0x7fffffff);
    out->route_fp32 = h->routes.fp32;
    out->route_fp16 = h->routes.fp16;
    out->route_q8 = h->routes.q8;
    out->route_q8_lone = h->routes.q8_lone;
    out->route_multi_fp32 = h->routes.multi_fp32;
    out->route_multi_fp16 = h->routes.multi_fp16;
    out->route_multi_q8 = h->routes.multi_q8;
    out->route_mfma_two_pass = h->routes.mfma_two_pass;
    out->route_exact_queue = 0;
    if (h->bq.ready) {
        // queries the batched path handed to the exact scan: counted on the device (counters[5], never reset).  Read on the
        // handle's own stream (behind what the synchronous API enqueued there) — not a device-wide synchronisation: a
        // serving loop that polls the statistics does not stall the other streams.  Work still in flight on a caller's
        // stream is not in the figure yet.
        int queued = 0;
        if (hipMemcpyAsync(&queued, h->bq.counters + 5, sizeof queued, hipMemcpyDeviceToHost, h->stream) == hipSuccess &&
            hipStreamSynchronize(h->stream) == hipSuccess)
            out->route_exact_queue = queued;
    }
    out->device_bytes_per_row = 48 + (h->d_half ? 24 : 0) + (h->d_q8 ? 12 : 0);
    return MI355REC_OK;
}

// ---- asynchronous device API -------------------------------------------------

int mi355rec_enqueue_row_keys(mi355rec_t* h, int64_t local_row, int topn,
                              mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    rc = order_stream(h, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    return enqueue_query(h, h->d_feats + local_row * kDim, nullptr, h->row_base + local_row, topn, out_keys_dev, nullptr,
                         nullptr, static_cast<hipStream_t>(stream));
}

int mi355rec_enqueue_query_keys(mi355rec_t* h, const float* query12, int64_t exclude_global,
                                int topn, mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev || !query12) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    rc = order_stream(h, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    return enqueue_query(h, nullptr, query12, exclude_global, topn, out_keys_dev, nullptr, nullptr,
                         static_cast<hipStream_t>(stream));
}

int mi355rec_enqueue_row_keys_streamed(mi355rec_t* h, int64_t local_row, int topn, mi355rec_key_t* out_keys_dev,
                                       void* stream) {
    if (!h || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    return enqueue_streamed(h, h->d_feats + local_row * kDim, nullptr, h->row_base + local_row, topn, out_keys_dev, s);
}

int mi355rec_enqueue_query_keys_streamed(mi355rec_t* h, const float* query12, int64_t exclude_global, int topn,
                                         mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev || !query12) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    if (h->n < 1) {   // an EMPTY shard answers with an all-empty list at once (it still takes part in the merge)
        HIP_TRY(h, hipMemsetAsync(out_keys_dev, 0, sizeof(uint64_t) * static_cast<size_t>(topn), s));
        return MI355REC_OK;
    }
    return enqueue_streamed(h, nullptr, query12, exclude_global, topn, out_keys_dev, s);
}

int mi355rec_enqueue_batch_keys_streamed(mi355rec_t* h, const float* queries, const int64_t* exclude_global, int batch,
                                         int topn, mi355rec_key_t* out_keys_dev, void* stream) {
    if (!queries) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    return mi355rec_enqueue_batch_mixed_keys_streamed(h, queries, nullptr, exclude_global, batch, topn, out_keys_dev, stream);
}

int mi355rec_enqueue_batch_mixed_keys_streamed(mi355rec_t* h, const float* queries, const float* const* query_ptrs_dev,
                                               const int64_t* exclude_global, int batch, int topn,
                                               mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev || (!queries && !query_ptrs_dev)) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    const bool streamable = h->n >= kBqMinRows && half_multi_ok(h, topn) && h->batch_path != MI355REC_BATCH_MULTI &&
                            h->batch_path != MI355REC_BATCH_MFMA && h->batch_path != MI355REC_BATCH_MFMA_NOSKIP;
    if (!streamable) {
        // nothing to stream on: the batch is served at once (complete in stream order behind this call)
        if (query_ptrs_dev) return mi355rec_enqueue_batch_mixed_keys(h, queries, query_ptrs_dev, exclude_global, batch, topn, out_keys_dev, stream);
        return mi355rec_enqueue_batch_keys(h, queries, exclude_global, batch, topn, out_keys_dev, stream);
    }
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    rc = flush_streamed(h, s);   // a stream of SINGLE queries on this handle is closed first
    if (rc) return rc;
    for (int g0 = 0; g0 < batch; g0 += kHmQueries) {
        const int nq = batch - g0 < kHmQueries ? batch - g0 : kHmQueries;
        rc = enqueue_mstream(h, queries, query_ptrs_dev, exclude_global, g0, nq, topn, out_keys_dev + static_cast<size_t>(g0) * topn, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// ---- a batch whose queries are partly vectors, partly pointers (the sharded engine's windows) ----

int mi355rec_batch_pointers_ok(const mi355rec_t* h, int topn) {
    return h && h->n >= kBqMinRows && half_multi_ok(h, topn) ? 1 : 0;
}

int mi355rec_enqueue_batch_mixed_keys(mi355rec_t* h, const float* queries, const float* const* query_ptrs_dev,
                                      const int64_t* exclude_global, int batch, int topn, mi355rec_key_t* out_keys_dev,
                                      void* stream) {
    if (!h || !out_keys_dev || (!queries && !query_ptrs_dev)) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    if (h->n < 1) {   // an empty shard answers with all-empty lists
        HIP_TRY(h, hipMemsetAsync(out_keys_dev, 0, sizeof(uint64_t) * static_cast<size_t>(batch) * topn, s));
        return MI355REC_OK;
    }
    if (batch >= 2 && mi355rec_batch_pointers_ok(h, topn) && h->batch_path != MI355REC_BATCH_MULTI) {
        for (int b = 0; b < batch; b += kMultiChain) {
            const int count = batch - b < kMultiChain ? batch - b : kMultiChain;
            rc = enqueue_half_multi(h, queries ? queries + static_cast<size_t>(b) * kDim : nullptr,
                                    query_ptrs_dev ? query_ptrs_dev + b : nullptr, exclude_global ? exclude_global + b : nullptr,
                                    count, topn, out_keys_dev + static_cast<size_t>(b) * topn, nullptr, nullptr, s);
            if (rc) return rc;
        }
        return MI355REC_OK;
    }
    // no multi-query pass over a replica here: one scan per query, either kind of query
    for (int b = 0; b < batch; ++b) {
        const float* ptr = query_ptrs_dev ? query_ptrs_dev[b] : nullptr;
        if (!ptr && !queries) return fail(h, MI355REC_ERR_INVALID_ARG, "query %d has neither a vector nor a pointer", b);
        rc = enqueue_query(h, ptr, queries ? queries + static_cast<size_t>(b) * kDim : nullptr,
                           exclude_global ? exclude_global[b] : -1, topn, out_keys_dev + static_cast<size_t>(b) * topn, nullptr,
                           nullptr, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// ---- queries whose 12 floats already live in device-readable memory -----------------

int mi355rec_row_ptr(mi355rec_t* h, int64_t local_row, const float** out_dev) {
    if (!h || !out_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    *out_dev = h->d_feats + local_row * kDim;
    return MI355REC_OK;
}

int mi355rec_enqueue_ptr_keys(mi355rec_t* h, const float* query12_dev, int64_t exclude_global, int topn,
                              mi355rec_key_t* out_keys_dev, int64_t* out_idx_dev, float* out_score_dev, void* stream) {
    if (!h || !out_keys_dev || !query12_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    return enqueue_query(h, h->n > 0 ? query12_dev : nullptr, query12_dev, exclude_global, topn, out_keys_dev, out_idx_dev,
                         out_score_dev, s);
}

int mi355rec_enqueue_ptr_keys_streamed(mi355rec_t* h, const float* query12_dev, int64_t exclude_global, int topn,
                                       mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !out_keys_dev || !query12_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    if (h->n < 1) {
        HIP_TRY(h, hipMemsetAsync(out_keys_dev, 0, sizeof(uint64_t) * static_cast<size_t>(topn), s));
        return MI355REC_OK;
    }
    return enqueue_streamed(h, query12_dev, nullptr, exclude_global, topn, out_keys_dev, s);
}

int mi355rec_enqueue_flush(mi355rec_t* h, void* stream) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = order_stream(h, s);
    if (rc) return rc;
    rc = flush_streamed(h, s);
    if (rc) return rc;
    return flush_mstream(h, s);
}

int mi355rec_enqueue_batch_keys(mi355rec_t* h, const float* queries, const int64_t* exclude_global,
                                int batch, int topn, mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !queries || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    return enqueue_batch(h, queries, exclude_global, batch, topn, out_keys_dev, nullptr, nullptr, s);
}

int mi355rec_enqueue_batch_keys_dev(mi355rec_t* h, const float* queries_dev, const int64_t* exclude_global_dev,
                                    int batch, int topn, mi355rec_key_t* out_keys_dev, void* stream) {
    if (!h || !queries_dev || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    if (topn <= 0 || topn > kMultiMaxTopK)
        return fail(h, MI355REC_ERR_INVALID_ARG, "topn must be in [1, %d] for device-resident batches, got %d",
                    kMultiMaxTopK, topn);
    if (h->n < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "empty shard: use the host-query entry point");
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = order_stream(h, s);
    if (rc) return rc;
    rc = ensure_bq(h);
    if (rc) return rc;
    static_assert(sizeof(long long) == sizeof(int64_t), "exclude ids are passed through unchanged");
    for (int b0 = 0; b0 < batch; b0 += kBqMaxQueries) {
        const int count = batch - b0 < kBqMaxQueries ? batch - b0 : kBqMaxQueries;
        rc = enqueue_bq_chunk(h, queries_dev + static_cast<size_t>(b0) * kDim,
                              exclude_global_dev ? reinterpret_cast<const long long*>(exclude_global_dev) + b0 : nullptr,
                              count, topn, out_keys_dev + static_cast<size_t>(b0) * topn, nullptr, nullptr, s);
        if (rc) return rc;
    }
    return MI355REC_OK;
}

int mi355rec_batched_last_counters(mi355rec_t* h, int32_t* special_rows, int32_t* queued_queries,
                                   int64_t* candidates_total, int32_t* candidates_max) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (!h->bq.ready) return fail(h, MI355REC_ERR_INVALID_ARG, "no batched call has been made on this handle");
    DeviceGuard guard(h->device);
    HIP_TRY(h, hipDeviceSynchronize());
    int counters[4] = {0, 0, 0, 0};
    std::vector<int> cand(kBqMaxQueries);
    std::vector<uint32_t> flags(kBqMaxQueries);
    HIP_TRY(h, hipMemcpy(counters, h->bq.counters, sizeof counters, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy2D(cand.data(), sizeof(int), h->bq.cand_count, sizeof(int) * kBqCountStride, sizeof(int),
                           kBqMaxQueries, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(flags.data(), h->bq.qflags, sizeof(uint32_t) * kBqMaxQueries, hipMemcpyDeviceToHost));
    int64_t total = 0;
    int mx = 0;
    for (int q = 0; q < h->bq.last_count; ++q) {   // slots past the last chunk hold an earlier chunk's values, or nothing
        if (flags[q] != kBqFlagOk) continue;       // queued to the exact scan
        total += cand[q];
        if (cand[q] > mx) mx = cand[q];
    }
    if (special_rows) *special_rows = counters[0];
    if (queued_queries) *queued_queries = counters[1];
    if (candidates_total) *candidates_total = total;
    if (candidates_max) *candidates_max = mx;
    return MI355REC_OK;
}

int mi355rec_batched_pass2_pairs(mi355rec_t* h, int64_t* pairs_done, int64_t* pairs_total) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (!h->bq.ready) return fail(h, MI355REC_ERR_INVALID_ARG, "no batched call has been made on this handle");
    DeviceGuard guard(h->device);
    HIP_TRY(h, hipDeviceSynchronize());
    int counters[4] = {0, 0, 0, 0};
    HIP_TRY(h, hipMemcpy(counters, h->bq.counters, sizeof counters, hipMemcpyDeviceToHost));
    const int blocks = (h->bq.last_count + 31) / 32;
    int nb = 1;
    while (nb < blocks) nb *= 2;
    const int64_t total = ((h->n + 63) / 64) * nb;
    if (pairs_total) *pairs_total = total;
    // (0 = the last chunk ran without tile maxima: every pair was looked at)
    if (pairs_done) *pairs_done = counters[3] > 0 ? counters[3] : total;
    return MI355REC_OK;
}

int mi355rec_set_batch_path(mi355rec_t* h, int path) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (path != MI355REC_BATCH_AUTO && path != MI355REC_BATCH_MULTI && path != MI355REC_BATCH_MFMA && path != MI355REC_BATCH_HALF &&
        path != MI355REC_BATCH_Q8 && path != MI355REC_BATCH_MFMA_NOSKIP)
        return fail(h, MI355REC_ERR_INVALID_ARG, "unknown batch path %d", path);
#ifndef MI355REC_EXPERIMENTS
    if (path == MI355REC_BATCH_Q8)
        return fail(h, MI355REC_ERR_INVALID_ARG, "MI355REC_BATCH_Q8 (the 8-bit front end of the multi-query pass) exists in MI355REC_EXPERIMENTS builds only");
#endif
    h->batch_path = path;
    return MI355REC_OK;
}

int mi355rec_set_replica(mi355rec_t* h, int mode) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (mode != MI355REC_REPLICA_AUTO && mode != MI355REC_REPLICA_OFF && mode != MI355REC_REPLICA_ON && mode != MI355REC_REPLICA_FP16)
        return fail(h, MI355REC_ERR_INVALID_ARG, "unknown replica mode %d", mode);
#ifndef MI355REC_EXPERIMENTS
    if (mode == MI355REC_REPLICA_FP16)
        return fail(h, MI355REC_ERR_INVALID_ARG, "MI355REC_REPLICA_FP16 (single queries over the fp16 replica) exists in MI355REC_EXPERIMENTS builds only");
#endif
    if ((mode == MI355REC_REPLICA_ON || mode == MI355REC_REPLICA_FP16) && !h->d_half && h->n > 0) {
        if (!h->replica_allowed)
            return fail(h, MI355REC_ERR_INVALID_ARG, "this handle was created without a replica (MI355REC_CREATE_NO_REPLICA)");
        // a small shard (or one whose replica could not be allocated at create): build it now
        const int rc = mi355rec_rebuild_replica(h);
        if (rc) return rc;
    }
    h->replica_mode = mode;
    return MI355REC_OK;
}

// Test hook for the hand-offs that must fail safe (replica.hip.h): see include/mi355rec.h.
int mi355rec_debug_handoff(mi355rec_t* h, int flags) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    DeviceGuard guard(h->device);
    if (flags & MI355REC_DEBUG_HANDOFF_POISON) {
        HIP_TRY(h, hipDeviceSynchronize());
        // what a reader would find if the stores it depends on had not landed: the values of EARLIER queries (the
        // buffers alternate), here the most hostile ones — a perfect score under each of the last three epochs
        const uint32_t one = score_to_ordered(1.0f);
        const size_t per = static_cast<size_t>(kSampleSlots);   // (the neighbourhood's slot included)
        std::vector<unsigned long long> vals(static_cast<size_t>(kHmSampleSlots));
        for (size_t i = 0; i < vals.size(); ++i)
            vals[i] = (static_cast<unsigned long long>(h->epoch_ctr - 1u - static_cast<uint32_t>(i % 3)) << 32) | one;
        unsigned long long* single[] = {h->d_half_seed, h->d_stream_seed[0], h->d_stream_seed[1]};
        for (unsigned long long* b : single)
            if (b) HIP_TRY(h, hipMemcpy(b, vals.data(), sizeof(unsigned long long) * per, hipMemcpyHostToDevice));
        unsigned long long* multi[] = {h->d_half_mseed, h->d_mstream_seed[0], h->d_mstream_seed[1]};
        for (unsigned long long* b : multi)
            if (b) HIP_TRY(h, hipMemcpy(b, vals.data(), sizeof(unsigned long long) * vals.size(), hipMemcpyHostToDevice));
        // ... and a cutoff of +1.0 (it would rule out every row) under the epoch before the current one
        unsigned int one_bits;
        const float onef = 1.0f;
        std::memcpy(&one_bits, &onef, sizeof one_bits);
        const unsigned long long stale_cut = (static_cast<unsigned long long>(h->epoch_ctr - 1u) << 32) | one_bits;
        if (h->d_stream_ctl) {
            SeedCtl ctl[2];
            HIP_TRY(h, hipMemcpy(ctl, h->d_stream_ctl, sizeof ctl, hipMemcpyDeviceToHost));
            ctl[0].cutoff = ctl[1].cutoff = stale_cut;   // (the arrival counters stay what they are)
            HIP_TRY(h, hipMemcpy(h->d_stream_ctl, ctl, sizeof ctl, hipMemcpyHostToDevice));
        }
        if (h->d_lone_ctl) {
            SeedCtl ctl;
            HIP_TRY(h, hipMemcpy(&ctl, h->d_lone_ctl, sizeof ctl, hipMemcpyDeviceToHost));
            ctl.cutoff = stale_cut;
            HIP_TRY(h, hipMemcpy(h->d_lone_ctl, &ctl, sizeof ctl, hipMemcpyHostToDevice));
        }
        std::vector<unsigned long long> cuts(2 * kHmQueries, stale_cut);
        if (h->d_mstream_cuts)
            HIP_TRY(h, hipMemcpy(h->d_mstream_cuts, cuts.data(), sizeof(unsigned long long) * cuts.size(), hipMemcpyHostToDevice));
        if (h->d_half_mcuts)
            HIP_TRY(h, hipMemcpy(h->d_half_mcuts, cuts.data(), sizeof(unsigned long long) * kHmQueries, hipMemcpyHostToDevice));
    }
    if (flags & MI355REC_DEBUG_HANDOFF_DROP_STORES) h->dbg_skip_regions = kHalfSeedMaxGrid / 2;
    if (flags & MI355REC_DEBUG_HANDOFF_NO_LAST_RIDER) h->dbg_no_last = true;
    return MI355REC_OK;
}

#ifdef MI355REC_PHASE_CLOCK   // tools/phase_clock.py builds only
int mi355rec_debug_phase_clock(unsigned long long* out, int n_words) {
    if (hipDeviceSynchronize() != hipSuccess) return MI355REC_ERR_HIP;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mi355::g_phase_clock), sizeof(unsigned long long) * n_words) == hipSuccess
               ? MI355REC_OK : MI355REC_ERR_HIP;
}
#endif

int mi355rec_replica_counters(mi355rec_t* h, int64_t* scans, int64_t* rescored_rows) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (scans) *scans = h->half_scans;
    if (rescored_rows) *rescored_rows = 0;
    if (!h->d_half_rescored || !rescored_rows) return MI355REC_OK;
    DeviceGuard guard(h->device);
    std::vector<unsigned long long> slots(kRideMaxLists);
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(slots.data(), h->d_half_rescored, sizeof(unsigned long long) * kRideMaxLists, hipMemcpyDeviceToHost));
    unsigned long long sum = 0;
    for (unsigned long long v : slots) sum += v;
    *rescored_rows = static_cast<int64_t>(sum);
    return MI355REC_OK;
}

int mi355rec_rebuild_replica(mi355rec_t* h) {
    if (!h) return fail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->n == 0) return MI355REC_OK;
    if (!h->replica_allowed)
        return fail(h, MI355REC_ERR_INVALID_ARG, "this handle was created without a replica (MI355REC_CREATE_NO_REPLICA)");
    DeviceGuard guard(h->device);
    int rc = sync_api_begin(h);
    if (rc) return rc;
    // A stashed streamed query carries a sample taken from the OLD replica, and the pending one's
    // lists wait for their merge: both are completed first (on the handle's own stream).
    rc = flush_streamed(h, h->stream);
    if (rc) return rc;
    rc = flush_mstream(h, h->stream);
    if (rc) return rc;
    return build_replica(h);
}

int mi355rec_enqueue_merge_keys(mi355rec_t* h, const mi355rec_key_t* lists_dev, int n_lists,
                                int list_len, int topn, mi355rec_key_t* out_keys_dev,
                                int64_t* out_idx_dev, float* out_score_dev, void* stream) {
    if (!h || !lists_dev || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (n_lists < 1 || n_lists > kMergeMaxLists || list_len < 1)
        return fail(h, MI355REC_ERR_INVALID_ARG, "n_lists %d / list_len %d out of range", n_lists, list_len);
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    rc = order_stream(h, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    return enqueue_merge(h, lists_dev, n_lists, list_len, topn, out_keys_dev, out_idx_dev,
                         out_score_dev, static_cast<hipStream_t>(stream));
}

int mi355rec_enqueue_merge_keys_batch(mi355rec_t* h, const mi355rec_key_t* lists_dev, int n_lists,
                                      int list_len, int64_t list_stride, int64_t query_stride, int batch,
                                      int topn, mi355rec_key_t* out_keys_dev, int64_t* out_idx_dev,
                                      float* out_score_dev, void* stream) {
    if (!h || !lists_dev || !out_keys_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (n_lists < 1 || n_lists > kMergeMaxLists || list_len < 1 || batch < 1 || list_stride < list_len)
        return fail(h, MI355REC_ERR_INVALID_ARG, "merge geometry out of range");
    int rc = check_topn(h, topn, false);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = order_stream(h, s);
    if (rc) return rc;
    const int slot = timing_begin(h, h->ev_merge, h->n_merge_pairs, h->merge_launches, s);
    hipLaunchKernelGGL(merge_kernel, dim3(batch), dim3(kMergeBlock), 0, s, lists_dev, n_lists, list_len,
                       list_stride, query_stride, topn, out_keys_dev, out_idx_dev, out_score_dev,
                       static_cast<int64_t>(topn));
    timing_end(h, h->ev_merge, h->n_merge_pairs, slot, s);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

int mi355rec_enqueue_scores(mi355rec_t* h, int64_t local_row, const float* query12,
                            float* out_scores_dev, void* stream) {
    if (!h || !out_scores_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row >= h->n) return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    if (local_row < 0 && !query12) return fail(h, MI355REC_ERR_INVALID_ARG, "null query");
    if (h->n == 0) return MI355REC_OK;  // empty shard: no scores
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = order_stream(h, s);
    if (rc) return rc;
    QueryArg qa;
    std::memset(&qa, 0, sizeof qa);
    qa.margin = h->margin_mix;
    NextSeed no_next;
    std::memset(&no_next, 0, sizeof no_next);
    if (local_row >= 0) {
        hipLaunchKernelGGL((scan_kernel<ScanConfig, true, true>), dim3(h->grid), dim3(kScanBlock), 0, s, h->d_feats,
                           h->n, h->rows_per_block, h->iters, h->row_base, qa, h->d_feats + local_row * kDim,
                           static_cast<int64_t>(-1), 1, static_cast<uint64_t*>(nullptr), out_scores_dev,
                           static_cast<const uint64_t*>(nullptr), PrevMerge{nullptr, 0, 0, nullptr},
                           static_cast<const unsigned long long*>(nullptr), static_cast<const unsigned long long*>(nullptr), 0u, no_next);
    } else {
        std::memcpy(qa.q, query12, sizeof qa.q);
        hipLaunchKernelGGL((scan_kernel<ScanConfig, false, true>), dim3(h->grid), dim3(kScanBlock), 0, s, h->d_feats,
                           h->n, h->rows_per_block, h->iters, h->row_base, qa, kNoQueryPtr,
                           static_cast<int64_t>(-1), 1, static_cast<uint64_t*>(nullptr), out_scores_dev,
                           static_cast<const uint64_t*>(nullptr), PrevMerge{nullptr, 0, 0, nullptr},
                           static_cast<const unsigned long long*>(nullptr), static_cast<const unsigned long long*>(nullptr), 0u, no_next);
    }
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

int mi355rec_enqueue_stream_probe(mi355rec_t* h, uint32_t* sink_dev, void* stream) {
    return mi355rec_enqueue_stream_probe_of(h, MI355REC_PROBE_FP32_ROWS, sink_dev, stream);
}

int mi355rec_enqueue_stream_probe_of(mi355rec_t* h, int which, uint32_t* sink_dev, void* stream) {
    if (!h || !sink_dev) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (h->n == 0) return MI355REC_OK;
    const float4* buf = nullptr;
    int64_t n_vec = 0;   // 16-byte vectors
    if (which == MI355REC_PROBE_FP32_ROWS) {
        buf = reinterpret_cast<const float4*>(h->d_feats);
        n_vec = h->n * 3;
    } else if (which == MI355REC_PROBE_FP16_REPLICA && h->d_half) {
        buf = reinterpret_cast<const float4*>(h->d_half);
        n_vec = ((h->n + 1) / 2) * 3;
    } else if (which == MI355REC_PROBE_Q8_REPLICA && h->d_q8) {
        buf = reinterpret_cast<const float4*>(h->d_q8);
        n_vec = ((h->n + 3) / 4) * 3;
    } else {
        return fail(h, MI355REC_ERR_INVALID_ARG, "no such buffer to probe (%d)", which);
    }
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    LAUNCH_TIMED(h, h->ev_scan, h->n_scan_pairs, h->scan_launches, stream_probe_kernel,
                 dim3(h->cus), dim3(kProbeBlock), s, buf, n_vec, sink_dev);
    HIP_TRY(h, hipGetLastError());
    return MI355REC_OK;
}

// ---- synchronous host API ------------------------------------------------------

static int scores_common(mi355rec_t* h, int64_t local_row, const float* query12, float* out_host) {
    if (!h || !out_host) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    DeviceGuard guard(h->device);
    if (h->n == 0) return MI355REC_OK;
    if (!h->d_scores_full) HIP_TRY(h, hipMalloc(&h->d_scores_full, sizeof(float) * static_cast<size_t>(h->n)));
    int rc = mi355rec_enqueue_scores(h, local_row, query12, h->d_scores_full, h->stream);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(out_host, h->d_scores_full, sizeof(float) * static_cast<size_t>(h->n),
                              hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MI355REC_OK;
}

int mi355rec_fetch_row(mi355rec_t* h, int64_t local_row, float* out12_host) {
    if (!h || !out12_host) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    DeviceGuard guard(h->device);
    HIP_TRY(h, hipMemcpy(out12_host, h->d_feats + local_row * kDim, sizeof(float) * kDim, hipMemcpyDeviceToHost));
    return MI355REC_OK;
}

int mi355rec_scores_row(mi355rec_t* h, int64_t local_row, float* out_host) {
    if (h && (local_row < 0 || local_row >= h->n))
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    return scores_common(h, local_row, nullptr, out_host);
}

int mi355rec_scores(mi355rec_t* h, const float* query12, float* out_host) {
    if (!query12) return fail(h, MI355REC_ERR_INVALID_ARG, "null query");
    return scores_common(h, -1, query12, out_host);
}

int mi355rec_query_batch_topn(mi355rec_t* h, const float* queries, int batch,
                              const int64_t* exclude_global, int topn, int64_t* out_idx,
                              float* out_score, int* out_count) {
    if (!h || !queries || !out_idx) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return fail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    // internal lists are `eff` long: a shard of n rows cannot return more
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    const size_t cnt = static_cast<size_t>(batch) * eff;
    rc = ensure_slots(h, cnt);
    if (rc) return rc;
    rc = sync_api_begin(h);
    if (rc) return rc;
    const bool direct = cnt <= static_cast<size_t>(kDirectResultSlots);
    rc = enqueue_batch(h, queries, exclude_global, batch, eff, h->d_keys, direct ? h->hd_idx : h->d_idx,
                       direct ? h->hd_score : h->d_score, h->stream);
    if (rc) return rc;
    if (!direct) {
        HIP_TRY(h, hipMemcpyAsync(h->h_idx, h->d_idx, cnt * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->h_score, h->d_score, cnt * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (int b = 0; b < batch; ++b) {
        const int64_t* src_i = h->h_idx + static_cast<size_t>(b) * eff;
        const float* src_s = h->h_score + static_cast<size_t>(b) * eff;
        int64_t* dst_i = out_idx + static_cast<size_t>(b) * topn;
        std::memcpy(dst_i, src_i, static_cast<size_t>(eff) * sizeof(int64_t));
        for (int i = eff; i < topn; ++i) dst_i[i] = -1;
        if (out_score) {
            float* dst_s = out_score + static_cast<size_t>(b) * topn;
            std::memcpy(dst_s, src_s, static_cast<size_t>(eff) * sizeof(float));
            for (int i = eff; i < topn; ++i) dst_s[i] = 0.0f;
        }
        if (out_count) {
            int c = 0;
            while (c < eff && src_i[c] >= 0) ++c;
            out_count[b] = c;
        }
    }
    return MI355REC_OK;
}

int mi355rec_query_topn(mi355rec_t* h, const float* query12, int64_t exclude_global, int topn,
                        int64_t* out_idx, float* out_score, int* out_count) {
    return mi355rec_query_batch_topn(h, query12, 1, &exclude_global, topn, out_idx, out_score, out_count);
}

int mi355rec_query_row_topn(mi355rec_t* h, int64_t local_row, int topn, int64_t* out_idx,
                            float* out_score, int* out_count) {
    if (!h || !out_idx) return fail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (local_row < 0 || local_row >= h->n)
        return fail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)local_row);
    int rc = check_topn(h, topn, true);
    if (rc) return rc;
    DeviceGuard guard(h->device);
    // device / pinned slots are sized by what the shard can return, not by topn
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    rc = ensure_slots(h, static_cast<size_t>(eff));
    if (rc) return rc;
    rc = sync_api_begin(h);
    if (rc) return rc;
    // Small results go straight into the pinned host buffers from the merge kernel
    // (zero-copy stores over PCIe: no D2H copy launches on the latency path).
    const bool direct = eff <= kDirectResultSlots;
    // one round (eff <= 1024): the merge kernel stores the results AND a completion word in pinned host
    // memory; the host spins on the word
    const bool notify = direct && eff <= kMaxTopK && eff > 0;
    const uint32_t want = notify ? (++h->done_seq ? h->done_seq : ++h->done_seq) : 0u;   // never 0
    rc = enqueue_query(h, h->d_feats + local_row * kDim, nullptr, h->row_base + local_row, eff, h->d_keys, direct ? h->hd_idx : h->d_idx,
                       direct ? h->hd_score : h->d_score, h->stream, want);
    if (rc) return rc;
    if (!direct) {
        HIP_TRY(h, hipMemcpyAsync(h->h_idx, h->d_idx, eff * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->h_score, h->d_score, eff * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    if (notify) {
        rc = wait_done(h, want);
        if (rc) return rc;
    } else {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    int c = 0;
    while (c < eff && h->h_idx[c] >= 0) ++c;
    std::memcpy(out_idx, h->h_idx, static_cast<size_t>(eff) * sizeof(int64_t));
    if (out_score) std::memcpy(out_score, h->h_score, static_cast<size_t>(eff) * sizeof(float));
    for (int i = eff; i < topn; ++i) {
        out_idx[i] = -1;
        if (out_score) out_score[i] = 0.0f;
    }
    if (out_count) *out_count = c;
    return MI355REC_OK;
}

// ---- key helpers -----------------------------------------------------------------

mi355rec_key_t mi355rec_pack_key(float score, int64_t global_row) {
    return pack_key(score, static_cast<uint32_t>(global_row));
}

float mi355rec_key_score(mi355rec_key_t key) {
    return key ? ordered_to_score(static_cast<uint32_t>(key >> 32)) : 0.0f;
}

int64_t mi355rec_key_row(mi355rec_key_t key) {
    return key ? static_cast<int64_t>(static_cast<uint32_t>(~static_cast<uint32_t>(key))) : -1;
}

}  // extern "C"
