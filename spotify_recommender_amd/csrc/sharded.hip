// sharded.hip — the row-sharded engine behind the C-ABI (include/mi355rec.h,
// "row-sharded catalogue" section): ONE host process drives every GPU of the node.
//
// Replaces the reference's single-device residency (cudaSetDevice(0),
// Recommender.cu:124): shard r owns the contiguous rows [lo_r, hi_r) on device
// devices[r] (an ordinary mi355rec handle created with row_base = lo_r, so its
// keys carry GLOBAL row ids).  Per query every shard runs the fused scan + local
// merge on its own stream, the per-shard top-N key lists meet on device
// devices[0], and the same merge kernel that merges per-workgroup lists merges
// the G per-shard lists.  The result does not depend on G.
//
// Exchange (the only inter-GPU step; 800 B per shard at top-100):
//   PEER  every shard's merge kernel STORES its keys straight into the gather
//         buffer on device 0 through a peer mapping (xGMI point-to-point), an
//         event per shard orders device 0's merge behind them.  No collective, no
//         copy launch.  Default when every device can map device 0's memory.
//   RCCL  one grouped ncclAllGather of topn x batch uint64 keys per shard
//         (ncclCommInitAll, ncclGroupStart/End; the exchange BASELINE.json's
//         north_star names).  librccl is opened lazily, only for this transport.
// This layer is host orchestration over the single-device C-ABI and the HIP
// runtime: it launches no kernel of its own.
//
// Two ways in:
//   * synchronous calls (mi355rec_sharded_query_*): what Recommender::recommendByIndex
//     maps to.  One shard -> straight to that shard's handle.  More: scan + local merge per
//     shard, exchange, merge on the first device straight into mapped host memory, ONE
//     host wait (the first device's stream; everything it depends on is ordered by events).
//   * the STREAM (mi355rec_sharded_enqueue_* / _flush / _wait): a serving loop.  One streamed
//     scan launch per shard per query, one exchange + one batched merge per WINDOW of
//     queries, tickets instead of host waits.
// Host threads.  With more than one shard every shard has a WORKER thread that owns its device context,
// its stream and its single-device handle: the caller's thread only posts small task records (a
// single-producer ring per worker) and the workers issue the HIP launches in parallel — one launch per
// shard per query costs the caller ~0.1 us instead of G x ~5 us, and a synchronous query on G GPUs pays
// one shard's launch chain, not G of them.  Cross-shard ordering stays on the device (one event per shard,
// or the collective); the only host-side hand-off is "shard r has RECORDED its event" before the first
// device's worker makes its stream wait for it.  With one shard there are no threads at all.
// A query by ROW never comes back to the host: every shard's kernels read its 48 bytes from
// the owning shard's memory through the peer mapping (checked once at create time against
// the by-value path; without all-pairs peer access the row is fetched once per query).
//
// PLACEMENT (mi355rec_create_placed).  Row sharding is what north_star specifies and the only way to a lower
// latency once a scan is longer than its launches; but a shard below ~4 M rows is launch-bound (measured on one
// MI355X, streamed queries over the 8-bit replica: 12.1 us per query at 1.25 M rows, 15.4 at 3 M, 24.4 at 10 M, 207
// at 100 M — 11 us of fixed cost + 1.95 us per million rows), and every query of a sharded catalogue pays the
// exchange on top.  So
//   SHARDED     rows split over G devices; G given, or (AUTO) as many as keep >= 4 M rows per shard: 1 for the 114 k
//               catalogue of BASELINE configs[0] and up to 7.9 M rows, 2 at 10 M, all 8 from 32 M rows on;
//   REPLICATED  every device holds ALL rows (840 MB at 10 M rows with both replicas) and serves whole WINDOWS of the
//               stream by itself, round-robin: no exchange, no merge across devices, queries/s scale with the
//               devices for any catalogue that fits one.  Synchronous calls go to the replicas in turn.
// A "shard" of a replicated handle is a full replica (lo = 0, hi = n).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cpu_backend.h"
#include "mi355rec.h"

namespace {

thread_local std::string g_sharded_error;
thread_local bool t_worker_thread = false;   // a worker reports through its Worker record, never through the handle's string

// ---- the few RCCL entry points, resolved at run time -------------------------------
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;   // ncclSuccess == 0
constexpr int kNcclUint64 = 5;   // ncclUint64 in nccl.h / rccl.h

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& why) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            why = std::string("cannot open librccl: ") + dlerror();
            return false;
        }
        auto sym = [&](const char* s) { return dlsym(lib, s); };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllGather || !GroupStart || !GroupEnd) {
            why = "librccl lacks the expected symbols";
            return false;
        }
        return true;
    }
};

Rccl g_rccl;

struct Shard {
    int device = 0;
    int64_t lo = 0, hi = 0;
    mi355rec_t* engine = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;          // this shard's keys are in place
    mi355rec_key_t* local_keys = nullptr;   // RCCL transport: send buffer on the shard's device
    mi355rec_key_t* gathered = nullptr;     // RCCL transport: receive buffer on the shard's device
    ncclComm_t comm = nullptr;
    // the stream of single queries (RCCL transport): [kStreamDepth][window][topn] send,
    // [kStreamDepth][shards][window][topn] receive
    mi355rec_key_t* s_local = nullptr;
    mi355rec_key_t* s_gathered = nullptr;
    // replicated placement: where the pinned result ring of the stream lives in THIS device's address space
    int64_t* s_hdidx = nullptr;
    float* s_hdscore = nullptr;
};

// ---- one unit of work for a shard's worker (plain data: copied into the worker's ring) --------------------
enum TaskKind {
    kTaskStreamQuery,   // one streamed scan launch: query by pointer or by value -> dst
    kTaskSyncQuery,     // scan + local merge now (one query, or `count` of them as a batch) -> dst
    kTaskBatch,         // a window: `count` queries (vectors and / or pointers) as a streamed batch -> dst
    kTaskFlush,         // mi355rec_enqueue_flush on the shard's engine
    kTaskPublish,       // PEER: record the shard's `done` event, then publish `seq` to the host
    kTaskAllGather,     // RCCL: this shard's ncclAllGather
    kTaskMerge,         // first device: wait for the other shards' events (PEER), merge the gathered lists
};

struct Task {
    int kind = 0;
    const float* qptr = nullptr;          // a query by pointer (device-readable memory)
    float q[MI355REC_DIM] = {0};          // ... or by value
    bool by_value = false;
    const float* queries = nullptr;       // a batch by value (host memory that stays valid until the task has run)
    const float* const* qptrs = nullptr;
    const int64_t* excls = nullptr;
    int64_t excl = -1;
    int count = 1, topn = 0;
    mi355rec_key_t* dst = nullptr;
    bool flush_after = false;             // kTaskBatch: drain the engine's pipeline right behind it
    uint64_t seq = 0;                     // exchange number (publish / merge)
    const mi355rec_key_t* send = nullptr; // all-gather
    mi355rec_key_t* recv = nullptr;
    size_t stride = 0;
    const mi355rec_key_t* lists = nullptr;   // merge
    int n_lists = 0;                         // ... of this many lists per query (0: one per shard)
    mi355rec_key_t* out_keys = nullptr;
    int64_t* out_idx = nullptr;
    float* out_score = nullptr;
    hipEvent_t record_after = nullptr;    // merge: recorded on the stream behind it (a window's `merged`)
    bool copy_back = false;               // merge: results too large for mapped stores: two copies to h_idx / h_score
};

struct Worker {
    static constexpr uint64_t kCap = 2048;
    std::thread th;
    std::vector<Task> ring;
    std::atomic<uint64_t> head{0};        // tasks posted
    std::atomic<uint64_t> tail{0};        // tasks done
    std::atomic<bool> sleeping{false};
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> published{0};   // the exchange number the shard's `done` event was last recorded for
    std::atomic<int> err{0};              // first failure of a task (sticky); err_msg is written before it
    std::string err_msg;
    std::mutex mu;
    std::condition_variable cv;
};

constexpr int kStreamDepth = 4;     // windows whose results are kept (ring)
constexpr int kStreamLag = 2;       // the keys of streamed query k are complete, in stream order, behind call k + 2
constexpr int kMaxWindow = 64;

struct Window {
    hipEvent_t merged = nullptr;    // first device: this window's batched merge has run (results are in host memory)
    int64_t abs = -1;               // which window of the stream the ring entry holds (-1: none)
    int count = 0;                  // queries in it
    bool handed = false;            // batched windows: the shards have received it (one streamed batch call each)
    bool issued = false;            // its exchange + merge have been enqueued (posted to the first device's worker)
    uint64_t merge_task = 0;        // ... as that worker's task number: `merged` is recorded once it has run
    int owner = 0;                  // replicated placement: the replica that serves this window (0 otherwise)
};
constexpr int kWindowLag = 2;       // a streamed batch is complete, in stream order, behind the second batch call after it

int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

struct mi355rec_sharded {
    int64_t n = 0;
    // A host WITHOUT a HIP device: the product's own CPU backend (csrc/cpu_backend.h; the reference's
    // "Falling back to CPU", Recommender.cu:117-127,176-181) serves the whole C-ABI of this handle and
    // everything below stays empty.  Never set when a device is visible.
    mi355cpu::Node* cpu = nullptr;
    std::vector<Shard> shards;
    int transport = MI355REC_TRANSPORT_PEER;
    bool peer_ok = true;
    bool rccl_ready = false;
    size_t cap = 0;                     // keys per shard the buffers hold (batch x topn)
    mi355rec_key_t* gather0 = nullptr;  // [shards][cap] on devices[0] (PEER transport)
    mi355rec_key_t* d_keys = nullptr;   // merged results on devices[0]
    int64_t* d_idx = nullptr;
    float* d_score = nullptr;
    int64_t* h_idx = nullptr;           // pinned + mapped: the final merge stores its results here itself
    float* h_score = nullptr;
    int64_t* hd_idx = nullptr;          // their device-side addresses
    float* hd_score = nullptr;
    bool peer_rows = true;              // every shard's device can read every other shard's rows
    bool batched_windows = true;        // mi355rec_sharded_set_window_mode
    bool replicated = false;            // every "shard" holds all rows (mi355rec_create_placed, MI355REC_PLACEMENT_REPLICATED)
    int next_replica = 0;               // whose turn the next synchronous call is (replicated)
    std::vector<hipEvent_t> r_merged;   // replicated: [kStreamDepth][replicas] "this window's results are in host memory",
                                        // each on its replica's device (an event belongs to a device)
    std::vector<std::unique_ptr<Worker>> workers;   // one per shard when there are several shards, none otherwise
    uint64_t exchange_seq = 0;
    std::string note;                   // why a fast path was switched off at create time (diagnostics)

    // ---- the stream of single queries -------------------------------------------------
    int s_topn = 0;                     // geometry the stream buffers were allocated for (0: none yet)
    int s_window = 16;
    int s_alloc_window = 0;
    mi355rec_key_t* s_gather0 = nullptr;   // first device: [kStreamDepth][shards][window][topn] (PEER transport)
    mi355rec_key_t* s_keys = nullptr;      // first device: [kStreamDepth][window][topn] merged keys
    int64_t* s_hidx = nullptr;             // pinned + mapped: [kStreamDepth][window][topn]
    float* s_hscore = nullptr;
    int64_t* s_hdidx = nullptr;
    float* s_hdscore = nullptr;
    Window win[kStreamDepth];
    // BATCHED windows: where every shard can take a window of queries in multi-query passes over its
    // replica (mi355rec_batch_pointers_ok), the queries of a window are only collected on the host and go
    // to every shard in ONE mi355rec_enqueue_batch_mixed_keys call when the window closes: 3 launches per
    // shard per window instead of one per query, and one pass over the shard per 32 queries.
    bool s_batched = false;
    std::vector<float> w_q;             // [kStreamDepth][window][12]
    std::vector<const float*> w_ptr;    // [kStreamDepth][window]
    std::vector<int64_t> w_excl;
    int64_t next_ticket = 0;            // tickets handed out so far (window-aligned after a flush)
    int64_t issued_upto = 0;            // every ticket below has had its window's exchange enqueued
    int64_t st_queries = 0, st_exchanges = 0, st_host_ns = 0;
    std::string err;
};

namespace {

int sfail(mi355rec_sharded* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h && !t_worker_thread) h->err = buf;
    g_sharded_error = buf;
    return code;
}

#define S_HIP(h, expr)                                                                             \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return sfail((h), e_ == hipErrorOutOfMemory ? MI355REC_ERR_OUT_OF_MEMORY : MI355REC_ERR_HIP, \
                         "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define S_ENG(h, shard, expr)                                                                      \
    do {                                                                                           \
        const int rc_ = (expr);                                                                    \
        if (rc_ != MI355REC_OK)                                                                    \
            return sfail((h), rc_, "shard on device %d: %s", (shard).device, mi355rec_last_error((shard).engine)); \
    } while (0)

// the CPU backend's calls report (code, message) like this
int cpu_result(mi355rec_sharded* h, int rc, const char* why) {
    return rc == MI355REC_OK ? rc : sfail(h, rc, "%s", why ? why : "CPU backend: invalid argument");
}

// balanced contiguous blocks: the first n % g shards hold one row more
void bounds(int64_t n, int g, int r, int64_t& lo, int64_t& hi) {
    const int64_t per = n / g, rem = n % g;
    lo = r * per + (r < rem ? r : rem);
    hi = lo + per + (r < rem ? 1 : 0);
}

int drain_workers(mi355rec_sharded* h);

int ensure_capacity(mi355rec_sharded* h, size_t keys_per_shard) {
    if (keys_per_shard <= h->cap) return MI355REC_OK;
    {
        const int rc = drain_workers(h);   // buffers are about to be replaced
        if (rc) return rc;
    }
    size_t cap = h->cap ? h->cap : 1024;
    while (cap < keys_per_shard) cap *= 2;
    const int g = static_cast<int>(h->shards.size());
    S_HIP(h, hipSetDevice(h->shards[0].device));
    S_HIP(h, hipDeviceSynchronize());
    if (h->gather0) (void)hipFree(h->gather0);
    if (h->d_keys) (void)hipFree(h->d_keys);
    if (h->d_idx) (void)hipFree(h->d_idx);
    if (h->d_score) (void)hipFree(h->d_score);
    if (h->h_idx) (void)hipHostFree(h->h_idx);
    if (h->h_score) (void)hipHostFree(h->h_score);
    h->gather0 = nullptr; h->d_keys = nullptr; h->d_idx = nullptr; h->d_score = nullptr;
    h->h_idx = nullptr; h->h_score = nullptr;
    h->cap = 0;
    S_HIP(h, hipMalloc(&h->gather0, sizeof(mi355rec_key_t) * cap * g));
    S_HIP(h, hipMalloc(&h->d_keys, sizeof(mi355rec_key_t) * cap));
    S_HIP(h, hipMalloc(&h->d_idx, sizeof(int64_t) * cap));
    S_HIP(h, hipMalloc(&h->d_score, sizeof(float) * cap));
    S_HIP(h, hipHostMalloc(&h->h_idx, sizeof(int64_t) * cap, hipHostMallocMapped));
    S_HIP(h, hipHostMalloc(&h->h_score, sizeof(float) * cap, hipHostMallocMapped));
    S_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hd_idx), h->h_idx, 0));
    S_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->hd_score), h->h_score, 0));
    for (Shard& s : h->shards) {
        S_HIP(h, hipSetDevice(s.device));
        S_HIP(h, hipDeviceSynchronize());
        if (s.local_keys) (void)hipFree(s.local_keys);
        if (s.gathered) (void)hipFree(s.gathered);
        s.local_keys = nullptr;
        s.gathered = nullptr;
        S_HIP(h, hipMalloc(&s.local_keys, sizeof(mi355rec_key_t) * cap));
        S_HIP(h, hipMalloc(&s.gathered, sizeof(mi355rec_key_t) * cap * g));
    }
    h->cap = cap;
    return MI355REC_OK;
}

int ensure_rccl(mi355rec_sharded* h) {
    if (h->rccl_ready) return MI355REC_OK;
    {
        const int rc = drain_workers(h);
        if (rc) return rc;
    }
    std::string why;
    if (!g_rccl.load(why)) return sfail(h, MI355REC_ERR_HIP, "%s", why.c_str());
    const int g = static_cast<int>(h->shards.size());
    std::vector<int> devs(g);
    for (int r = 0; r < g; ++r) devs[r] = h->shards[r].device;
    for (int a = 0; a < g; ++a)
        for (int b = a + 1; b < g; ++b)
            if (devs[a] == devs[b])
                return sfail(h, MI355REC_ERR_INVALID_ARG,
                             "the RCCL transport needs one device per shard (device %d holds two)", devs[a]);
    std::vector<ncclComm_t> comms(g, nullptr);
    const ncclResult_t rc = g_rccl.CommInitAll(comms.data(), g, devs.data());
    if (rc != 0)
        return sfail(h, MI355REC_ERR_HIP, "ncclCommInitAll: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    for (int r = 0; r < g; ++r) h->shards[r].comm = comms[r];
    h->rccl_ready = true;
    return MI355REC_OK;
}

// ---- tasks: what a shard's worker (or, with one shard, the caller itself) executes ------------------

int run_task(mi355rec_sharded* h, int r, const Task& t) {
    Shard& s = h->shards[r];
    const int g = static_cast<int>(h->shards.size());
    switch (t.kind) {
        case kTaskStreamQuery:
            if (t.by_value) {
                S_ENG(h, s, mi355rec_enqueue_query_keys_streamed(s.engine, t.q, t.excl, t.topn, t.dst, s.stream));
            } else {
                S_ENG(h, s, mi355rec_enqueue_ptr_keys_streamed(s.engine, t.qptr, t.excl, t.topn, t.dst, s.stream));
            }
            return MI355REC_OK;
        case kTaskSyncQuery:
            if (t.count > 1) {
                S_ENG(h, s, mi355rec_enqueue_batch_keys(s.engine, t.queries, t.excls, t.count, t.topn, t.dst, s.stream));
            } else if (t.by_value) {
                S_ENG(h, s, mi355rec_enqueue_query_keys(s.engine, t.q, t.excl, t.topn, t.dst, s.stream));
            } else {
                S_ENG(h, s, mi355rec_enqueue_ptr_keys(s.engine, t.qptr, t.excl, t.topn, t.dst, nullptr, nullptr, s.stream));
            }
            return MI355REC_OK;
        case kTaskBatch:
            S_ENG(h, s, mi355rec_enqueue_batch_mixed_keys_streamed(s.engine, t.queries, t.qptrs, t.excls, t.count, t.topn, t.dst, s.stream));
            if (t.flush_after) S_ENG(h, s, mi355rec_enqueue_flush(s.engine, s.stream));
            return MI355REC_OK;
        case kTaskFlush:
            S_ENG(h, s, mi355rec_enqueue_flush(s.engine, s.stream));
            return MI355REC_OK;
        case kTaskPublish:
            S_HIP(h, hipEventRecord(s.done, s.stream));
            h->workers[r]->published.store(t.seq, std::memory_order_release);   // (several shards: there are workers)
            return MI355REC_OK;
        case kTaskAllGather: {
            const ncclResult_t nrc = g_rccl.AllGather(t.send, t.recv, t.stride, kNcclUint64, s.comm, s.stream);
            if (nrc != 0)
                return sfail(h, MI355REC_ERR_HIP, "ncclAllGather: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc) : "error");
            return MI355REC_OK;
        }
        case kTaskMerge: {
            if (t.seq != 0) {   // PEER transport (0: the collective has ordered the streams)
                // the other shards' events must have been RECORDED (on the host) before this stream can be told to
                // wait for them; their workers publish the exchange number right behind the record
                for (int o = 1; o < g; ++o) {
                    Shard& other = h->shards[o];
                    for (uint64_t spins = 0; h->workers[o]->published.load(std::memory_order_acquire) < t.seq; ++spins) {
                        if (h->workers[o]->err.load(std::memory_order_acquire) != 0)
                            return sfail(h, MI355REC_ERR_HIP, "shard on device %d failed before its keys were in place", other.device);
                        if ((spins & 63) == 63) std::this_thread::yield(); else __builtin_ia32_pause();
                    }
                    S_HIP(h, hipStreamWaitEvent(s.stream, other.done, 0));
                }
            }
            // list l of query b starts at b * topn + l * stride
            S_ENG(h, s, mi355rec_enqueue_merge_keys_batch(s.engine, t.lists, t.n_lists ? t.n_lists : g, t.topn, static_cast<int64_t>(t.stride),
                                                          static_cast<int64_t>(t.topn), t.count, t.topn, t.out_keys, t.out_idx,
                                                          t.out_score, s.stream));
            if (t.copy_back) {
                const size_t cnt = static_cast<size_t>(t.count) * t.topn;
                S_HIP(h, hipMemcpyAsync(h->h_idx, h->d_idx, cnt * sizeof(int64_t), hipMemcpyDeviceToHost, s.stream));
                S_HIP(h, hipMemcpyAsync(h->h_score, h->d_score, cnt * sizeof(float), hipMemcpyDeviceToHost, s.stream));
            }
            if (t.record_after) S_HIP(h, hipEventRecord(t.record_after, s.stream));
            return MI355REC_OK;
        }
    }
    return sfail(h, MI355REC_ERR_INVALID_ARG, "unknown task %d", t.kind);
}

void worker_main(mi355rec_sharded* h, int r) {
    Worker& w = *h->workers[r];
    t_worker_thread = true;
    (void)hipSetDevice(h->shards[r].device);
    uint64_t next = 0;
    for (;;) {
        int idle = 0;
        while (w.head.load(std::memory_order_acquire) == next) {
            if (w.stop.load(std::memory_order_acquire)) return;
            if (++idle < 400000) {
                __builtin_ia32_pause();
            } else {   // nothing for a few milliseconds: sleep until a task is posted (a wake-up costs tens of microseconds,
                       // which a lone query right after a burst should not pay)
                std::unique_lock<std::mutex> lk(w.mu);
                w.sleeping.store(true, std::memory_order_seq_cst);
                if (w.head.load(std::memory_order_seq_cst) == next && !w.stop.load(std::memory_order_seq_cst))
                    w.cv.wait_for(lk, std::chrono::milliseconds(5));
                w.sleeping.store(false, std::memory_order_seq_cst);
                idle = 0;
            }
        }
        const Task& t = w.ring[next % Worker::kCap];
        if (w.err.load(std::memory_order_relaxed) == 0) {   // after a failure the remaining tasks are dropped
            const int rc = run_task(h, r, t);
            if (rc != MI355REC_OK) {
                w.err_msg = g_sharded_error;   // sfail wrote the worker thread's copy
                w.err.store(rc, std::memory_order_release);
            }
        } else if (t.kind == kTaskPublish) {
            w.published.store(t.seq, std::memory_order_release);   // nobody must wait for a dead shard forever
        }
        ++next;
        w.tail.store(next, std::memory_order_release);
    }
}

// Hands a task to shard r: posted to its worker, or run here when there are no workers.  Returns the task's
// number on that worker (0 without workers) through *id.
int post(mi355rec_sharded* h, int r, const Task& t, uint64_t* id = nullptr) {
    if (h->workers.empty()) {
        S_HIP(h, hipSetDevice(h->shards[r].device));
        if (id) *id = 0;
        return run_task(h, r, t);
    }
    Worker& w = *h->workers[r];
    const uint64_t at = w.head.load(std::memory_order_relaxed);
    while (at - w.tail.load(std::memory_order_acquire) >= Worker::kCap) __builtin_ia32_pause();   // ring full: the worker is behind
    w.ring[at % Worker::kCap] = t;
    w.head.store(at + 1, std::memory_order_seq_cst);
    if (w.sleeping.load(std::memory_order_seq_cst)) {
        std::lock_guard<std::mutex> lk(w.mu);
        w.cv.notify_one();
    }
    if (id) *id = at + 1;
    return MI355REC_OK;
}

// First failure any worker has met (sticky: the handle is unusable afterwards, as after a HIP error).
int worker_error(mi355rec_sharded* h) {
    for (size_t r = 0; r < h->workers.size(); ++r) {
        const int rc = h->workers[r]->err.load(std::memory_order_acquire);
        if (rc != 0) return sfail(h, rc, "%s", h->workers[r]->err_msg.c_str());
    }
    return MI355REC_OK;
}

// Waits until shard r's worker has run its first `id` tasks (all of them when id == 0).
int wait_worker(mi355rec_sharded* h, int r, uint64_t id = 0) {
    if (h->workers.empty()) return MI355REC_OK;
    Worker& w = *h->workers[r];
    const uint64_t want = id ? id : w.head.load(std::memory_order_relaxed);
    for (uint64_t spins = 0; w.tail.load(std::memory_order_acquire) < want; ++spins) {
        if ((spins & 1023) == 1023) std::this_thread::yield(); else __builtin_ia32_pause();
    }
    return worker_error(h);
}

// Every worker idle: what anything that touches the engines from the caller's thread does first.
int drain_workers(mi355rec_sharded* h) {
    for (size_t r = 0; r < h->workers.size(); ++r) {
        const int rc = wait_worker(h, static_cast<int>(r));
        if (rc) return rc;
    }
    return MI355REC_OK;
}

void stop_workers(mi355rec_sharded* h) {
    for (auto& w : h->workers) {
        w->stop.store(true, std::memory_order_seq_cst);
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->cv.notify_one();
        }
        if (w->th.joinable()) w->th.join();
    }
    h->workers.clear();
}

// The exchange + final merge of `count` queries whose per-shard key lists ([shard][query][key],
// `stride` keys from one shard's block to the next) are being produced on the shards' streams:
// PEER: lists already land in `peer_lists` on the first device; one event per shard orders the
// merge behind them.  RCCL: one ncclAllGather of `stride` keys per shard from send_of(r) into
// recv_of(r), each issued by its shard's worker.  Results go to out_keys (device) and out_idx /
// out_score (device-visible addresses; mapped host memory on the hot paths).  Returns the merge's
// task number on the first device's worker through *merge_task.
template <typename SendOf, typename RecvOf>
int exchange_and_merge(mi355rec_sharded* h, bool rccl, const mi355rec_key_t* peer_lists, SendOf send_of, RecvOf recv_of,
                       size_t stride, int count, int topn, mi355rec_key_t* out_keys, int64_t* out_idx, float* out_score,
                       hipEvent_t record_after, bool copy_back, uint64_t* merge_task) {
    const int g = static_cast<int>(h->shards.size());
    const uint64_t seq = ++h->exchange_seq;
    Task m;
    m.kind = kTaskMerge;
    m.seq = rccl ? 0 : seq;   // 0: nothing to wait for on the host (the collective orders the streams)
    m.stride = stride;
    m.count = count;
    m.topn = topn;
    m.out_keys = out_keys;
    m.out_idx = out_idx;
    m.out_score = out_score;
    m.record_after = record_after;
    m.copy_back = copy_back;
    if (rccl) {
        for (int r = 0; r < g; ++r) {   // every rank's call from its own thread (one rank: from here)
            Task a;
            a.kind = kTaskAllGather;
            a.send = send_of(r);
            a.recv = recv_of(r);
            a.stride = stride;
            const int rc = post(h, r, a);
            if (rc) return rc;
        }
        m.lists = recv_of(0);   // the collective is ordered on the first device's stream already
    } else {
        for (int r = 1; r < g; ++r) {
            Task pb;
            pb.kind = kTaskPublish;
            pb.seq = seq;
            const int rc = post(h, r, pb);
            if (rc) return rc;
        }
        m.lists = peer_lists;
    }
    return post(h, 0, m, merge_task);
}

// `count` queries (host vectors; or, with count == 1, `qptr` = where the query's 12 floats live
// in device memory every shard can read) -> merged results in the pinned host mirrors.
// ONE device wait: the first device's stream.  Its merge waited (events / the collective) for
// everything the other shards did for this call, and their streams order the next call's
// writes behind this call's reads, so nothing else needs draining.
int run_queries(mi355rec_sharded* h, const float* queries, const float* qptr, const int64_t* exclude, int count, int topn) {
    const int g = static_cast<int>(h->shards.size());
    const size_t per_shard = static_cast<size_t>(count) * topn;
    int rc = ensure_capacity(h, per_shard);
    if (rc) return rc;
    const bool rccl = h->transport == MI355REC_TRANSPORT_RCCL;
    if (rccl && (rc = ensure_rccl(h)) != MI355REC_OK) return rc;

    // every shard: scan + local merge on its own stream; keys land either directly in
    // device 0's gather buffer (peer stores) or in the shard's send buffer (RCCL)
    for (int r = 0; r < g; ++r) {
        Shard& s = h->shards[r];
        Task t;
        t.kind = kTaskSyncQuery;
        t.count = count;
        t.topn = topn;
        t.dst = rccl ? s.local_keys : h->gather0 + static_cast<size_t>(r) * per_shard;
        t.excl = exclude ? exclude[0] : -1;
        t.excls = exclude;
        t.queries = queries;
        if (qptr) {
            t.qptr = qptr;
        } else if (count == 1) {
            t.by_value = true;
            std::memcpy(t.q, queries, sizeof t.q);
        }
        rc = post(h, r, t);
        if (rc) return rc;
    }
    // results up to a few thousand slots are stored by the merge kernel straight into mapped host
    // memory (no copy launches on the latency path); larger ones come back in two copies
    const bool direct = per_shard <= 4096;
    uint64_t merge_task = 0;
    rc = exchange_and_merge(
        h, rccl, h->gather0, [&](int r) { return h->shards[r].local_keys; }, [&](int r) { return h->shards[r].gathered; },
        per_shard, count, topn, h->d_keys, direct ? h->hd_idx : h->d_idx, direct ? h->hd_score : h->d_score, nullptr, !direct,
        &merge_task);
    if (rc) return rc;
    rc = wait_worker(h, 0, merge_task);   // the merge has been ENQUEUED on the first device's stream ...
    if (rc) return rc;
    Shard& root = h->shards[0];
    S_HIP(h, hipSetDevice(root.device));
    S_HIP(h, hipStreamSynchronize(root.stream));   // ... and now it has run
    return worker_error(h);
}

// topn above the single-launch merge limit (the CLI's `-n 5000`): every shard serves the
// query in rounds of 1024 (as a single engine does), the G sorted key lists come back
// to the host and are merged there — pure key ordering, no arithmetic on scores.  Cold
// path, one query at a time.
int run_queries_large(mi355rec_sharded* h, const float* queries, const int64_t* exclude, int count, int eff, int topn,
                      int64_t* out_idx, float* out_score, int* out_count) {
    const int g = static_cast<int>(h->shards.size());
    int rc = drain_workers(h);   // cold path: the caller's thread drives every shard itself
    if (rc) return rc;
    rc = ensure_capacity(h, static_cast<size_t>(eff));
    if (rc) return rc;
    std::vector<mi355rec_key_t> all(static_cast<size_t>(g) * eff);
    for (int b = 0; b < count; ++b) {
        for (int r = 0; r < g; ++r) {
            Shard& s = h->shards[r];
            S_HIP(h, hipSetDevice(s.device));
            S_ENG(h, s, mi355rec_enqueue_query_keys(s.engine, queries + static_cast<size_t>(b) * MI355REC_DIM,
                                                    exclude ? exclude[b] : -1, eff, s.local_keys, s.stream));
            S_HIP(h, hipMemcpyAsync(all.data() + static_cast<size_t>(r) * eff, s.local_keys, sizeof(mi355rec_key_t) * eff,
                                    hipMemcpyDeviceToHost, s.stream));
        }
        for (int r = 0; r < g; ++r) {
            S_HIP(h, hipSetDevice(h->shards[r].device));
            S_HIP(h, hipStreamSynchronize(h->shards[r].stream));
        }
        std::sort(all.begin(), all.end(), [](mi355rec_key_t a, mi355rec_key_t b2) { return a > b2; });
        int c = 0;
        for (int i = 0; i < topn; ++i) {
            const mi355rec_key_t k = i < eff ? all[i] : 0;
            out_idx[static_cast<size_t>(b) * topn + i] = mi355rec_key_row(k);
            if (out_score) out_score[static_cast<size_t>(b) * topn + i] = mi355rec_key_score(k);
            if (k) ++c;
        }
        if (out_count) out_count[b] = c;
    }
    return MI355REC_OK;
}

void copy_rows(const int64_t* src_i, const float* src_s, int count, int eff, int topn, int64_t* out_idx, float* out_score,
               int* out_count) {
    for (int b = 0; b < count; ++b) {
        const int64_t* si = src_i + static_cast<size_t>(b) * eff;
        const float* ss = src_s + static_cast<size_t>(b) * eff;
        int64_t* dst_i = out_idx + static_cast<size_t>(b) * topn;
        std::memcpy(dst_i, si, sizeof(int64_t) * eff);
        for (int i = eff; i < topn; ++i) dst_i[i] = -1;
        if (out_score) {
            float* dst_s = out_score + static_cast<size_t>(b) * topn;
            std::memcpy(dst_s, ss, sizeof(float) * eff);
            for (int i = eff; i < topn; ++i) dst_s[i] = 0.0f;
        }
        if (out_count) {
            int c = 0;
            while (c < eff && si[c] >= 0) ++c;
            out_count[b] = c;
        }
    }
}

struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

const Shard* owner_of(const mi355rec_sharded* h, int64_t global_row) {
    for (const Shard& s : h->shards)
        if (global_row >= s.lo && global_row < s.hi) return &s;
    return nullptr;
}

// Where the 12 floats of a catalogue row live, for kernels on ANY shard's device; null when
// the row has to travel by value (no all-pairs peer access): then *q_host receives it.
int locate_row(mi355rec_sharded* h, int64_t global_row, const float** qptr, float* q_host) {
    const Shard* own = owner_of(h, global_row);
    if (!own) return sfail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)global_row);
    *qptr = nullptr;
    if (h->peer_rows) {
        const int rc = mi355rec_row_ptr(own->engine, global_row - own->lo, qptr);
        if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", own->device, mi355rec_last_error(own->engine));
        return MI355REC_OK;
    }
    const int rc = mi355rec_fetch_row(own->engine, global_row - own->lo, q_host);
    if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", own->device, mi355rec_last_error(own->engine));
    return MI355REC_OK;
}

// Replicated placement, synchronous calls: the replicas take turns; the caller's thread drives the chosen replica's
// handle itself once that replica's worker is idle (a handle is used by one thread at a time).
int take_replica(mi355rec_sharded* h, Shard** out) {
    const int r = h->next_replica;
    h->next_replica = (r + 1) % static_cast<int>(h->shards.size());
    const int rc = wait_worker(h, r);
    if (rc) return rc;
    S_HIP(h, hipSetDevice(h->shards[r].device));
    *out = &h->shards[r];
    return MI355REC_OK;
}

// ---- the stream of single queries ------------------------------------------------------

void free_stream(mi355rec_sharded* h) {
    if (h->shards.empty()) return;
    for (Shard& s : h->shards) {
        if (hipSetDevice(s.device) != hipSuccess) continue;
        if (s.s_local) (void)hipFree(s.s_local);
        if (s.s_gathered) (void)hipFree(s.s_gathered);
        s.s_local = s.s_gathered = nullptr;
    }
    if (hipSetDevice(h->shards[0].device) == hipSuccess) {
        if (h->s_gather0) (void)hipFree(h->s_gather0);
        if (h->s_keys) (void)hipFree(h->s_keys);
        if (h->s_hidx) (void)hipHostFree(h->s_hidx);
        if (h->s_hscore) (void)hipHostFree(h->s_hscore);
    }
    h->s_gather0 = h->s_keys = nullptr;
    h->s_hidx = nullptr;
    h->s_hscore = nullptr;
    h->s_topn = 0;
    h->s_alloc_window = 0;
}

int stream_alloc(mi355rec_sharded* h, int topn) {
    const int g = static_cast<int>(h->shards.size());
    const size_t wk = static_cast<size_t>(h->s_window) * topn;   // keys of one shard in one window
    S_HIP(h, hipSetDevice(h->shards[0].device));
    if (!h->replicated) {
        S_HIP(h, hipMalloc(&h->s_gather0, sizeof(mi355rec_key_t) * kStreamDepth * g * wk));
        S_HIP(h, hipMalloc(&h->s_keys, sizeof(mi355rec_key_t) * kStreamDepth * wk));
    }
    // (portable: in the replicated placement every replica's kernels store their windows' results here)
    S_HIP(h, hipHostMalloc(&h->s_hidx, sizeof(int64_t) * kStreamDepth * wk, hipHostMallocMapped | hipHostMallocPortable));
    S_HIP(h, hipHostMalloc(&h->s_hscore, sizeof(float) * kStreamDepth * wk, hipHostMallocMapped | hipHostMallocPortable));
    S_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->s_hdidx), h->s_hidx, 0));
    S_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&h->s_hdscore), h->s_hscore, 0));
    for (Shard& s : h->shards) {
        S_HIP(h, hipSetDevice(s.device));
        S_HIP(h, hipMalloc(&s.s_local, sizeof(mi355rec_key_t) * kStreamDepth * wk));
        // sharded: the all-gather's receive buffer; replicated: the window's unpacked keys
        S_HIP(h, hipMalloc(&s.s_gathered, sizeof(mi355rec_key_t) * kStreamDepth * (h->replicated ? 1 : g) * wk));
        S_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&s.s_hdidx), h->s_hidx, 0));
        S_HIP(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&s.s_hdscore), h->s_hscore, 0));
    }
    h->s_topn = topn;
    h->s_alloc_window = h->s_window;
    h->s_batched = h->s_window >= 2 && h->batched_windows;
    for (const Shard& s : h->shards)
        if (s.hi > s.lo && !mi355rec_batch_pointers_ok(s.engine, topn)) h->s_batched = false;
    const size_t slots = static_cast<size_t>(kStreamDepth) * h->s_window;
    h->w_q.assign(slots * MI355REC_DIM, 0.0f);
    h->w_ptr.assign(slots, nullptr);
    h->w_excl.assign(slots, -1);
    return MI355REC_OK;
}

int stream_flush(mi355rec_sharded* h);

// Waits until ring entry w's results are in host memory (its merge has been enqueued, then has run).
int wait_window(mi355rec_sharded* h, int w) {
    Window& win = h->win[w];
    const int owner = h->replicated ? win.owner : 0;
    const int rc = wait_worker(h, owner, win.merge_task);   // `merged` has been recorded ...
    if (rc) return rc;
    hipEvent_t ev = h->replicated ? h->r_merged[static_cast<size_t>(w) * h->shards.size() + owner] : win.merged;
    S_HIP(h, hipSetDevice(h->shards[owner].device));
    S_HIP(h, hipEventSynchronize(ev));                       // ... and has happened
    return MI355REC_OK;
}

// Buffers for (topn, window); a change of geometry closes the stream first.  All or nothing.
int ensure_stream(mi355rec_sharded* h, int topn) {
    if (h->s_topn == topn && h->s_alloc_window == h->s_window) return MI355REC_OK;
    if (h->s_topn) {
        int rc = stream_flush(h);
        if (rc) return rc;
        rc = drain_workers(h);
        if (rc) return rc;
        for (Shard& s : h->shards) {
            S_HIP(h, hipSetDevice(s.device));
            S_HIP(h, hipStreamSynchronize(s.stream));
        }
        free_stream(h);
    }
    for (Window& w : h->win) {
        w.abs = -1;
        w.count = 0;
        w.handed = false;
        w.issued = false;
    }
    // tickets keep growing across a change of geometry, window-aligned in the new one
    h->next_ticket = (h->next_ticket + h->s_window - 1) / h->s_window * h->s_window;
    h->issued_upto = h->next_ticket;
    const int rc = stream_alloc(h, topn);
    if (rc != MI355REC_OK) free_stream(h);
    return rc;
}

// Enqueues the exchange + batched merge of ring entry `w` (its `count` queries are complete on
// every shard's stream, in stream order).
int stream_issue(mi355rec_sharded* h, int w) {
    Window& win = h->win[w];
    const int g = static_cast<int>(h->shards.size());
    const int topn = h->s_topn;
    const size_t wk = static_cast<size_t>(h->s_window) * topn;
    if (h->replicated) {
        // no exchange: the window's owner unpacks its own key lists (a "merge" of one list per query) straight into
        // the pinned result ring and records the window's event on its own stream
        Shard& own = h->shards[win.owner];
        Task m;
        m.kind = kTaskMerge;
        m.seq = 0;
        m.n_lists = 1;
        m.lists = own.s_local + static_cast<size_t>(w) * wk;
        m.stride = wk;
        m.count = win.count;
        m.topn = topn;
        m.out_keys = own.s_gathered + static_cast<size_t>(w) * wk;
        m.out_idx = own.s_hdidx + static_cast<size_t>(w) * wk;
        m.out_score = own.s_hdscore + static_cast<size_t>(w) * wk;
        m.record_after = h->r_merged[static_cast<size_t>(w) * g + win.owner];
        const int prc = post(h, win.owner, m, &win.merge_task);
        if (prc) return prc;
        win.issued = true;
        return MI355REC_OK;
    }
    const bool rccl = h->transport == MI355REC_TRANSPORT_RCCL;
    const int rc = exchange_and_merge(
        h, rccl, h->s_gather0 + static_cast<size_t>(w) * g * wk,
        [&](int r) { return h->shards[r].s_local + static_cast<size_t>(w) * wk; },
        [&](int r) { return h->shards[r].s_gathered + static_cast<size_t>(w) * g * wk; }, wk, win.count, topn,
        h->s_keys + static_cast<size_t>(w) * wk, h->s_hdidx + static_cast<size_t>(w) * wk, h->s_hdscore + static_cast<size_t>(w) * wk,
        win.merged, false, &win.merge_task);
    if (rc) return rc;
    win.issued = true;
    ++h->st_exchanges;
    return MI355REC_OK;
}

// Windows that have become complete (every query of theirs is at least kStreamLag calls old) get
// their exchange now; `all`: whatever is open as well (the caller has drained the shard pipelines).
int stream_issue_ready(mi355rec_sharded* h, bool all) {
    const int W = h->s_window;
    while (h->issued_upto < h->next_ticket) {
        const int64_t first = h->issued_upto;
        const int64_t end = first + W;   // windows are ticket-aligned
        if (!all && end + kStreamLag > h->next_ticket) break;
        const int w = static_cast<int>((first / W) % kStreamDepth);
        const int rc = stream_issue(h, w);
        if (rc) return rc;
        h->issued_upto = end < h->next_ticket || !all ? end : h->next_ticket;
    }
    return MI355REC_OK;
}

int stream_issue_batched(mi355rec_sharded* h, int w, bool close_all);

int stream_flush(mi355rec_sharded* h) {
    if (!h->s_topn || h->issued_upto >= h->next_ticket) return MI355REC_OK;
    if (h->s_batched) {   // the open window goes out as it is, the shards' pipelines are drained, every exchange issued
        const int W = h->s_window;
        const int64_t last = (h->next_ticket - 1) / W;   // the newest window that holds a query
        Window& win = h->win[static_cast<int>(last % kStreamDepth)];
        if (win.abs != last) return sfail(h, MI355REC_ERR_HIP, "stream bookkeeping: window %lld is not in the ring", (long long)last);
        const int rc = stream_issue_batched(h, static_cast<int>(last % kStreamDepth), true);
        if (rc) return rc;
        h->next_ticket = (h->next_ticket + W - 1) / W * W;
        h->issued_upto = h->next_ticket;
        return MI355REC_OK;
    }
    if (h->replicated) {   // only the open window is outstanding (a full one was closed by its last query): its owner drains
        const int W = h->s_window;
        const int64_t last = (h->next_ticket - 1) / W;
        const int w = static_cast<int>(last % kStreamDepth);
        Task t;
        t.kind = kTaskFlush;
        int rc = post(h, h->win[w].owner, t);
        if (rc) return rc;
        rc = stream_issue(h, w);
        if (rc) return rc;
        h->next_ticket = (h->next_ticket + W - 1) / W * W;
        h->issued_upto = h->next_ticket;
        return MI355REC_OK;
    }
    for (int r = 0; r < static_cast<int>(h->shards.size()); ++r) {
        Task t;
        t.kind = kTaskFlush;
        const int prc = post(h, r, t);
        if (prc) return prc;
    }
    const int rc = stream_issue_ready(h, true);
    if (rc) return rc;
    const int W = h->s_window;
    h->next_ticket = (h->next_ticket + W - 1) / W * W;   // the next query opens a new window
    h->issued_upto = h->next_ticket;
    return MI355REC_OK;
}

// A collected window goes to every shard in ONE call: a streamed batch (multi-query passes over the replica
// whose merge rides in the shard's next launch), so its keys are complete kWindowLag windows later — or at
// the flush, which drains the shards' pipelines.  `close_all`: issue the exchange of every window handed out.
int stream_issue_batched(mi355rec_sharded* h, int w, bool close_all) {
    Window& win = h->win[w];
    if (win.count == 0 || win.issued) return MI355REC_OK;
    const int g = static_cast<int>(h->shards.size());
    const int W = h->s_window, topn = h->s_topn;
    const size_t wk = static_cast<size_t>(W) * topn;
    const bool rccl = h->transport == MI355REC_TRANSPORT_RCCL;
    const size_t at = static_cast<size_t>(w) * W;
    bool any_ptr = false, any_vec = false;
    for (int i = 0; i < win.count; ++i) (h->w_ptr[at + i] ? any_ptr : any_vec) = true;
    if (h->replicated) {   // the whole window to its owner, drained behind it; nothing to exchange
        Shard& own = h->shards[win.owner];
        Task t;
        t.kind = kTaskBatch;
        t.queries = any_vec ? &h->w_q[at * MI355REC_DIM] : nullptr;
        t.qptrs = any_ptr ? &h->w_ptr[at] : nullptr;
        t.excls = &h->w_excl[at];
        t.count = win.count;
        t.topn = topn;
        t.dst = own.s_local + static_cast<size_t>(w) * wk;
        t.flush_after = true;
        const int prc = post(h, win.owner, t);
        if (prc) return prc;
        win.handed = true;
        (void)close_all;
        return stream_issue(h, w);
    }
    for (int r = 0; r < g; ++r) {
        Shard& s = h->shards[r];
        Task t;
        if (!win.handed) {
            t.kind = kTaskBatch;
            t.queries = any_vec ? &h->w_q[at * MI355REC_DIM] : nullptr;   // the ring entry stays untouched until its window
            t.qptrs = any_ptr ? &h->w_ptr[at] : nullptr;                   // has been merged (back-pressure in stream_enqueue)
            t.excls = &h->w_excl[at];
            t.count = win.count;
            t.topn = topn;
            t.dst = rccl ? s.s_local + static_cast<size_t>(w) * wk : h->s_gather0 + (static_cast<size_t>(w) * g + r) * wk;
            t.flush_after = close_all;
        } else if (close_all) {
            t.kind = kTaskFlush;
        } else {
            continue;
        }
        const int prc = post(h, r, t);
        if (prc) return prc;
    }
    win.handed = true;
    // exchanges, oldest first: everything at least kWindowLag windows old — or everything, behind a flush
    for (int64_t a = win.abs - (kStreamDepth - 1); a <= win.abs; ++a) {
        if (a < 0) continue;
        Window& old = h->win[static_cast<int>(a % kStreamDepth)];
        if (old.abs != a || !old.handed || old.issued) continue;
        if (!close_all && a + kWindowLag > win.abs) continue;
        const int rc = stream_issue(h, static_cast<int>(a % kStreamDepth));
        if (rc) return rc;
    }
    return MI355REC_OK;
}

// `row` >= 0: the query is that catalogue row and qptr / query12 already locate it for a SHARDED handle; a replicated one
// reads the row from the replica that serves the window.
int stream_enqueue(mi355rec_sharded* h, const float* qptr, const float* query12, int64_t exclude_global, int topn,
                   int64_t* ticket, int64_t row = -1) {
    if (topn <= 0 || topn > MI355REC_MAX_TOPN_FAST)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "topn must be in [1, %d] for streamed queries, got %d", MI355REC_MAX_TOPN_FAST, topn);
    const int64_t t0 = now_ns();
    int rc = ensure_stream(h, topn);
    if (rc) return rc;
    const bool rccl = !h->replicated && h->transport == MI355REC_TRANSPORT_RCCL;
    if (rccl && (rc = ensure_rccl(h)) != MI355REC_OK) return rc;
    const int g = static_cast<int>(h->shards.size());
    const int W = h->s_window;
    const int64_t t = h->next_ticket;
    const int64_t abs = t / W;
    const int w = static_cast<int>(abs % kStreamDepth);
    const int slot = static_cast<int>(t % W);
    Window& win = h->win[w];
    if (slot == 0) {
        // The ring entry's previous window (kStreamDepth windows ago) must be done on the device before
        // any shard writes into its buffers again: host back-pressure, normally long satisfied.
        if (win.abs >= 0 && win.issued) {
            rc = wait_window(h, w);
            if (rc) return rc;
        }
        win.abs = abs;
        win.count = 0;
        win.handed = false;
        win.issued = false;
        win.owner = h->replicated ? static_cast<int>(abs % g) : 0;   // whole windows are dealt round-robin
    }
    const size_t wk = static_cast<size_t>(W) * topn;
    if (h->replicated && row >= 0) {   // every replica holds the row: the window's owner reads its own copy
        const Shard& own = h->shards[win.owner];
        const int prc = mi355rec_row_ptr(own.engine, row, &qptr);
        if (prc != MI355REC_OK) return sfail(h, prc, "replica on device %d: %s", own.device, mi355rec_last_error(own.engine));
    }
    if (h->s_batched) {
        const size_t at = static_cast<size_t>(w) * W + slot;
        h->w_ptr[at] = qptr;
        if (!qptr) std::memcpy(&h->w_q[at * MI355REC_DIM], query12, sizeof(float) * MI355REC_DIM);
        h->w_excl[at] = exclude_global;
        ++win.count;
        ++h->next_ticket;
        ++h->st_queries;
        if (ticket) *ticket = t;
        if (slot == W - 1) {
            rc = stream_issue_batched(h, w, false);
            if (h->replicated) h->issued_upto = h->next_ticket;   // (closed and issued at once: nothing lags behind)
        }
        h->st_host_ns += now_ns() - t0;
        return rc;
    }
    for (int r = 0; r < g; ++r) {
        if (h->replicated && r != win.owner) continue;   // one replica serves the whole window
        Shard& s = h->shards[r];
        Task t;
        t.kind = kTaskStreamQuery;
        t.topn = topn;
        t.excl = exclude_global;
        t.dst = (rccl || h->replicated) ? s.s_local + static_cast<size_t>(w) * wk + static_cast<size_t>(slot) * topn
                                        : h->s_gather0 + (static_cast<size_t>(w) * g + r) * wk + static_cast<size_t>(slot) * topn;
        if (qptr) {
            t.qptr = qptr;
        } else {
            t.by_value = true;
            std::memcpy(t.q, query12, sizeof t.q);
        }
        rc = post(h, r, t);
        if (rc) return rc;
    }
    ++win.count;
    ++h->next_ticket;
    ++h->st_queries;
    if (ticket) *ticket = t;
    if (h->replicated) {
        if (slot == W - 1) {   // the window is full: its owner drains its pipeline and unpacks the results
            Task f;
            f.kind = kTaskFlush;
            rc = post(h, win.owner, f);
            if (!rc) rc = stream_issue(h, w);
            h->issued_upto = h->next_ticket;
        }
    } else {
        rc = stream_issue_ready(h, false);
    }
    h->st_host_ns += now_ns() - t0;
    return rc;
}

}  // namespace

extern "C" {

const char* mi355rec_sharded_last_error(const mi355rec_sharded_t* h) {
    return h ? h->err.c_str() : g_sharded_error.c_str();
}

void mi355rec_sharded_destroy(mi355rec_sharded_t* h) {
    if (!h) return;
    if (h->cpu) {
        mi355cpu::node_destroy(h->cpu);
        delete h;
        return;
    }
    DeviceRestore restore;
    stop_workers(h);   // they finish what has been posted first
    for (Shard& s : h->shards) {
        if (hipSetDevice(s.device) != hipSuccess) continue;
        if (s.stream) (void)hipStreamSynchronize(s.stream);
    }
    free_stream(h);
    for (size_t i = 0; i < h->r_merged.size(); ++i) {
        if (!h->r_merged[i]) continue;
        if (hipSetDevice(h->shards[i % h->shards.size()].device) == hipSuccess) (void)hipEventDestroy(h->r_merged[i]);
    }
    for (Shard& s : h->shards) {
        if (hipSetDevice(s.device) != hipSuccess) continue;
        if (s.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s.comm);
        if (s.engine) mi355rec_destroy(s.engine);
        if (s.local_keys) (void)hipFree(s.local_keys);
        if (s.gathered) (void)hipFree(s.gathered);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    if (!h->shards.empty() && hipSetDevice(h->shards[0].device) == hipSuccess) {
        if (h->gather0) (void)hipFree(h->gather0);
        if (h->d_keys) (void)hipFree(h->d_keys);
        if (h->d_idx) (void)hipFree(h->d_idx);
        if (h->d_score) (void)hipFree(h->d_score);
        if (h->h_idx) (void)hipHostFree(h->h_idx);
        if (h->h_score) (void)hipHostFree(h->h_score);
        for (Window& w : h->win)
            if (w.merged) (void)hipEventDestroy(w.merged);
    }
    delete h;
}

}  // extern "C"

namespace {

// Shard r of a SHARDED handle holds its balanced block of rows; of a REPLICATED one, all of them.
int create_on(const float* feats_host, int64_t n, int dim, const int* devices, int n_shards, bool replicated,
              mi355rec_sharded_t** out) {
    if (out) *out = nullptr;
    if (!out || !feats_host || !devices) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null argument");
    if (dim != MI355REC_DIM) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "dim must be %d, got %d", MI355REC_DIM, dim);
    if (n < 1 || n > 0xfffffffell) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "rows %lld out of range", (long long)n);
    if (n_shards < 1 || n_shards > MI355REC_MAX_SHARDS)
        return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "n_shards must be in [1, %d], got %d", MI355REC_MAX_SHARDS, n_shards);
    const int visible = mi355rec_device_count();
    if (visible <= 0)
        return sfail(nullptr, MI355REC_ERR_NO_DEVICE, "no HIP device visible: the MI355X engine has no CPU fallback");
    for (int r = 0; r < n_shards; ++r)
        if (devices[r] < 0 || devices[r] >= visible)
            return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "device %d not in [0,%d)", devices[r], visible);

    DeviceRestore restore;
    mi355rec_sharded* h = new mi355rec_sharded();
    h->n = n;
    h->replicated = replicated;
    h->shards.resize(n_shards);
    auto bail = [&](int code) {
        const std::string keep = g_sharded_error;
        mi355rec_sharded_destroy(h);
        g_sharded_error = keep;
        return code;
    };
    const int root = devices[0];
    for (int r = 0; r < n_shards; ++r) {
        Shard& s = h->shards[r];
        s.device = devices[r];
        if (replicated) {
            s.lo = 0;
            s.hi = n;
        } else {
            bounds(n, n_shards, r, s.lo, s.hi);
        }
        if (hipSetDevice(s.device) != hipSuccess) return bail(sfail(nullptr, MI355REC_ERR_HIP, "hipSetDevice(%d) failed", s.device));
        const int rc = mi355rec_create(s.hi > s.lo ? feats_host + s.lo * MI355REC_DIM : nullptr, s.hi - s.lo, dim, s.device,
                                       s.lo, &s.engine);
        if (rc != MI355REC_OK) return bail(sfail(nullptr, rc, "shard %d on device %d: %s", r, s.device, mi355rec_last_global_error()));
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)
            return bail(sfail(nullptr, MI355REC_ERR_HIP, "stream / event creation on device %d failed", s.device));
    }
    // Peer mappings.  Stores into the first device's gather buffers need ITS memory mapped on every
    // other device (PEER transport); queries by row need every device's rows mapped on every other.
    // (Replicas never read or write each other's memory.)
    for (int a = 0; a < n_shards && !replicated; ++a) {
        for (int b = 0; b < n_shards; ++b) {
            const int da = devices[a], db = devices[b];
            if (da == db) continue;
            bool ok = false;
            int can = 0;
            if (hipSetDevice(da) == hipSuccess && hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
                const hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
            }
            (void)hipGetLastError();
            if (!ok) {
                h->peer_rows = false;
                if (db == root) h->peer_ok = false;
            }
        }
    }
    if (hipSetDevice(root) != hipSuccess) return bail(sfail(nullptr, MI355REC_ERR_HIP, "hipSetDevice(%d) failed", root));
    for (Window& w : h->win)
        if (hipEventCreateWithFlags(&w.merged, hipEventDisableTiming) != hipSuccess)
            return bail(sfail(nullptr, MI355REC_ERR_HIP, "event creation on device %d failed", root));
    h->transport = h->peer_ok ? MI355REC_TRANSPORT_PEER : MI355REC_TRANSPORT_RCCL;
    if (replicated) {   // one "results are in host memory" event per (ring entry, replica), on the replica's device
        h->r_merged.assign(static_cast<size_t>(kStreamDepth) * n_shards, nullptr);
        for (int w = 0; w < kStreamDepth; ++w)
            for (int r = 0; r < n_shards; ++r)
                if (hipSetDevice(devices[r]) != hipSuccess ||
                    hipEventCreateWithFlags(&h->r_merged[static_cast<size_t>(w) * n_shards + r], hipEventDisableTiming) != hipSuccess)
                    return bail(sfail(nullptr, MI355REC_ERR_HIP, "event creation on device %d failed", devices[r]));
    }

    if (n_shards > 1) {   // one worker per shard (none for a single shard: its calls are made by the caller)
        for (int r = 0; r < n_shards; ++r) {
            h->workers.emplace_back(new Worker());
            h->workers.back()->ring.resize(Worker::kCap);
        }
        for (int r = 0; r < n_shards; ++r) h->workers[r]->th = std::thread(worker_main, h, r);
    }

    // Queries by row through the peer mapping have to give the by-value path's keys: one query
    // per shard boundary, checked here, once, on real multi-device placements (a mismatch or an
    // error switches the pointer path off for this handle; mi355rec_sharded_info's note says so).
    bool distinct = false;
    for (int r = 1; r < n_shards; ++r) distinct = distinct || devices[r] != devices[0];
    if (replicated) {
        // nothing to check: a replica reads its own rows
    } else if (distinct && h->peer_rows && n >= 2) {
        const int topn = n - 1 < 16 ? static_cast<int>(n - 1) : 16;
        std::vector<int64_t> i_ptr(topn), i_val(topn);
        std::vector<float> s_ptr(topn), s_val(topn);
        for (int r = 0; r < n_shards && h->peer_rows; ++r) {
            const Shard& s = h->shards[r];
            if (s.hi == s.lo) continue;
            const int64_t row = s.hi - 1;
            int c0 = 0, c1 = 0;
            int rc = mi355rec_sharded_query_row_topn(h, row, topn, i_ptr.data(), s_ptr.data(), &c0);
            h->peer_rows = false;
            const int rc2 = mi355rec_sharded_query_row_topn(h, row, topn, i_val.data(), s_val.data(), &c1);
            h->peer_rows = true;
            if (rc2 != MI355REC_OK) return bail(sfail(nullptr, rc2, "self-check query failed: %s", h->err.c_str()));
            if (rc != MI355REC_OK || c0 != c1 || i_ptr != i_val ||
                std::memcmp(s_ptr.data(), s_val.data(), sizeof(float) * topn) != 0) {
                h->peer_rows = false;
                h->note = "queries by row travel by value: reading a row through the peer mapping did not reproduce the by-value result";
            }
        }
    } else if (!h->peer_rows) {
        h->note = "queries by row travel by value: no all-pairs peer access between the shards' devices";
    }
    *out = h;
    return MI355REC_OK;
}

}  // namespace

extern "C" {

int mi355rec_create_sharded_on(const float* feats_host, int64_t n, int dim, const int* devices, int n_shards,
                               mi355rec_sharded_t** out) {
    return create_on(feats_host, n, dim, devices, n_shards, false, out);
}

// How many devices a row-sharded catalogue of n rows is spread over when the caller does not say (see "PLACEMENT"
// at the top): as many as keep at least kRowsPerShardAuto rows per shard — a smaller shard is launch-bound, and
// every further shard adds to the exchange.
constexpr int64_t kRowsPerShardAuto = 4000000;

int mi355rec_auto_shards(int64_t n, int visible_devices) {
    if (visible_devices < 1) return 0;
    int64_t g = n / kRowsPerShardAuto;
    if (g < 1) g = 1;
    if (g > visible_devices) g = visible_devices;
    if (g > MI355REC_MAX_SHARDS) g = MI355REC_MAX_SHARDS;
    return static_cast<int>(g);
}

int mi355rec_create_placed(const float* feats_host, int64_t n, int dim, const int* devices, int n_devices, int placement,
                           mi355rec_sharded_t** out) {
    if (out) *out = nullptr;
    if (placement != MI355REC_PLACEMENT_AUTO && placement != MI355REC_PLACEMENT_SHARDED && placement != MI355REC_PLACEMENT_REPLICATED)
        return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "unknown placement %d", placement);
    const int visible = mi355rec_device_count();
    if (visible <= 0) {
        // No device at all.  An explicit device list cannot be honoured; otherwise the catalogue is served by the
        // product's own CPU backend, as the reference falls back to its CPU loop (Recommender.cu:117-127,176-181).
        if (devices || n_devices > 0)
            return sfail(nullptr, MI355REC_ERR_NO_DEVICE, "no HIP device visible, and %d device(s) were asked for", n_devices);
        if (!out || !feats_host) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null argument");
        if (dim != MI355REC_DIM) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "dim must be %d, got %d", MI355REC_DIM, dim);
        if (n < 1 || n > 0xfffffffell) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "rows %lld out of range", (long long)n);
        mi355rec_sharded* h = nullptr;
        try {   // (nothing may leave a C entry point as an exception: the handle and its note allocate)
            h = new mi355rec_sharded();
            h->n = n;
            h->peer_rows = false;
            h->cpu = mi355cpu::node_create(feats_host, n);
            if (!h->cpu) {
                delete h;
                return sfail(nullptr, MI355REC_ERR_OUT_OF_MEMORY, "CPU backend: cannot hold %lld rows", (long long)n);
            }
            h->note = "CPU backend: no HIP device visible (" + std::to_string(mi355cpu::threads(mi355cpu::node_catalogue(h->cpu))) +
                      " OpenMP thread(s))";
        } catch (const std::bad_alloc&) {
            if (h && h->cpu) mi355cpu::node_destroy(h->cpu);
            if (h) h->cpu = nullptr;
            delete h;
            return sfail(nullptr, MI355REC_ERR_OUT_OF_MEMORY, "out of host memory for the CPU backend's handle");
        }
        *out = h;
        return MI355REC_OK;
    }
    std::vector<int> devs;
    if (devices) {   // an explicit list (a device may repeat: virtual shards / replicas on a one-GPU box)
        if (n_devices < 1 || n_devices > MI355REC_MAX_SHARDS)
            return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "n_devices must be in [1, %d] with an explicit list, got %d", MI355REC_MAX_SHARDS, n_devices);
        devs.assign(devices, devices + n_devices);
    } else {
        if (n_devices < 0 || n_devices > visible)
            return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "n_devices %d but %d device(s) visible", n_devices, visible);
        int g = n_devices;
        if (g == 0)   // the library decides: replicas on every device; shards by the size of the catalogue
            g = placement == MI355REC_PLACEMENT_REPLICATED ? (visible < MI355REC_MAX_SHARDS ? visible : MI355REC_MAX_SHARDS)
                                                            : mi355rec_auto_shards(n, visible);
        devs.resize(g);
        for (int d = 0; d < g; ++d) devs[d] = d;
    }
    return create_on(feats_host, n, dim, devs.data(), static_cast<int>(devs.size()), placement == MI355REC_PLACEMENT_REPLICATED, out);
}

int mi355rec_create_sharded(const float* feats_host, int64_t n, int dim, int n_devices, mi355rec_sharded_t** out) {
    return mi355rec_create_placed(feats_host, n, dim, nullptr, n_devices, MI355REC_PLACEMENT_SHARDED, out);
}

int mi355rec_sharded_placement(const mi355rec_sharded_t* h) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return MI355REC_PLACEMENT_CPU;
    return h->replicated ? MI355REC_PLACEMENT_REPLICATED : MI355REC_PLACEMENT_SHARDED;
}

int mi355rec_sharded_set_transport(mi355rec_sharded_t* h, int transport) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->replicated || h->cpu) {   // nothing is ever exchanged (replicas; the CPU backend): either value is accepted and ignored
        if (transport != MI355REC_TRANSPORT_PEER && transport != MI355REC_TRANSPORT_RCCL)
            return sfail(h, MI355REC_ERR_INVALID_ARG, "unknown transport %d", transport);
        return MI355REC_OK;
    }
    if (transport == MI355REC_TRANSPORT_PEER) {
        if (!h->peer_ok) return sfail(h, MI355REC_ERR_INVALID_ARG, "peer access to device %d is not available from every shard", h->shards[0].device);
    } else if (transport != MI355REC_TRANSPORT_RCCL) {
        return sfail(h, MI355REC_ERR_INVALID_ARG, "unknown transport %d", transport);
    }
    if (transport != h->transport) {   // an open stream window is closed under the transport it was filled with
        DeviceRestore restore;
        int rc = stream_flush(h);
        if (rc) return rc;
        rc = drain_workers(h);
        if (rc) return rc;
    }
    h->transport = transport;
    return MI355REC_OK;
}

int mi355rec_sharded_info(const mi355rec_sharded_t* h, int* n_shards, int* transport, int64_t* rows,
                          int* devices_out, int64_t* shard_rows_out) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (n_shards) *n_shards = static_cast<int>(h->shards.size());
    if (transport) *transport = h->transport;
    if (rows) *rows = h->n;
    for (size_t r = 0; r < h->shards.size(); ++r) {
        if (devices_out) devices_out[r] = h->shards[r].device;
        if (shard_rows_out) shard_rows_out[r] = h->shards[r].hi - h->shards[r].lo;
    }
    return MI355REC_OK;
}

const char* mi355rec_sharded_note(const mi355rec_sharded_t* h) { return h ? h->note.c_str() : ""; }

int mi355rec_sharded_set_timing(mi355rec_sharded_t* h, int enabled) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return MI355REC_OK;   // (no kernels to time)
    const int drc = drain_workers(h);
    if (drc) return drc;
    for (Shard& s : h->shards) S_ENG(h, s, mi355rec_set_timing(s.engine, enabled));
    return MI355REC_OK;
}

int mi355rec_sharded_shard_stats(const mi355rec_sharded_t* hc, int shard, mi355rec_stats_t* out) {
    mi355rec_sharded_t* h = const_cast<mi355rec_sharded_t*>(hc);
    if (!h || !out) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (shard < 0 || shard >= static_cast<int>(h->shards.size()))
        return sfail(h, MI355REC_ERR_INVALID_ARG, h->cpu ? "the CPU backend has no device shard %d" : "no shard %d", shard);
    const int drc = drain_workers(h);
    if (drc) return drc;
    const Shard& s = h->shards[shard];
    S_ENG(h, s, mi355rec_stats(s.engine, out));
    return MI355REC_OK;
}

int mi355rec_sharded_set_replica(mi355rec_sharded_t* h, int mode) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return MI355REC_OK;   // (the CPU backend scans the fp32 rows, always)
    DeviceRestore restore;
    int rc = stream_flush(h);
    if (rc) return rc;
    rc = drain_workers(h);
    if (rc) return rc;
    for (Shard& s : h->shards) {
        if (s.hi == s.lo) continue;
        S_ENG(h, s, mi355rec_set_replica(s.engine, mode));
    }
    return MI355REC_OK;
}

int mi355rec_sharded_rows_by_pointer(const mi355rec_sharded_t* h) { return h && h->peer_rows ? 1 : 0; }

int mi355rec_sharded_query_batch_topn(mi355rec_sharded_t* h, const float* queries, int batch,
                                      const int64_t* exclude_global, int topn, int64_t* out_idx, float* out_score,
                                      int* out_count) {
    if (!h || !queries || !out_idx) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return sfail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    if (topn <= 0) return sfail(h, MI355REC_ERR_INVALID_ARG, "topn must be positive, got %d", topn);
    if (h->cpu) {
        for (int b = 0; b < batch; ++b) {
            const char* why = nullptr;
            const int rc = mi355cpu::node_query(h->cpu, queries + static_cast<size_t>(b) * MI355REC_DIM, exclude_global ? exclude_global[b] : -1,
                                                topn, out_idx + static_cast<size_t>(b) * topn,
                                                out_score ? out_score + static_cast<size_t>(b) * topn : nullptr, out_count ? out_count + b : nullptr, &why);
            if (rc) return cpu_result(h, rc, why);
        }
        return MI355REC_OK;
    }
    if (h->shards.size() == 1) {   // one shard: its own handle is the whole engine (row_base 0)
        const int rc = mi355rec_query_batch_topn(h->shards[0].engine, queries, batch, exclude_global, topn, out_idx, out_score, out_count);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "%s", mi355rec_last_error(h->shards[0].engine));
    }
    if (h->replicated) {   // any replica is the whole engine
        DeviceRestore restore;
        Shard* s = nullptr;
        int rc = take_replica(h, &s);
        if (rc) return rc;
        rc = mi355rec_query_batch_topn(s->engine, queries, batch, exclude_global, topn, out_idx, out_score, out_count);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "replica on device %d: %s", s->device, mi355rec_last_error(s->engine));
    }
    // lists are at most n long
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    DeviceRestore restore;
    if (eff > MI355REC_MAX_TOPN_FAST) return run_queries_large(h, queries, exclude_global, batch, eff, topn, out_idx, out_score, out_count);
    const int rc = run_queries(h, queries, nullptr, exclude_global, batch, eff);
    if (rc) return rc;
    copy_rows(h->h_idx, h->h_score, batch, eff, topn, out_idx, out_score, out_count);
    return MI355REC_OK;
}

int mi355rec_sharded_query_topn(mi355rec_sharded_t* h, const float* query12, int64_t exclude_global, int topn,
                                int64_t* out_idx, float* out_score, int* out_count) {
    return mi355rec_sharded_query_batch_topn(h, query12, 1, &exclude_global, topn, out_idx, out_score, out_count);
}

int mi355rec_sharded_query_row_topn(mi355rec_sharded_t* h, int64_t global_row, int topn, int64_t* out_idx,
                                    float* out_score, int* out_count) {
    if (!h || !out_idx) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (global_row < 0 || global_row >= h->n)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)global_row);
    if (topn <= 0) return sfail(h, MI355REC_ERR_INVALID_ARG, "topn must be positive, got %d", topn);
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_query(h->cpu, mi355cpu::row(mi355cpu::node_catalogue(h->cpu), global_row), global_row, topn,
                                                  out_idx, out_score, out_count, &why), why);
    }
    if (h->shards.size() == 1) {   // what Recommender::recommendByIndex costs on a one-GPU box: exactly mi355rec_query_row_topn
        const int rc = mi355rec_query_row_topn(h->shards[0].engine, global_row, topn, out_idx, out_score, out_count);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "%s", mi355rec_last_error(h->shards[0].engine));
    }
    if (h->replicated) {
        DeviceRestore restore;
        Shard* s = nullptr;
        int rc = take_replica(h, &s);
        if (rc) return rc;
        rc = mi355rec_query_row_topn(s->engine, global_row, topn, out_idx, out_score, out_count);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "replica on device %d: %s", s->device, mi355rec_last_error(s->engine));
    }
    DeviceRestore restore;
    float q[MI355REC_DIM];
    const float* qptr = nullptr;
    int rc = locate_row(h, global_row, &qptr, q);
    if (rc) return rc;
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    if (eff > MI355REC_MAX_TOPN_FAST) {
        if (qptr) {   // cold path: by value
            const Shard* own = owner_of(h, global_row);
            rc = mi355rec_fetch_row(own->engine, global_row - own->lo, q);
            if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", own->device, mi355rec_last_error(own->engine));
        }
        return run_queries_large(h, q, &global_row, 1, eff, topn, out_idx, out_score, out_count);
    }
    rc = run_queries(h, q, qptr, &global_row, 1, eff);
    if (rc) return rc;
    copy_rows(h->h_idx, h->h_score, 1, eff, topn, out_idx, out_score, out_count);
    return MI355REC_OK;
}

int mi355rec_sharded_scores_row(mi355rec_sharded_t* h, int64_t global_row, float* out_host) {
    if (!h || !out_host) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (global_row < 0 || global_row >= h->n)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)global_row);
    if (h->cpu) {
        const mi355cpu::Catalogue* c = mi355cpu::node_catalogue(h->cpu);
        mi355cpu::scores(c, mi355cpu::row(c, global_row), out_host);
        return MI355REC_OK;
    }
    DeviceRestore restore;
    if (h->replicated) {
        Shard* s = nullptr;
        int rc = take_replica(h, &s);
        if (rc) return rc;
        rc = mi355rec_scores_row(s->engine, global_row, out_host);
        return rc == MI355REC_OK ? rc : sfail(h, rc, "replica on device %d: %s", s->device, mi355rec_last_error(s->engine));
    }
    float q[MI355REC_DIM];
    const Shard* own = owner_of(h, global_row);
    int rc = drain_workers(h);   // cold path: the caller's thread drives every shard itself
    if (rc) return rc;
    rc = mi355rec_fetch_row(own->engine, global_row - own->lo, q);
    if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", own->device, mi355rec_last_error(own->engine));
    for (const Shard& s : h->shards) {
        if (s.hi == s.lo) continue;
        rc = mi355rec_scores(s.engine, q, out_host + s.lo);
        if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", s.device, mi355rec_last_error(s.engine));
    }
    return MI355REC_OK;
}

// ---- the stream ---------------------------------------------------------------------------

int mi355rec_sharded_set_window(mi355rec_sharded_t* h, int window) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (window < 1 || window > kMaxWindow) return sfail(h, MI355REC_ERR_INVALID_ARG, "window must be in [1, %d], got %d", kMaxWindow, window);
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_set_window(h->cpu, window, &why), why);
    }
    if (window == h->s_window) return MI355REC_OK;
    DeviceRestore restore;
    int rc = stream_flush(h);   // the open window is closed in the old geometry
    if (rc) return rc;
    rc = drain_workers(h);
    if (rc) return rc;
    // buffers are re-made by the next enqueue (ensure_stream sees s_alloc_window != s_window)
    h->next_ticket = (h->next_ticket + window - 1) / window * window;
    h->issued_upto = h->next_ticket;
    for (Shard& s : h->shards) {
        S_HIP(h, hipSetDevice(s.device));
        S_HIP(h, hipStreamSynchronize(s.stream));
    }
    free_stream(h);
    for (Window& w : h->win) {
        w.abs = -1;
        w.count = 0;
        w.handed = false;
        w.issued = false;
    }
    h->s_window = window;
    return MI355REC_OK;
}

int mi355rec_sharded_set_window_mode(mi355rec_sharded_t* h, int batched) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return MI355REC_OK;   // (a query is computed when it is enqueued, either way)
    if ((batched != 0) == h->batched_windows) return MI355REC_OK;
    DeviceRestore restore;
    int rc = stream_flush(h);
    if (rc) return rc;
    rc = drain_workers(h);
    if (rc) return rc;
    for (Shard& s : h->shards) {
        S_HIP(h, hipSetDevice(s.device));
        S_HIP(h, hipStreamSynchronize(s.stream));
    }
    free_stream(h);   // the next enqueue re-makes the buffers and decides the mode again
    for (Window& w : h->win) {
        w.abs = -1;
        w.count = 0;
        w.handed = false;
        w.issued = false;
    }
    h->batched_windows = batched != 0;
    return MI355REC_OK;
}

int mi355rec_sharded_enqueue_query(mi355rec_sharded_t* h, const float* query12, int64_t exclude_global, int topn,
                                   int64_t* ticket) {
    if (!h || !query12) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_enqueue(h->cpu, query12, exclude_global, topn, ticket, &why), why);
    }
    DeviceRestore restore;
    return stream_enqueue(h, nullptr, query12, exclude_global, topn, ticket);
}

int mi355rec_sharded_enqueue_row(mi355rec_sharded_t* h, int64_t global_row, int topn, int64_t* ticket) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (global_row < 0 || global_row >= h->n)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)global_row);
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_enqueue(h->cpu, mi355cpu::row(mi355cpu::node_catalogue(h->cpu), global_row), global_row, topn,
                                                    ticket, &why), why);
    }
    DeviceRestore restore;
    float q[MI355REC_DIM] = {0};
    const float* qptr = nullptr;
    if (!h->replicated) {
        const int rc = locate_row(h, global_row, &qptr, q);
        if (rc) return rc;
    }
    return stream_enqueue(h, qptr, q, global_row, topn, ticket, global_row);
}

int mi355rec_sharded_enqueue_flush(mi355rec_sharded_t* h) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) return mi355cpu::node_flush(h->cpu);
    DeviceRestore restore;
    const int64_t t0 = now_ns();
    const int rc = stream_flush(h);
    h->st_host_ns += now_ns() - t0;
    return rc;
}

int mi355rec_sharded_wait(mi355rec_sharded_t* h, int64_t ticket, int64_t* out_idx, float* out_score, int* out_count) {
    if (!h || !out_idx) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (h->cpu) {
        const char* why = nullptr;
        return cpu_result(h, mi355cpu::node_wait(h->cpu, ticket, out_idx, out_score, out_count, &why), why);
    }
    if (!h->s_topn || ticket < 0 || ticket >= h->next_ticket)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "ticket %lld was never handed out", (long long)ticket);
    DeviceRestore restore;
    const int W = h->s_window;
    const int64_t abs = ticket / W;
    const int w = static_cast<int>(abs % kStreamDepth);
    const int slot = static_cast<int>(ticket % W);
    Window& win = h->win[w];
    if (win.abs != abs || slot >= win.count)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "the results of ticket %lld are no longer kept (ring of %d windows of %d)",
                     (long long)ticket, kStreamDepth, W);
    if (!win.issued) {   // its window is still open (or waits for the lag): close it now
        const int rc = stream_flush(h);
        if (rc) return rc;
    }
    {
        const int rc = wait_window(h, w);   // the merge has been enqueued, `merged` recorded, and the results are in host memory
        if (rc) return rc;
    }
    const int topn = h->s_topn;
    const size_t off = (static_cast<size_t>(w) * W + slot) * topn;
    copy_rows(h->s_hidx + off, h->s_hscore + off, 1, topn, topn, out_idx, out_score, out_count);
    return MI355REC_OK;
}

int mi355rec_sharded_stream_stats(const mi355rec_sharded_t* h, int64_t* queries, int64_t* exchanges, int64_t* host_ns) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (h->cpu) {
        mi355cpu::node_stream_stats(h->cpu, queries, exchanges, host_ns);   // ("exchanges": windows opened)
        return MI355REC_OK;
    }
    if (queries) *queries = h->st_queries;
    if (exchanges) *exchanges = h->st_exchanges;
    if (host_ns) *host_ns = h->st_host_ns;
    return MI355REC_OK;
}

}  // extern "C"
