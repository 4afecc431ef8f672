// node_state.hip.h — the node-level handle behind mi355rec_create_placed / _sharded (include/mi355rec.h): its shards, the
// worker thread of each, the exchange of per-shard key lists (peer stores or one ncclAllGather per rank) and the synchronous
// queries.  Host orchestration over the single-device C-ABI: it launches no kernel of its own.  (Part of sharded.hip's
// translation unit; the header of sharded.hip describes the design.)
#pragma once

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
#include "mi355rec_diag.h"

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
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // (optional: diagnostics only)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
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
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
        CommUserRank = reinterpret_cast<decltype(CommUserRank)>(sym("ncclCommUserRank"));
        if (!CommInitAll || !CommDestroy || !AllGather) {
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
    hipStream_t stream = nullptr;       // the engine's own stream (mi355rec_own_stream): not this struct's to destroy
    bool owns_stream = false;
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

}  // namespace
