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
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "mi355rec.h"

namespace {

thread_local std::string g_sharded_error;

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
};

}  // namespace

struct mi355rec_sharded {
    int64_t n = 0;
    std::vector<Shard> shards;
    int transport = MI355REC_TRANSPORT_PEER;
    bool peer_ok = true;
    bool rccl_ready = false;
    size_t cap = 0;                     // keys per shard the buffers hold (batch x topn)
    mi355rec_key_t* gather0 = nullptr;  // [shards][cap] on devices[0] (PEER transport)
    mi355rec_key_t* d_keys = nullptr;   // merged results on devices[0]
    int64_t* d_idx = nullptr;
    float* d_score = nullptr;
    int64_t* h_idx = nullptr;           // pinned
    float* h_score = nullptr;
    hipEvent_t merged = nullptr;        // device 0 has consumed the gather buffer
    std::string err;
};

namespace {

int sfail(mi355rec_sharded* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
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

// balanced contiguous blocks: the first n % g shards hold one row more
void bounds(int64_t n, int g, int r, int64_t& lo, int64_t& hi) {
    const int64_t per = n / g, rem = n % g;
    lo = r * per + (r < rem ? r : rem);
    hi = lo + per + (r < rem ? 1 : 0);
}

int ensure_capacity(mi355rec_sharded* h, size_t keys_per_shard) {
    if (keys_per_shard <= h->cap) return MI355REC_OK;
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
    S_HIP(h, hipHostMalloc(&h->h_idx, sizeof(int64_t) * cap, hipHostMallocDefault));
    S_HIP(h, hipHostMalloc(&h->h_score, sizeof(float) * cap, hipHostMallocDefault));
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

// `count` queries (host vectors) -> merged results in h->d_keys / d_idx / d_score on
// devices[0] and, after the final sync, in the pinned host mirrors.
int run_queries(mi355rec_sharded* h, const float* queries, const int64_t* exclude, int count, int topn) {
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
        S_HIP(h, hipSetDevice(s.device));
        mi355rec_key_t* dst = rccl ? s.local_keys : h->gather0 + static_cast<size_t>(r) * per_shard;
        if (count == 1) {
            S_ENG(h, s, mi355rec_enqueue_query_keys(s.engine, queries, exclude ? exclude[0] : -1, topn, dst, s.stream));
        } else {
            S_ENG(h, s, mi355rec_enqueue_batch_keys(s.engine, queries, exclude, count, topn, dst, s.stream));
        }
    }
    Shard& root = h->shards[0];
    const mi355rec_key_t* lists = h->gather0;
    if (rccl) {
        // ONE grouped all-gather: every device receives [shard][query][key]
        ncclResult_t nrc = g_rccl.GroupStart();
        for (int r = 0; r < g && nrc == 0; ++r) {
            Shard& s = h->shards[r];
            nrc = g_rccl.AllGather(s.local_keys, s.gathered, per_shard, kNcclUint64, s.comm, s.stream);
        }
        const ncclResult_t erc = g_rccl.GroupEnd();
        if (nrc != 0 || erc != 0)
            return sfail(h, MI355REC_ERR_HIP, "ncclAllGather: %s",
                         g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc ? nrc : erc) : "error");
        lists = root.gathered;   // the collective is ordered on root.stream already
    } else {
        for (int r = 1; r < g; ++r) {
            Shard& s = h->shards[r];
            S_HIP(h, hipSetDevice(s.device));
            S_HIP(h, hipEventRecord(s.done, s.stream));
        }
        S_HIP(h, hipSetDevice(root.device));
        for (int r = 1; r < g; ++r) S_HIP(h, hipStreamWaitEvent(root.stream, h->shards[r].done, 0));
    }
    S_HIP(h, hipSetDevice(root.device));
    // [shard][query][key]: list l of query b starts at b * topn + l * per_shard
    S_ENG(h, root, mi355rec_enqueue_merge_keys_batch(root.engine, lists, g, topn, static_cast<int64_t>(per_shard),
                                                     static_cast<int64_t>(topn), count, topn, h->d_keys, h->d_idx,
                                                     h->d_score, root.stream));
    S_HIP(h, hipMemcpyAsync(h->h_idx, h->d_idx, per_shard * sizeof(int64_t), hipMemcpyDeviceToHost, root.stream));
    S_HIP(h, hipMemcpyAsync(h->h_score, h->d_score, per_shard * sizeof(float), hipMemcpyDeviceToHost, root.stream));
    S_HIP(h, hipStreamSynchronize(root.stream));
    // the calls are synchronous, so nothing of the next call can overtake this merge;
    // the other shards' streams are drained too before their buffers are reused
    for (int r = 1; r < g; ++r) {
        S_HIP(h, hipSetDevice(h->shards[r].device));
        S_HIP(h, hipStreamSynchronize(h->shards[r].stream));
    }
    return MI355REC_OK;
}

// topn above the single-launch merge limit (the CLI's `-n 5000`): every shard serves the
// query in rounds of 1024 (as a single engine does), the G sorted key lists come back
// to the host and are merged there — pure key ordering, no arithmetic on scores.  Cold
// path, one query at a time.
int run_queries_large(mi355rec_sharded* h, const float* queries, const int64_t* exclude, int count, int eff, int topn,
                      int64_t* out_idx, float* out_score, int* out_count) {
    const int g = static_cast<int>(h->shards.size());
    int rc = ensure_capacity(h, static_cast<size_t>(eff));
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

void copy_out(const mi355rec_sharded* h, int count, int eff, int topn, int64_t* out_idx, float* out_score, int* out_count) {
    for (int b = 0; b < count; ++b) {
        const int64_t* src_i = h->h_idx + static_cast<size_t>(b) * eff;
        const float* src_s = h->h_score + static_cast<size_t>(b) * eff;
        int64_t* dst_i = out_idx + static_cast<size_t>(b) * topn;
        std::memcpy(dst_i, src_i, sizeof(int64_t) * eff);
        for (int i = eff; i < topn; ++i) dst_i[i] = -1;
        if (out_score) {
            float* dst_s = out_score + static_cast<size_t>(b) * topn;
            std::memcpy(dst_s, src_s, sizeof(float) * eff);
            for (int i = eff; i < topn; ++i) dst_s[i] = 0.0f;
        }
        if (out_count) {
            int c = 0;
            while (c < eff && src_i[c] >= 0) ++c;
            out_count[b] = c;
        }
    }
}

struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

extern "C" {

const char* mi355rec_sharded_last_error(const mi355rec_sharded_t* h) {
    return h ? h->err.c_str() : g_sharded_error.c_str();
}

void mi355rec_sharded_destroy(mi355rec_sharded_t* h) {
    if (!h) return;
    DeviceRestore restore;
    for (Shard& s : h->shards) {
        if (hipSetDevice(s.device) != hipSuccess) continue;
        if (s.stream) (void)hipStreamSynchronize(s.stream);
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
        if (h->merged) (void)hipEventDestroy(h->merged);
    }
    delete h;
}

int mi355rec_create_sharded_on(const float* feats_host, int64_t n, int dim, const int* devices, int n_shards,
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
        bounds(n, n_shards, r, s.lo, s.hi);
        if (hipSetDevice(s.device) != hipSuccess) return bail(sfail(nullptr, MI355REC_ERR_HIP, "hipSetDevice(%d) failed", s.device));
        const int rc = mi355rec_create(s.hi > s.lo ? feats_host + s.lo * MI355REC_DIM : nullptr, s.hi - s.lo, dim, s.device,
                                       s.lo, &s.engine);
        if (rc != MI355REC_OK) return bail(sfail(nullptr, rc, "shard %d on device %d: %s", r, s.device, mi355rec_last_global_error()));
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)
            return bail(sfail(nullptr, MI355REC_ERR_HIP, "stream / event creation on device %d failed", s.device));
        // peer stores into the root's gather buffer need the root's memory mapped here
        if (s.device != root) {
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, s.device, root) != hipSuccess || !can) {
                h->peer_ok = false;
            } else {
                const hipError_t e = hipDeviceEnablePeerAccess(root, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) h->peer_ok = false;
                (void)hipGetLastError();
            }
        }
    }
    if (hipSetDevice(root) != hipSuccess || hipEventCreateWithFlags(&h->merged, hipEventDisableTiming) != hipSuccess)
        return bail(sfail(nullptr, MI355REC_ERR_HIP, "event creation on device %d failed", root));
    h->transport = h->peer_ok ? MI355REC_TRANSPORT_PEER : MI355REC_TRANSPORT_RCCL;
    *out = h;
    return MI355REC_OK;
}

int mi355rec_create_sharded(const float* feats_host, int64_t n, int dim, int n_devices, mi355rec_sharded_t** out) {
    const int visible = mi355rec_device_count();
    if (visible <= 0) {
        if (out) *out = nullptr;
        return sfail(nullptr, MI355REC_ERR_NO_DEVICE, "no HIP device visible: the MI355X engine has no CPU fallback");
    }
    if (n_devices == 0) n_devices = visible < MI355REC_MAX_SHARDS ? visible : MI355REC_MAX_SHARDS;
    if (n_devices < 0 || n_devices > visible) {
        if (out) *out = nullptr;
        return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "n_devices %d but %d device(s) visible", n_devices, visible);
    }
    std::vector<int> devs(n_devices);
    for (int d = 0; d < n_devices; ++d) devs[d] = d;
    return mi355rec_create_sharded_on(feats_host, n, dim, devs.data(), n_devices, out);
}

int mi355rec_sharded_set_transport(mi355rec_sharded_t* h, int transport) {
    if (!h) return sfail(nullptr, MI355REC_ERR_INVALID_ARG, "null handle");
    if (transport == MI355REC_TRANSPORT_PEER) {
        if (!h->peer_ok) return sfail(h, MI355REC_ERR_INVALID_ARG, "peer access to device %d is not available from every shard", h->shards[0].device);
    } else if (transport != MI355REC_TRANSPORT_RCCL) {
        return sfail(h, MI355REC_ERR_INVALID_ARG, "unknown transport %d", transport);
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

int mi355rec_sharded_query_batch_topn(mi355rec_sharded_t* h, const float* queries, int batch,
                                      const int64_t* exclude_global, int topn, int64_t* out_idx, float* out_score,
                                      int* out_count) {
    if (!h || !queries || !out_idx) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (batch < 1) return sfail(h, MI355REC_ERR_INVALID_ARG, "batch must be positive");
    if (topn <= 0) return sfail(h, MI355REC_ERR_INVALID_ARG, "topn must be positive, got %d", topn);
    // lists are at most n long
    const int eff = static_cast<int64_t>(topn) < h->n ? topn : static_cast<int>(h->n);
    DeviceRestore restore;
    if (eff > MI355REC_MAX_TOPN_FAST) return run_queries_large(h, queries, exclude_global, batch, eff, topn, out_idx, out_score, out_count);
    const int rc = run_queries(h, queries, exclude_global, batch, eff);
    if (rc) return rc;
    copy_out(h, batch, eff, topn, out_idx, out_score, out_count);
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
    // the query vector lives on the shard that owns the row: 48 bytes come back to the host
    // and go to every shard as kernel arguments (no collective needed)
    DeviceRestore restore;
    float q[MI355REC_DIM];
    for (const Shard& s : h->shards) {
        if (global_row >= s.lo && global_row < s.hi) {
            const int rc = mi355rec_fetch_row(s.engine, global_row - s.lo, q);
            if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", s.device, mi355rec_last_error(s.engine));
            break;
        }
    }
    return mi355rec_sharded_query_batch_topn(h, q, 1, &global_row, topn, out_idx, out_score, out_count);
}

int mi355rec_sharded_scores_row(mi355rec_sharded_t* h, int64_t global_row, float* out_host) {
    if (!h || !out_host) return sfail(h, MI355REC_ERR_INVALID_ARG, "null argument");
    if (global_row < 0 || global_row >= h->n)
        return sfail(h, MI355REC_ERR_INVALID_ARG, "Invalid song index: %lld", (long long)global_row);
    DeviceRestore restore;
    float q[MI355REC_DIM];
    for (const Shard& s : h->shards) {
        if (global_row >= s.lo && global_row < s.hi) {
            const int rc = mi355rec_fetch_row(s.engine, global_row - s.lo, q);
            if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", s.device, mi355rec_last_error(s.engine));
            break;
        }
    }
    for (const Shard& s : h->shards) {
        if (s.hi == s.lo) continue;
        const int rc = mi355rec_scores(s.engine, q, out_host + s.lo);
        if (rc != MI355REC_OK) return sfail(h, rc, "shard on device %d: %s", s.device, mi355rec_last_error(s.engine));
    }
    return MI355REC_OK;
}

}  // extern "C"
