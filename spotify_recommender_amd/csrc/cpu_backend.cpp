// cpu_backend.cpp — see cpu_backend.h.  Own code; arithmetic contract: Recommender.cu:256-273, selection: :293-315.
#include "cpu_backend.h"

#include <omp.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace mi355cpu {

namespace {

constexpr int kDim = 12;   // Song.h:12 FEATURE_COUNT

// Order-preserving image of a score (larger image = larger score; -0.0f and +0.0f share one), and the packed
// candidate key of include/mi355rec.h: image << 32 | ~row, so that a larger key is the better candidate in the
// canonical order (score descending, then row ascending).  Must agree with mi355rec_pack_key (tests compare).
inline uint32_t ordered(float s) {
    s = s + 0.0f;
    uint32_t u;
    std::memcpy(&u, &s, sizeof u);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

inline float unordered(uint32_t o) {
    const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    float s;
    std::memcpy(&s, &u, sizeof s);
    return s;
}

// CPUs the process may use by its cgroup quota (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), rounded up;
// 0 = no quota (or not readable).  sched affinity is already in omp_get_max_threads().
int cpu_quota() {
    long long quota = -1, period = 0;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = {0};
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
        std::fclose(f);
    } else {
        if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(g, "%lld", &quota) != 1) quota = -1;
            std::fclose(g);
        }
        if (FILE* g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(g, "%lld", &period) != 1) period = 0;
            std::fclose(g);
        }
    }
    if (quota <= 0 || period <= 0) return 0;
    return static_cast<int>((quota + period - 1) / period);
}

inline uint64_t pack(float s, uint32_t row) { return (static_cast<uint64_t>(ordered(s)) << 32) | static_cast<uint32_t>(~row); }

inline float query_norm(const float* q) {   // Recommender.cu:259-261
    float qn = 0.0f;
    for (int j = 0; j < kDim; ++j) qn += q[j] * q[j];
    return std::sqrt(qn);
}

inline float score(const float* q, float qn, const float* f) {   // Recommender.cu:262-272, one row
    float dot = 0.0f;
    float norm = 0.0f;
    for (int j = 0; j < kDim; ++j) {
        dot += q[j] * f[j];
        norm += f[j] * f[j];
    }
    norm = std::sqrt(norm) * qn;
    if (norm > 1e-8f) return std::max(-1.0f, std::min(1.0f, dot / norm));
    return 0.0f;
}

// Keeps the best `cap` keys it has been shown in a caller-owned buffer of 2 * cap keys, cut back by selection when
// full.  It never allocates: it is used inside an OpenMP region, which no exception may leave.
struct Best {
    uint64_t* keys = nullptr;
    size_t size = 0;
    size_t cap = 0;
    uint64_t floor = 0;   // a key must beat this to be worth keeping (rises at every cut)
    inline void offer(uint64_t k) {
        if (k <= floor) return;
        keys[size++] = k;
        if (size >= 2 * cap) cut();
    }
    void cut() {
        if (size <= cap) return;
        std::nth_element(keys, keys + (cap - 1), keys + size, std::greater<uint64_t>());
        size = cap;
        floor = keys[cap - 1] - 1;   // keys are unique (they carry the row): the cap-th best stays, nothing below it can matter
    }
};

}  // namespace

struct Catalogue {
    std::vector<float> feats;
    int64_t n = 0;
    int threads = 1;
};

Catalogue* create(const float* feats_rowmajor, int64_t n, int threads) {
    if (n < 0 || (n > 0 && !feats_rowmajor)) return nullptr;
    Catalogue* c = nullptr;
    try {   // (called from behind a C-ABI: no exception may leave)
        c = new Catalogue();
        c->n = n;
        c->feats.assign(feats_rowmajor, feats_rowmajor + static_cast<size_t>(n) * kDim);
    } catch (const std::bad_alloc&) {
        delete c;
        return nullptr;
    }
    int t = threads > 0 ? threads : omp_get_max_threads();
    if (threads <= 0) {   // no more threads than the process may really run (a cgroup quota throttles the rest)
        const int quota = cpu_quota();
        if (quota > 0 && t > quota) t = quota;
    }
    if (t < 1) t = 1;
    // a thread per ~8 k rows at most (~0.1 ms of rows: below that the fork costs more than the rows).  Measured on an
    // 8-vCPU host at 114 000 rows x top-10, the team being exactly the threads with rows (num_threads below): 1 040
    // queries/s on 2 threads, 2 970 on 4, 4 340 on 8.  (Round 4 allowed a thread per 64 k rows: with the process's whole
    // default team forked per call the idle threads' spinning made more threads look worse than they are.)
    const int64_t useful = n / 8192 + 1;
    if (t > useful) t = static_cast<int>(useful);
    c->threads = t;
    return c;
}

void destroy(Catalogue* c) { delete c; }
int64_t rows(const Catalogue* c) { return c->n; }
int threads(const Catalogue* c) { return c->threads; }
const float* row(const Catalogue* c, int64_t r) { return c->feats.data() + static_cast<size_t>(r) * kDim; }

// The team of a call is exactly the threads that have rows to score (c->threads: bounded by the cgroup quota and by one
// thread per ~64 k rows).  Without the clause every call forks the process's whole default team; the threads without
// work then spin at the join barrier — in a CPU-quota container they burn the quota the workers need.
void scores(const Catalogue* c, const float* q12, float* out_n) {
    const float qn = query_norm(q12);
    const float* f = c->feats.data();
    const int64_t n = c->n;
#pragma omp parallel for schedule(static) num_threads(c->threads)
    for (int64_t i = 0; i < n; ++i) out_n[i] = score(q12, qn, f + i * kDim);
}

int topn(const Catalogue* c, const float* q12, int64_t exclude, int topn, int64_t* out_idx, float* out_score) {
    if (topn <= 0 || c->n <= 0) return 0;
    const float qn = query_norm(q12);
    const float* f = c->feats.data();
    const int64_t n = c->n;
    const size_t cap = static_cast<size_t>(static_cast<int64_t>(topn) < n ? topn : n);
    // The rows go to `parts` threads in contiguous blocks.  Everything the region needs is allocated HERE, before it (a
    // std::bad_alloc from these lines reaches the caller's try / catch; one thrown inside the region would be
    // std::terminate behind the C-ABI): one flat pool, 2 * cap keys per part, and the parts' bookkeeping.
    const int parts = c->threads;
    std::vector<uint64_t> pool(static_cast<size_t>(parts) * 2 * cap);
    std::vector<Best> per_part(static_cast<size_t>(parts));
    for (int p = 0; p < parts; ++p) {
        per_part[static_cast<size_t>(p)].keys = pool.data() + static_cast<size_t>(p) * 2 * cap;
        per_part[static_cast<size_t>(p)].cap = cap;
    }
#pragma omp parallel num_threads(parts)
    {
        const int team = omp_get_num_threads();   // (may be smaller than asked for: nested regions, OMP_THREAD_LIMIT)
        const int use = parts < team ? parts : team;
        const int t = omp_get_thread_num();
        if (t < use) {
            // (fewer threads than parts: the last thread takes the remaining parts as well)
            const int p0 = t, p1 = (t == use - 1) ? parts : t + 1;
            for (int p = p0; p < p1; ++p) {
                Best& mine = per_part[static_cast<size_t>(p)];
                const int64_t lo = n * p / parts, hi = n * (p + 1) / parts;
                for (int64_t i = lo; i < hi; ++i) {
                    if (i == exclude) continue;   // by index, not by score (Recommender.cu:296)
                    mine.offer(pack(score(q12, qn, f + i * kDim), static_cast<uint32_t>(i)));
                }
                mine.cut();   // at most cap keys per part from here on
            }
        }
    }
    // the parts' survivors, compacted to the front of the pool (part p's keys start at or behind where they go) and cut
    size_t total = 0;
    for (const Best& b : per_part) {
        std::memmove(pool.data() + total, b.keys, b.size * sizeof(uint64_t));
        total += b.size;
    }
    const size_t count = total < cap ? total : cap;
    std::partial_sort(pool.begin(), pool.begin() + count, pool.begin() + total, std::greater<uint64_t>());
    for (size_t i = 0; i < count; ++i) {
        out_idx[i] = static_cast<int64_t>(static_cast<uint32_t>(~static_cast<uint32_t>(pool[i])));
        if (out_score) out_score[i] = unordered(static_cast<uint32_t>(pool[i] >> 32));
    }
    return static_cast<int>(count);
}

}  // namespace mi355cpu

// ---- the handle: synchronous calls + the ticketed stream ----------------------------------------------------------
#include <chrono>

#include "mi355rec_diag.h"

namespace mi355cpu {

namespace {
constexpr int kDepth = 4;        // windows whose results are kept (the GPU path's ring, csrc/sharded.hip kStreamDepth)
constexpr int kMaxWindow = 64;

int64_t now_ns() {
    return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

struct Node {
    Catalogue* cat = nullptr;
    int window = 16;
    int s_topn = 0;                       // topn of the stream's current geometry (0: no query yet)
    int64_t next_ticket = 0;
    struct Win {
        int64_t abs = -1;
        int count = 0;
    } win[kDepth];
    std::vector<int64_t> idx;             // [kDepth][window][s_topn]
    std::vector<float> score;
    std::vector<int> counts;              // [kDepth][window]
    int64_t st_queries = 0, st_windows = 0, st_ns = 0;
};

Node* node_create(const float* feats_rowmajor, int64_t n) {
    Catalogue* c = create(feats_rowmajor, n, 0);
    if (!c) return nullptr;
    Node* h = new (std::nothrow) Node();
    if (!h) {
        destroy(c);
        return nullptr;
    }
    h->cat = c;
    return h;
}

void node_destroy(Node* h) {
    if (!h) return;
    destroy(h->cat);
    delete h;
}

const Catalogue* node_catalogue(const Node* h) { return h->cat; }

int node_query(Node* h, const float* q12, int64_t exclude, int topn_asked, int64_t* out_idx, float* out_score, int* out_count,
               const char** why) {
    if (topn_asked <= 0) {
        *why = "topn must be positive";
        return MI355REC_ERR_INVALID_ARG;
    }
    int c = 0;
    try {
        c = topn(h->cat, q12, exclude, topn_asked, out_idx, out_score);
    } catch (const std::bad_alloc&) {
        *why = "out of host memory";
        return MI355REC_ERR_OUT_OF_MEMORY;
    }
    for (int i = c; i < topn_asked; ++i) {   // the C-ABI pads with -1 / 0
        out_idx[i] = -1;
        if (out_score) out_score[i] = 0.0f;
    }
    if (out_count) *out_count = c;
    return MI355REC_OK;
}

int node_set_window(Node* h, int window, const char** why) {
    if (window < 1 || window > kMaxWindow) {
        *why = "window out of range";
        return MI355REC_ERR_INVALID_ARG;
    }
    if (window == h->window) return MI355REC_OK;
    node_flush(h);
    h->next_ticket = (h->next_ticket + window - 1) / window * window;
    h->window = window;
    h->s_topn = 0;   // buffers are re-made by the next enqueue
    for (auto& w : h->win) w = Node::Win();
    return MI355REC_OK;
}

int node_enqueue(Node* h, const float* q12, int64_t exclude, int topn_asked, int64_t* ticket, const char** why) {
    if (topn_asked <= 0 || topn_asked > MI355REC_MAX_TOPN_FAST) {
        *why = "topn must be in [1, 1024] for streamed queries";
        return MI355REC_ERR_INVALID_ARG;
    }
    const int64_t t0 = now_ns();
    const int W = h->window;
    if (h->s_topn != topn_asked) {   // a change of geometry closes the stream; tickets keep growing, window-aligned
        node_flush(h);
        h->s_topn = 0;
        try {
            h->idx.assign(static_cast<size_t>(kDepth) * W * topn_asked, -1);
            h->score.assign(static_cast<size_t>(kDepth) * W * topn_asked, 0.0f);
            h->counts.assign(static_cast<size_t>(kDepth) * W, 0);
        } catch (const std::bad_alloc&) {
            *why = "out of host memory";
            return MI355REC_ERR_OUT_OF_MEMORY;
        }
        h->s_topn = topn_asked;
        for (auto& w : h->win) w = Node::Win();
    }
    const int64_t t = h->next_ticket;
    const int64_t abs = t / W;
    const int w = static_cast<int>(abs % kDepth);
    const int slot = static_cast<int>(t % W);
    if (slot == 0) {
        h->win[w].abs = abs;
        h->win[w].count = 0;
        ++h->st_windows;
    }
    const size_t at = (static_cast<size_t>(w) * W + slot) * topn_asked;
    int c = 0;
    try {
        c = topn(h->cat, q12, exclude, topn_asked, &h->idx[at], &h->score[at]);
    } catch (const std::bad_alloc&) {
        *why = "out of host memory";
        return MI355REC_ERR_OUT_OF_MEMORY;
    }
    for (int i = c; i < topn_asked; ++i) {
        h->idx[at + i] = -1;
        h->score[at + i] = 0.0f;
    }
    h->counts[static_cast<size_t>(w) * W + slot] = c;
    ++h->win[w].count;
    ++h->next_ticket;
    ++h->st_queries;
    if (ticket) *ticket = t;
    h->st_ns += now_ns() - t0;
    return MI355REC_OK;
}

int node_flush(Node* h) {   // nothing is ever pending; the next query opens a new window, as on the GPU path
    const int W = h->window;
    h->next_ticket = (h->next_ticket + W - 1) / W * W;
    return MI355REC_OK;
}

int node_wait(Node* h, int64_t ticket, int64_t* out_idx, float* out_score, int* out_count, const char** why) {
    if (!h->s_topn || ticket < 0 || ticket >= h->next_ticket) {
        *why = "ticket was never handed out";
        return MI355REC_ERR_INVALID_ARG;
    }
    const int W = h->window;
    const int64_t abs = ticket / W;
    const int w = static_cast<int>(abs % kDepth);
    const int slot = static_cast<int>(ticket % W);
    if (h->win[w].abs != abs || slot >= h->win[w].count) {
        *why = "the results of this ticket are no longer kept (ring of 4 windows)";
        return MI355REC_ERR_INVALID_ARG;
    }
    const size_t at = (static_cast<size_t>(w) * W + slot) * h->s_topn;
    std::memcpy(out_idx, &h->idx[at], sizeof(int64_t) * h->s_topn);
    if (out_score) std::memcpy(out_score, &h->score[at], sizeof(float) * h->s_topn);
    if (out_count) *out_count = h->counts[static_cast<size_t>(w) * W + slot];
    return MI355REC_OK;
}

void node_stream_stats(const Node* h, int64_t* queries, int64_t* windows, int64_t* host_ns) {
    if (queries) *queries = h->st_queries;
    if (windows) *windows = h->st_windows;
    if (host_ns) *host_ns = h->st_ns;
}

}  // namespace mi355cpu
