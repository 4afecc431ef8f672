/*
 * cosine_oracle.c — CPU restatement of the reference's cosine top-N path.
 * TEST INFRASTRUCTURE ONLY (see cosine_oracle.h for the rules and the pinning
 * status).  Build: oracle/Makefile (gcc -O3 -ffp-contract=off -fopenmp, no
 * -march, no -ffast-math, mirroring the reference's Makefile:9 fp behaviour).
 *
 * Citations are file:line in /root/reference.
 */
#include "cosine_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define K ORACLE_FEATURE_COUNT

/* ---- scores: Recommender.cu:256-273 ------------------------------------ */

static inline float query_norm(const float* q) {
    /* Recommender.cu:259-261 */
    float qn = 0.0f;
    for (int j = 0; j < K; ++j) qn += q[j] * q[j];
    return sqrtf(qn);
}

static inline float one_score(const float* q, float qn, const float* f) {
    /* Recommender.cu:264-271 */
    float dot = 0.0f;
    float norm = 0.0f;
    for (int j = 0; j < K; ++j) {
        dot += q[j] * f[j];
        norm += f[j] * f[j];
    }
    norm = sqrtf(norm) * qn;
    if (norm > 1e-8f) {
        float s = dot / norm;
        /* std::min(1.0f, s): (s < 1.0f) ? s : 1.0f ; std::max(-1.0f, m):
         * (-1.0f < m) ? m : -1.0f — written out so NaN behaves as in libstdc++ */
        float m = (s < 1.0f) ? s : 1.0f;
        return (-1.0f < m) ? m : -1.0f;
    }
    return 0.0f;
}

void oracle_scores(const float* feats, int64_t stride, int64_t n,
                   const float* query, float* out) {
    const float qn = query_norm(query);
    for (int64_t i = 0; i < n; ++i) out[i] = one_score(query, qn, feats + i * stride);
}

void oracle_scores_omp(const float* feats, int64_t stride, int64_t n,
                       const float* query, float* out, int threads) {
    const float qn = query_norm(query);
    if (threads <= 0) threads = oracle_max_threads();
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t i = 0; i < n; ++i) out[i] = one_score(query, qn, feats + i * stride);
}

/* ---- heap top-N: Recommender.h:12-22, Recommender.cu:293-315 ------------- */

typedef struct {
    int32_t idx;
    float sim;
} rec_t;

/* Recommender.h:19-21: operator< is inverted, so std::less<Recommendation>
 * (the priority_queue comparator) is comp(a,b) = a.sim > b.sim and the heap
 * top is the LOWEST similarity. */
static inline int rec_less(const rec_t* a, const rec_t* b) { return a->sim > b->sim; }

/* libstdc++ bits/stl_heap.h std::__push_heap */
static void lib_push_heap(rec_t* first, int64_t hole, int64_t top, rec_t value) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && rec_less(&first[parent], &value)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = value;
}

/* libstdc++ bits/stl_heap.h std::__adjust_heap */
static void lib_adjust_heap(rec_t* first, int64_t hole, int64_t len, rec_t value) {
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (rec_less(&first[child], &first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    lib_push_heap(first, hole, top, value);
}

/* priority_queue::push = push_back + std::push_heap */
static void pq_push(rec_t* h, int64_t* size, rec_t v) {
    h[*size] = v;
    (*size)++;
    lib_push_heap(h, *size - 1, 0, h[*size - 1]);
}

/* priority_queue::pop = std::pop_heap + pop_back */
static void pq_pop(rec_t* h, int64_t* size) {
    if (*size > 1) {
        rec_t value = h[*size - 1];
        h[*size - 1] = h[0];
        lib_adjust_heap(h, 0, *size - 1, value);
    }
    (*size)--;
}

int64_t oracle_topn_heap(const float* scores, int64_t n, int64_t exclude,
                         int64_t topn, int32_t* out_idx) {
    if (topn <= 0) return 0; /* the reference segfaults here */
    int64_t cap = topn < n ? topn : n;
    rec_t* heap = (rec_t*)malloc(sizeof(rec_t) * (size_t)(cap > 0 ? cap : 1));
    int64_t size = 0;
    for (int64_t i = 0; i < n; ++i) {          /* Recommender.cu:295 */
        if (i == exclude) continue;            /* :296 */
        rec_t r = {(int32_t)i, scores[i]};     /* :298 */
        if (size < topn) {                     /* :300 */
            pq_push(heap, &size, r);
        } else if (r.sim > heap[0].sim) {      /* :302 */
            pq_pop(heap, &size);
            pq_push(heap, &size, r);
        }
    }
    int64_t count = size;                      /* :309-315 drain + reverse */
    for (int64_t k = count - 1; k >= 0; --k) {
        out_idx[k] = heap[0].idx;
        pq_pop(heap, &size);
    }
    free(heap);
    return count;
}

/* ---- canonical top-N: (score desc, index asc) ---------------------------- */

/* "a ranks before b" */
static inline int canon_before(float sa, int32_t ia, float sb, int32_t ib) {
    if (sa > sb) return 1;
    if (sa < sb) return 0;
    return ia < ib;
}

/* binary min-heap on canonical rank: root = worst kept candidate */
static void canon_sift_down(rec_t* h, int64_t size, int64_t i) {
    for (;;) {
        int64_t l = 2 * i + 1, r = l + 1, w = i;
        if (l < size && canon_before(h[w].sim, h[w].idx, h[l].sim, h[l].idx)) w = l;
        if (r < size && canon_before(h[w].sim, h[w].idx, h[r].sim, h[r].idx)) w = r;
        if (w == i) return;
        rec_t t = h[i]; h[i] = h[w]; h[w] = t;
        i = w;
    }
}

static void canon_sift_up(rec_t* h, int64_t i) {
    while (i > 0) {
        int64_t p = (i - 1) / 2;
        if (!canon_before(h[p].sim, h[p].idx, h[i].sim, h[i].idx)) return;
        rec_t t = h[i]; h[i] = h[p]; h[p] = t;
        i = p;
    }
}

static int canon_cmp_qsort(const void* a, const void* b) {
    const rec_t* x = (const rec_t*)a;
    const rec_t* y = (const rec_t*)b;
    if (canon_before(x->sim, x->idx, y->sim, y->idx)) return -1;
    if (canon_before(y->sim, y->idx, x->sim, x->idx)) return 1;
    return 0;
}

/* scan [lo,hi) of scores (global index = base + i) into a canonical heap */
static int64_t canon_scan(const float* scores, int64_t lo, int64_t hi,
                          int64_t exclude, int64_t topn, rec_t* heap) {
    int64_t size = 0;
    for (int64_t i = lo; i < hi; ++i) {
        if (i == exclude) continue;
        float s = scores[i];
        if (size < topn) {
            heap[size].idx = (int32_t)i;
            heap[size].sim = s;
            canon_sift_up(heap, size);
            size++;
        } else if (canon_before(s, (int32_t)i, heap[0].sim, heap[0].idx)) {
            heap[0].idx = (int32_t)i;
            heap[0].sim = s;
            canon_sift_down(heap, size, 0);
        }
    }
    return size;
}

int64_t oracle_topn_canonical(const float* scores, int64_t n, int64_t exclude,
                              int64_t topn, int32_t* out_idx, float* out_score) {
    if (topn <= 0) return 0;
    int64_t cap = topn < n ? topn : n;
    rec_t* heap = (rec_t*)malloc(sizeof(rec_t) * (size_t)(cap > 0 ? cap : 1));
    int64_t size = canon_scan(scores, 0, n, exclude, topn, heap);
    qsort(heap, (size_t)size, sizeof(rec_t), canon_cmp_qsort);
    for (int64_t k = 0; k < size; ++k) {
        out_idx[k] = heap[k].idx;
        if (out_score) out_score[k] = heap[k].sim;
    }
    free(heap);
    return size;
}

/* ---- recommendByIndex: Recommender.cu:275-318 ---------------------------- */

int64_t oracle_recommend_by_index(const float* feats, int64_t stride, int64_t n,
                                  int64_t song_index, int64_t topn,
                                  int32_t* out_idx, float* scratch_scores) {
    if (song_index < 0 || song_index >= n) return -1; /* :281-284 */
    float* scores = scratch_scores;
    if (!scores) scores = (float*)calloc((size_t)n, sizeof(float)); /* :287 */
    oracle_scores(feats, stride, n, feats + song_index * stride, scores); /* :290 */
    int64_t c = oracle_topn_heap(scores, n, song_index, topn, out_idx);
    if (!scratch_scores) free(scores);
    return c;
}

int64_t oracle_recommend_omp(const float* feats, int64_t stride, int64_t n,
                             int64_t song_index, int64_t topn, int32_t* out_idx,
                             float* out_score, int threads) {
    if (song_index < 0 || song_index >= n || topn <= 0) return -1;
    if (threads <= 0) threads = oracle_max_threads();
    const float* q = feats + song_index * stride;
    const float qn = query_norm(q);
    rec_t* all = (rec_t*)malloc(sizeof(rec_t) * (size_t)(topn * threads));
    int64_t* sizes = (int64_t*)calloc((size_t)threads, sizeof(int64_t));
#pragma omp parallel num_threads(threads)
    {
#ifdef _OPENMP
        int t = omp_get_thread_num();
        int nt = omp_get_num_threads();
#else
        int t = 0, nt = 1;
#endif
        int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        rec_t* heap = all + (int64_t)t * topn;
        int64_t size = 0;
        for (int64_t i = lo; i < hi; ++i) {
            if (i == song_index) continue;
            float s = one_score(q, qn, feats + i * stride);
            if (size < topn) {
                heap[size].idx = (int32_t)i;
                heap[size].sim = s;
                canon_sift_up(heap, size);
                size++;
            } else if (canon_before(s, (int32_t)i, heap[0].sim, heap[0].idx)) {
                heap[0].idx = (int32_t)i;
                heap[0].sim = s;
                canon_sift_down(heap, size, 0);
            }
        }
        sizes[t] = size;
    }
    /* compact + sort the per-thread survivors */
    int64_t total = 0;
    for (int t = 0; t < threads; ++t) {
        if (sizes[t] && total != (int64_t)t * topn)
            memmove(all + total, all + (int64_t)t * topn, sizeof(rec_t) * (size_t)sizes[t]);
        total += sizes[t];
    }
    qsort(all, (size_t)total, sizeof(rec_t), canon_cmp_qsort);
    int64_t c = total < topn ? total : topn;
    for (int64_t k = 0; k < c; ++k) {
        out_idx[k] = all[k].idx;
        if (out_score) out_score[k] = all[k].sim;
    }
    free(all);
    free(sizes);
    return c;
}

/* ---- std::mt19937 + uniform_real_distribution<float>(0,1), libstdc++ ------ */

void oracle_mt19937_uniform(uint32_t seed, int64_t count, float* out) {
    uint32_t mt[624];
    int pos = 624;
    mt[0] = seed;
    for (int i = 1; i < 624; ++i)
        mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    for (int64_t c = 0; c < count; ++c) {
        if (pos >= 624) {
            for (int i = 0; i < 624; ++i) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            pos = 0;
        }
        uint32_t y = mt[pos++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        /* generate_canonical<float,24>: one 32-bit draw, float arithmetic */
        float r = (float)y / 4294967296.0f;
        if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
        out[c] = r;
    }
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
