/*
 * cosine_oracle.h — CPU restatement of the reference's cosine top-N path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product path (spotify_recommender_amd/, include/) never links or calls it.
 *
 * Pinning status (see oracle/README.md): the SCORING chain is PARITY UNPINNED
 * by the strict rule — no vector made by a reference build, or held by the
 * reference's tests, exists for it:
 *   - The reference hot-path file (Recommender.cu) cannot be compiled in the
 *     build container without stand-ins for CUDA headers, so there is no
 *     oracle/_ref build of it, and the reference ships no tests or vectors.
 *   - The restatement reproduces the reference outputs recorded in
 *     SURVEY.md §8(c)/§6.2 (tie-order fixture, zero-query fixture, 1 M and
 *     10 M mt19937(12345) top-3 ids + scores) — tests/test_oracle_pins.py —
 *     which the survey took from a stub-header build: agreement, not a pin.
 *   - The heap replay is pinned against the real libstdc++
 *     std::priority_queue (tests/heap_check.cpp).
 *
 * All citations are file:line in /root/reference.
 */
#ifndef COSINE_ORACLE_H
#define COSINE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_FEATURE_COUNT 12 /* Song.h:12 */

/* Recommender.cu:256-273 (calculateSimilaritiesCPU).  `stride` is the distance
 * between consecutive rows in floats: 12 for the dense row-major matrix, 38 for
 * the reference's AoS vector<Song> (sizeof(Song)==152).  Strictly sequential
 * j=0..11, separate multiply and add roundings (built with -ffp-contract=off,
 * no -march, no -ffast-math — Makefile:9 of the reference). */
void oracle_scores(const float* feats, int64_t stride, int64_t n,
                   const float* query, float* out);

/* Same arithmetic, OpenMP `parallel for` over rows (BASELINE.md §3 "B1"). */
void oracle_scores_omp(const float* feats, int64_t stride, int64_t n,
                       const float* query, float* out, int threads);

/* Recommender.cu:293-315 + Recommender.h:12-22: ascending scan, skip
 * i==exclude, size-topn min-heap via the libstdc++ push_heap/pop_heap
 * algorithms (restated), replace only on strict '>', drain + reverse.
 * Returns the number of indices written (<= topn).  topn must be > 0. */
int64_t oracle_topn_heap(const float* scores, int64_t n, int64_t exclude,
                         int64_t topn, int32_t* out_idx);

/* Canonical order used by the product: (score desc, index asc).  -0.0 and
 * +0.0 compare equal, as they do under the reference's float '>'.  */
int64_t oracle_topn_canonical(const float* scores, int64_t n, int64_t exclude,
                              int64_t topn, int32_t* out_idx, float* out_score);

/* Recommender.cu:275-318 (recommendByIndex) over a dense/strided matrix.
 * Returns count, or -1 for an invalid index (the reference returns {}). */
int64_t oracle_recommend_by_index(const float* feats, int64_t stride, int64_t n,
                                  int64_t song_index, int64_t topn,
                                  int32_t* out_idx, float* scratch_scores);

/* B1 baseline: OpenMP scores + per-thread canonical top-N + merge. */
int64_t oracle_recommend_omp(const float* feats, int64_t stride, int64_t n,
                             int64_t song_index, int64_t topn, int32_t* out_idx,
                             float* out_score, int threads);

/* std::mt19937 + libstdc++ uniform_real_distribution<float>(0,1) restated:
 * fills `count` floats in call order.  Used to regenerate the SURVEY §8(c)
 * catalogues (seed 12345, row-major, i outer / j inner). */
void oracle_mt19937_uniform(uint32_t seed, int64_t count, float* out);

int oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
