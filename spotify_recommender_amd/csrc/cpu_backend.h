// cpu_backend.h — the product's own CPU backend, for hosts WITHOUT a HIP device.
//
// The reference degrades to a CPU loop when it finds no GPU ("No CUDA devices found. Falling back to CPU",
// Recommender.cu:117-127,176-181) and answers with calculateSimilaritiesCPU (:256-273) + the host heap of
// recommendByIndex (:293-315).  BASELINE configs[0] ("114 k tracks, CPU/OpenMP cosine path only") is that path.
// This file is its counterpart behind mi355rec_create_sharded / mi355rec_create_placed: the same arithmetic —
// strictly sequential j = 0..11, multiply and add rounded separately, IEEE sqrt and divide, `> 1e-8f` on
// sqrt(norm) * qnorm, the std::min / std::max clamp — over the dense row-major matrix, rows split over OpenMP
// threads with a top-N per thread and one merge.  Results are in the engine's canonical order (score descending,
// row ascending; -0.0f ranks and reports as +0.0f), so a caller cannot tell the backends apart by their answers.
//
// It is taken ONLY when hipGetDeviceCount reports no device at all: a host with a GPU never lands here, the
// single-device C-ABI (mi355rec_create) keeps failing with MI355REC_ERR_NO_DEVICE, and nothing under oracle/
// (test infrastructure) is included, linked or loaded by it.  Compiled by g++ with -O3 -ffp-contract=off and
// without -ffast-math / -march (the reference's own flags, Makefile:9): the sums stay unfused and in order.
#pragma once

#include <cstdint>

namespace mi355cpu {

struct Catalogue;

// Copies the n x 12 row-major matrix (the reference deep-copies its songs as well, Recommender.cu:109).
// threads = 0: what OpenMP would use (OMP_NUM_THREADS, else every core the process may run on).
Catalogue* create(const float* feats_rowmajor, int64_t n, int threads);
void destroy(Catalogue* c);
int64_t rows(const Catalogue* c);
int threads(const Catalogue* c);
const float* row(const Catalogue* c, int64_t r);

// calculateSimilaritiesCPU (Recommender.cu:256-273): out_n[i] = cosine(q12, row i).
void scores(const Catalogue* c, const float* q12, float* out_n);

// recommendByIndex's selection (Recommender.cu:293-315) in canonical order: the best min(topn, rows - [exclude is
// a row]) rows, `exclude` (-1: none) skipped by INDEX.  Returns how many were written.
int topn(const Catalogue* c, const float* q12, int64_t exclude, int topn, int64_t* out_idx, float* out_score);

}  // namespace mi355cpu

// ---- the whole "sharded" handle on a host without a device --------------------------------------------------------
// What mi355rec_create_sharded / _placed return there: the synchronous calls and the ticketed stream of
// include/mi355rec.h ("row-sharded catalogue"), served by the catalogue above.  The stream keeps the GPU path's
// bookkeeping — tickets, windows, a ring of 4 windows of results, flushes that round the next ticket up to a window
// boundary — so that a serving loop written against the C-ABI runs unchanged; a query is simply computed when it is
// enqueued.  Return values are the C-ABI's codes (0 = ok, MI355REC_ERR_INVALID_ARG); *why receives a message.
namespace mi355cpu {

struct Node;
Node* node_create(const float* feats_rowmajor, int64_t n);
void node_destroy(Node* h);
const Catalogue* node_catalogue(const Node* h);
int node_query(Node* h, const float* q12, int64_t exclude, int topn, int64_t* out_idx, float* out_score, int* out_count,
               const char** why);
int node_set_window(Node* h, int window, const char** why);
int node_enqueue(Node* h, const float* q12, int64_t exclude, int topn, int64_t* ticket, const char** why);
int node_flush(Node* h);
int node_wait(Node* h, int64_t ticket, int64_t* out_idx, float* out_score, int* out_count, const char** why);
void node_stream_stats(const Node* h, int64_t* queries, int64_t* windows, int64_t* host_ns);

}  // namespace mi355cpu
