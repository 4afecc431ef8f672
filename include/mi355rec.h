/*
 * mi355rec.h — C-ABI of the MI355X-native cosine top-N engine.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (Iamdarika/Spotify_recommender) has no FFI layer: its boundary is the public
 * surface of `class Recommender` (Recommender.h:28-83) whose private
 * `calculateSimilarities` (Recommender.h:114, Recommender.cu:184-254) drives
 * cuBLAS SGEMV + two CUDA kernels and whose `recommendByIndex`
 * (Recommender.cu:275-318) does a host heap top-N.  Each entry point below
 * names the reference interface it replaces.  The C++ shim in
 * include/Recommender.h + spotify_recommender_amd/csrc/Recommender.cpp keeps
 * the reference's class API on top of these calls; INTEGRATION.md shows the
 * binding a reference maintainer would add.
 *
 * Conventions: plain pointers and sizes only; 0 = success, negative = error
 * (never throws across the boundary); the caller owns every in/out buffer; the
 * library owns device memory it allocates.  A handle is not re-entrant: use it
 * from one host thread at a time (as the reference's Recommender).
 * Streams: all launches of one handle share its scratch buffers.  Calls on the
 * same stream are ordered by the stream; when the stream changes from one call
 * to the next (including between the asynchronous API and the synchronous one,
 * which runs on a private stream) the library inserts an event so the later
 * call waits for the earlier one.  For CONCURRENT queries use one handle per
 * stream (handles may share one device matrix through mi355rec_create_device).
 * A stream passed to a call must stay alive until the handle's next call.
 * Without a gfx950 device every single-device call (mi355rec_create*) fails with
 * MI355REC_ERR_NO_DEVICE; the node-level handle (mi355rec_create_sharded / _placed,
 * what the C++ Recommender sits on) is then served by the product's own CPU backend,
 * as the reference falls back to its CPU loop — see "NO HIP DEVICE" further down.
 *
 * Numerics: scores are bit-identical to the reference's CPU path
 * (calculateSimilaritiesCPU, Recommender.cu:256-273): strictly sequential
 * j=0..11 fp32 multiply-then-add (no FMA), IEEE sqrt and divide, threshold
 * 1e-8f on sqrt(norm)*qnorm, clamp with std::min/std::max semantics.
 * Top-N order is canonical: score descending, then row index ascending
 * (-0.0f is ranked and reported as +0.0f).  The reference's order inside runs
 * of exactly equal scores is a libstdc++ heap artefact.
 */
#ifndef MI355REC_H
#define MI355REC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355REC_DIM 12 /* Song.h:12 FEATURE_COUNT */
#define MI355REC_MAX_TOPN_FAST 1024 /* larger topn is served in rounds of this many (one scan each) */

#define MI355REC_OK 0
#define MI355REC_ERR_INVALID_ARG (-1)
#define MI355REC_ERR_NO_DEVICE (-2)
#define MI355REC_ERR_HIP (-3)
#define MI355REC_ERR_OUT_OF_MEMORY (-4)

typedef struct mi355rec mi355rec_t;

/* Packed candidate key: high 32 bits = order-preserving image of the fp32
 * score, low 32 bits = ~(global row index).  Larger key = better candidate in
 * canonical order.  0 = empty slot.  This is what crosses xGMI in the
 * multi-GPU merge (one all-gather of topn keys per rank). */
typedef uint64_t mi355rec_key_t;

typedef struct {
    int64_t rows;              /* rows held by this handle (local shard)      */
    int64_t row_base;          /* global index of local row 0                 */
    int32_t device;            /* HIP device ordinal                          */
    int32_t compute_units;     /* CUs of the device                           */
    int32_t grid_blocks;       /* resident workgroups of the streaming kernel */
    int32_t block_threads;
    int64_t bytes_per_query;   /* algorithmic bytes of one pass: rows * 48    */
    float last_scan_ms;        /* HIP-event time of the last timed scan       */
    float last_merge_ms;       /* HIP-event time of the last timed merge      */
    float last_pass_ms;        /* HIP-event time of the batched path's passes (mean of pass 1 and pass 2) */
    int32_t batched_grid_blocks; /* workgroups of a batched pass (0 before the first batched call) */
    float batched_margin;      /* error bound the fp16 pre-filter runs with: 1.0e-3 where the device keeps
                                  fp16 subnormals (checked on first use), 1.5e-3 otherwise */
    int64_t replica_bytes_per_query; /* algorithmic bytes of one pass over the fp16 replica: ceil(rows/2) * 48
                                        (0 = the handle has no replica)                     */
    int32_t replica_active;    /* 1 = single queries currently scan the replica (mi355rec_set_replica) */
    int32_t replica_grid_blocks; /* resident workgroups of the replica scan                 */
    float replica_build_ms;    /* device time of building the replica (once, at create)     */
    float replica_margin_single; /* error bound the replica pre-filters claim on this device: 1.0e-3 where the unit  */
    float replica_margin_multi;  /* demonstrably keeps fp16 subnormals (checked when the replica is built), else 1.5e-3:
                                    v_fma_mix_f32 (fp16 single-query scan) / the matrix core (multi-query pass)        */
    int64_t replica_single_bytes_per_query; /* algorithmic bytes of one SINGLE-query scan over the replica it currently
                                    uses: ceil(rows/4) * 48 over the 8-bit one, ceil(rows/2) * 48 over the fp16 one   */
    int32_t replica_single_row_bytes; /* 12 (8-bit replica), 24 (fp16 replica, MI355REC_REPLICA_FP16) or 0 (no replica)  */
    int32_t lone_fused_queries; /* synchronous single queries served by ONE scan launch that also merged and raised the
                                    completion word (8-bit replica, shards of >= 4 M rows), since create */
    /* WHICH ROUTE the work took, counted per launch since create (DESIGN.md has the table "AUTO route by rows, batch,
     * topn"; tests/test_gpu_routes.py asserts the counter that moves for each cell): */
    int64_t route_fp32;          /* single query: scan over the fp32 rows (48 B/row), plain or streamed             */
    int64_t route_fp16;          /* single query: scan over the fp16 replica (24 B/row; MI355REC_REPLICA_FP16)      */
    int64_t route_q8;            /* single query: scan over the 8-bit replica (12 B/row), plain or streamed         */
    int64_t route_q8_lone;       /* ... of a lone synchronous query that also merged and signalled (one launch)     */
    int64_t route_multi_fp32;    /* exact multi-query pass over the fp32 rows (<= 12 queries per pass)              */
    int64_t route_multi_fp16;    /* multi-query pass over the fp16 replica, fp16 matrix-core pre-filter (<= 32)     */
    int64_t route_multi_q8;      /* multi-query pass over the 8-bit replica, integer matrix-core pre-filter          */
    int64_t route_mfma_two_pass; /* chunks (<= 1024 queries) of the two-pass batched matrix-core path              */
    int64_t route_exact_queue;   /* queries that path handed to the exact scan on the device (reading it synchronises) */
    int32_t device_bytes_per_row; /* what the handle keeps resident per row: 48 (fp32) + 24 (fp16 replica) + 12 (8-bit) */
} mi355rec_stats_t;

/* What this build of the library was compiled with.  The product build returns 0.  The tools/ scripts build
 * instrumented copies under gpurun_out/ (never the product library): MI355REC_BUILD_EXPERIMENTS = environment knobs for
 * A/B runs and the routes that only exist for A/B (single queries over the fp16 replica: MI355REC_REPLICA_FP16; the
 * 8-bit front end of the multi-query pass: MI355REC_BATCH_Q8); MI355REC_BUILD_PHASE_CLOCK = per-workgroup phase stamps
 * (csrc/experiments.hip.h). */
#define MI355REC_BUILD_EXPERIMENTS 1
#define MI355REC_BUILD_PHASE_CLOCK 2
int mi355rec_build_flags(void);

/* Number of visible HIP devices (0 when there is none / no driver). */
int mi355rec_device_count(void);

/* Thread-local text of the last error raised with no handle to attach it to
 * (e.g. a failed create); with a handle, use mi355rec_last_error. */
const char* mi355rec_last_global_error(void);

/* Replaces Recommender::initialize's device half (Recommender.cu:155-168:
 * cudaMalloc + flatten + cudaMemcpy H2D).  `feats_host` is the row-major
 * n x 12 fp32 matrix (the reference flattens vector<Song> into exactly this,
 * Recommender.cu:162-167).  The matrix is copied to `device` once.
 * `row_base` is the global index of row 0 (0 unless this is one shard of a
 * row-sharded catalogue).  dim must be 12.  n in [0, 2^32-2]; n == 0 is an
 * EMPTY SHARD (feats may be NULL): its queries return all-empty lists, so a
 * rank that owns no rows still takes part in the merge. */
int mi355rec_create(const float* feats_host, int64_t n, int dim, int device,
                    int64_t row_base, mi355rec_t** out);

/* Same, over a matrix that is ALREADY resident in device memory (e.g. a torch
 * tensor's data_ptr).  The memory is borrowed, not copied, and must outlive
 * the handle; it must be 16-byte aligned.  The call synchronises the device
 * once, so a matrix still being produced on another stream is complete before
 * the first query; ordering LATER writes to it is the caller's business. */
int mi355rec_create_device(const float* feats_dev, int64_t n, int dim,
                           int device, int64_t row_base, mi355rec_t** out);

/* The same with FLAGS:
 *   MI355REC_CREATE_NO_REPLICA   the handle keeps the fp32 rows only: 48 B per row resident instead of 84 (no fp16 and no
 *                                8-bit copy is built, single queries scan the fp32 rows, batches of 13 and more take the
 *                                matrix-core path with rows from the fp32 matrix; mi355rec_set_replica(ON) is refused).
 *                                mi355rec_stats_t::device_bytes_per_row says what a handle holds. */
#define MI355REC_CREATE_NO_REPLICA 1
int mi355rec_create_ex(const float* feats_host, int64_t n, int dim, int device, int64_t row_base, int flags,
                       mi355rec_t** out);
int mi355rec_create_device_ex(const float* feats_dev, int64_t n, int dim, int device, int64_t row_base, int flags,
                              mi355rec_t** out);

/* THE fp16 REPLICA.  Next to the fp32 rows every handle keeps a second copy of
 * its shard that is only good enough to rule rows OUT: each row L2-normalised
 * and rounded to fp16, 24 B per row (+50 % device memory, built once inside
 * create; csrc/replica.hip.h).  A single query then streams 24 B per row
 * instead of 48: a row is skipped when its fp16 cosine is more than the derived
 * error bound (1.5e-3) below what the top-N needs; every row that is not
 * skipped is fetched from the fp32 matrix and scored by the reference's exact
 * chain, so ids, order and score bits are those of the fp32 scan and of
 * Recommender.cu:256-318.  Rows or queries the bound cannot be claimed for
 * (zero / tiny / huge / non-finite norms) are always scored exactly.
 *   AUTO (default): shards of >= 1 M rows scan the replica, smaller ones the
 *                   fp32 rows (a query is launch-bound there either way);
 *   OFF:            always the fp32 rows (the reference's own traffic, 48 B/row);
 *   ON:             always the replica;
 *   FP16:           as ON, with single queries on the fp16 replica (see below).
 * A handle with a replica holds two encodings of the normalised rows: fp16
 * (24 B/row, csrc/replica.hip.h: what the multi-query and batched passes
 * read, error bound 1.0e-3) and 8-bit (12 B/row, csrc/replica_q8.hip.h: what
 * single queries scan; the query stays fp32, the bound is per query,
 * l1(q/|q|)/254 + 3e-5 <= 0.0137).  Both are pre-filters in front of the same
 * exact chain; MI355REC_REPLICA_FP16 exists for A/B measurements.
 * The batched matrix-core path reads its rows from the replica too (they are
 * stored in exactly the form its MFMA operand wants) unless the mode is OFF.
 * The score vector (mi355rec_scores*), rounds of topn > 1024 after the first
 * and the exact multi-query pass always read the fp32 rows.
 * MI355REC_CREATE_NO_REPLICA (mi355rec_create_ex) creates a handle without either.
 * A replica is a SNAPSHOT: if the caller overwrites a borrowed matrix
 * (mi355rec_create_device) while the handle lives, it must call
 * mi355rec_rebuild_replica before the next query (synchronous). */
#define MI355REC_REPLICA_AUTO 0
#define MI355REC_REPLICA_OFF 1
#define MI355REC_REPLICA_ON 2
#define MI355REC_REPLICA_FP16 3
int mi355rec_set_replica(mi355rec_t* h, int mode);
int mi355rec_rebuild_replica(mi355rec_t* h);
/* Diagnostics, cumulative since create (synchronises the device): scans that
 * went over the replica, and rows those scans fetched from the fp32 matrix for
 * the exact chain.  Either pointer may be NULL. */
int mi355rec_replica_counters(mi355rec_t* h, int64_t* scans, int64_t* rescored_rows);

/* LANES: another handle over the SAME device rows and the same replicas (nothing is copied or rebuilt: a lane costs its
 * per-workgroup lists and sample buffers, ~20 MB at 10 M rows), with its own stream state — its own chain of streamed
 * queries, its own stash, its own hand-off buffers.  A handle is a single chain: query k + 1's launch carries query k's
 * merge and needs query k's sample, so consecutive launches of ONE handle never overlap and the chip idles while one
 * drains and the next ramps up (a streamed query at 10 M rows: 24.3 us per query where its kernel has the bandwidth for
 * 19).  Two lanes on two streams, queries dealt alternately, fill each other's ramps: 17-18 us per query on the same GPU
 * (bench.py --lanes).  Results per lane are exactly those of the parent (same kernels, same rows).
 * The reference has nothing like it (one query per process, Recommender.cu:275-318); mi355rec_sharded_* with
 * MI355REC_PLACEMENT_REPLICATED is the same idea across devices.
 * Rules: a lane is used from one thread at a time like any handle, on a stream of its own; the parent and its lanes may be
 * destroyed in any order (the rows and replicas go with the last one); mi355rec_rebuild_replica is refused while lanes
 * exist; a lane of a lane is a lane of the same group. */
int mi355rec_create_lane(mi355rec_t* parent, mi355rec_t** out);
/* The HIP stream (hipStream_t) the library created with the handle, for callers that want to enqueue on it: THE stream to
 * run a lane on.  A process has few hardware queues (four by default on ROCm) and a stream is bound to one when it is
 * created; two lanes whose streams share a hardware queue do not overlap at all (measured: two streams taken from a
 * framework's pool after the handles existed landed on ONE queue and the lanes ran at a single handle's rate).  The
 * parent's own stream and a lane's are created one after the other and sit on different queues.  The synchronous calls
 * of the handle use this stream too: work enqueued on it by the caller is ordered with them.  It was created with
 * hipStreamNonBlocking: it does NOT wait for work on the null stream (e.g. a memset of the buffer a query writes to —
 * synchronise or use an event first). */
void* mi355rec_own_stream(mi355rec_t* h);
/* What mi355rec_create_lane found when it chose the lane's stream: it times a small kernel on the parent's stream alone and
 * on both streams at once, and replaces the lane's stream (each new stream is bound to the next hardware queue) until the pair
 * runs side by side.  *stream_attempts = streams tried (0 on a handle that is not a lane); *overlaps_parent = 1 side by side,
 * 0 none of them did (the lane then buys nothing: use one handle), -1 not testable (under 10 000 rows). */
int mi355rec_lane_status(const mi355rec_t* h, int* stream_attempts, int* overlaps_parent);

/* Replaces Recommender::~Recommender (Recommender.cu:86-98). */
void mi355rec_destroy(mi355rec_t* h);

const char* mi355rec_last_error(const mi355rec_t* h);

int mi355rec_stats(const mi355rec_t* h, mi355rec_stats_t* out);
/* The same for a caller that may have been built against an EARLIER header: mi355rec_stats_t only ever grows at its end, and
 * this copies min(out_size, sizeof(mi355rec_stats_t)) bytes — a shorter struct gets the fields it knows, never an overrun.
 * Pass sizeof(mi355rec_stats_t) of the header you compiled with.  *written (may be NULL) = the bytes copied. */
int mi355rec_stats_sized(const mi355rec_t* h, void* out, size_t out_size, size_t* written);

/* ---- synchronous host API (what the C++ Recommender shim calls) ---------- */

/* Replaces the private Recommender::calculateSimilarities(int, float*)
 * (Recommender.h:114, Recommender.cu:184-254): cosine of catalogue row
 * `local_row` against every local row, n floats written to host memory. */
int mi355rec_scores_row(mi355rec_t* h, int64_t local_row, float* out_host);

/* Same for an arbitrary query vector (12 floats, host). */
int mi355rec_scores(mi355rec_t* h, const float* query12, float* out_host);

/* Replaces Recommender::recommendByIndex (Recommender.cu:275-318): query =
 * local row `local_row`, that row excluded by index, best `topn` returned as
 * global row indices (best first) with their scores.  *out_count =
 * min(topn, rows-1).  topn <= 0 is MI355REC_ERR_INVALID_ARG (the reference
 * segfaults); a bad row is MI355REC_ERR_INVALID_ARG (the
 * reference prints and returns {}, Recommender.cu:281-284). */
int mi355rec_query_row_topn(mi355rec_t* h, int64_t local_row, int topn,
                            int64_t* out_idx, float* out_score, int* out_count);

/* Arbitrary query vector; `exclude_global` = global row index to skip or -1. */
int mi355rec_query_topn(mi355rec_t* h, const float* query12,
                        int64_t exclude_global, int topn, int64_t* out_idx,
                        float* out_score, int* out_count);

/* `batch` independent queries (batch x 12 floats); exclude may be NULL.
 * Outputs are batch x topn, each row padded with idx -1 / score 0;
 * out_count has `batch` entries.  With batch > 1 and topn <= 128 the queries
 * run as multi-query passes (12 queries per scan of the catalogue). */
int mi355rec_query_batch_topn(mi355rec_t* h, const float* queries, int batch,
                              const int64_t* exclude_global, int topn,
                              int64_t* out_idx, float* out_score, int* out_count);

/* ---- asynchronous device API (serving loop, bench.py, multi-GPU) ---------
 * Everything below only ENQUEUES work on `stream` (a hipStream_t; NULL = the
 * device's default stream) and returns; outputs live in caller-provided
 * device memory.  No allocation, no synchronisation (hipGraph-capturable). */

/* Streaming scan of the local shard for one query: writes the shard's best
 * `topn` candidates as `topn` packed keys (sorted descending, 0-padded) to
 * out_keys_dev.  query = local row. */
int mi355rec_enqueue_row_keys(mi355rec_t* h, int64_t local_row, int topn,
                              mi355rec_key_t* out_keys_dev, void* stream);

/* Same for a host query vector (passed by value to the kernel). */
int mi355rec_enqueue_query_keys(mi355rec_t* h, const float* query12,
                                int64_t exclude_global, int topn,
                                mi355rec_key_t* out_keys_dev, void* stream);

/* STREAMED single queries: as mi355rec_enqueue_row_keys / _query_keys (topn <= 1024), but
 * the device merge of this query is deferred — it runs inside the scan launch of the NEXT
 * streamed query on this handle (one extra workgroup merges the previous query's
 * per-workgroup lists while the others scan), or in mi355rec_enqueue_flush.  On a shard
 * that scans its fp16 replica the stream additionally runs ONE CALL BEHIND: call k + 1
 * LAUNCHES the scan of query k, and a few workgroups of that launch take the sample that
 * seeds query k + 1's cutoff, so no query but the first needs a seed launch of its own
 * (the launch goes to the stream of the call that issues it; the query vector is copied
 * by the call that passes it).  The keys of query k are therefore complete when the work
 * enqueued by streamed call k + 2 (k + 1 over the fp32 rows), or by the flush, has
 * completed; out_keys_dev must stay valid until then — in practice: flush, then
 * synchronise, then read.  A stream of K queries costs K scan launches + ONE merge launch
 * (+ one seed launch) instead of 3 K, which removes the ~10 us one-workgroup merge kernel
 * and the ~5 us seed kernel from every step. */
int mi355rec_enqueue_row_keys_streamed(mi355rec_t* h, int64_t local_row, int topn,
                                       mi355rec_key_t* out_keys_dev, void* stream);
int mi355rec_enqueue_query_keys_streamed(mi355rec_t* h, const float* query12,
                                         int64_t exclude_global, int topn,
                                         mi355rec_key_t* out_keys_dev, void* stream);
/* Merges the last streamed query (no-op when nothing is pending). */
int mi355rec_enqueue_flush(mi355rec_t* h, void* stream);

/* Queries whose 12 floats ALREADY LIVE in memory this device can read: a resident
 * row of this handle, a row of ANOTHER shard on another GPU of the node (through
 * the peer mapping: how the row-sharded engine below hands a catalogue row to every
 * shard without a host round trip), a staged vector in mapped host memory.  The
 * kernels fetch them with scalar loads when they start; the 48 bytes must not
 * change until the query has completed.  mi355rec_row_ptr returns where a resident
 * row lives (valid for the lifetime of the handle).  Otherwise as
 * mi355rec_enqueue_query_keys / _query_keys_streamed; out_idx_dev / out_score_dev
 * may be NULL. */
int mi355rec_row_ptr(mi355rec_t* h, int64_t local_row, const float** out_dev);
int mi355rec_enqueue_ptr_keys(mi355rec_t* h, const float* query12_dev, int64_t exclude_global, int topn,
                              mi355rec_key_t* out_keys_dev, int64_t* out_idx_dev, float* out_score_dev,
                              void* stream);
int mi355rec_enqueue_ptr_keys_streamed(mi355rec_t* h, const float* query12_dev, int64_t exclude_global,
                                       int topn, mi355rec_key_t* out_keys_dev, void* stream);

/* `batch` queries (batch x 12 floats, host; exclude_global may be NULL).
 * On a shard with a replica (>= 65536 rows, topn <= 128) 2 ... 32 queries go in
 * ONE multi-query pass over the fp16 replica (csrc/replica_multi.hip.h: fp16
 * matrix-core pre-filter for up to 32 queries per 24 B/row pass, candidates
 * resolved by the exact chain in the same launch; 2 queries: over the 8-bit
 * replica), 33 and more take the two-pass
 * batched matrix-core path (mi355rec_set_batch_path forces either); without a
 * replica up to 12 queries — or any batch on a shard below 65536 rows — go in
 * exact multi-query passes over the fp32 rows (12 queries per pass; topn > 128
 * falls back to one scan per query).  Writes batch x topn packed keys (each row sorted descending,
 * 0-padded). */
int mi355rec_enqueue_batch_keys(mi355rec_t* h, const float* queries,
                                const int64_t* exclude_global, int batch, int topn,
                                mi355rec_key_t* out_keys_dev, void* stream);

/* A STREAM OF BATCHES: as mi355rec_enqueue_batch_keys, deferred like the streamed single queries.  On a
 * shard with a replica (>= 65536 rows, topn <= 128) the stream runs one call behind: call k + 1 LAUNCHES
 * batch k (groups of up to 32 queries), and that one launch carries, beside its scanners, a merging
 * workgroup per query of batch k - 1 and a few workgroups that take the sample of batch k + 1 — K batches
 * cost K launches + one sample launch at the head + one merge launch at the tail (mi355rec_enqueue_flush)
 * instead of 3 K.  The keys of a batch are complete when the work of the SECOND streamed batch call after it,
 * or of the flush, has completed; out_keys_dev (batch x topn) must stay valid until then.  Elsewhere the
 * batch is served at once.  Streams of single queries and of batches on one handle close each other. */
int mi355rec_enqueue_batch_keys_streamed(mi355rec_t* h, const float* queries, const int64_t* exclude_global,
                                         int batch, int topn, mi355rec_key_t* out_keys_dev, void* stream);

/* A batch whose queries are vectors (queries[i*12 ..], host) and / or POINTERS to 12 floats in
 * device-readable memory (query_ptrs_dev[i] != NULL wins; either array may be NULL when the other
 * covers every query): what a window of the row-sharded stream hands to every shard.  With a
 * replica (mi355rec_batch_pointers_ok: shard of >= 65536 rows, topn <= 128, sample large enough)
 * the batch goes in multi-query passes over the replica, up to 32 queries per pass
 * (csrc/replica_multi.hip.h); otherwise one scan per query.  topn <= 1024. */
int mi355rec_batch_pointers_ok(const mi355rec_t* h, int topn);
int mi355rec_enqueue_batch_mixed_keys(mi355rec_t* h, const float* queries, const float* const* query_ptrs_dev,
                                      const int64_t* exclude_global, int batch, int topn,
                                      mi355rec_key_t* out_keys_dev, void* stream);
/* ... and as a stream of batches (mi355rec_enqueue_batch_keys_streamed's deferred completion). */
int mi355rec_enqueue_batch_mixed_keys_streamed(mi355rec_t* h, const float* queries, const float* const* query_ptrs_dev,
                                               const int64_t* exclude_global, int batch, int topn,
                                               mi355rec_key_t* out_keys_dev, void* stream);

/* The same for queries that are ALREADY in device memory (batch x 12 floats;
 * exclude_global_dev = batch int64 global row ids or NULL): always the batched
 * matrix-core path below, nothing is staged through the host, the call is
 * hipGraph-capturable after a first (allocating) call.  topn <= 128. */
int mi355rec_enqueue_batch_keys_dev(mi355rec_t* h, const float* queries_dev,
                                    const int64_t* exclude_global_dev, int batch,
                                    int topn, mi355rec_key_t* out_keys_dev, void* stream);

/* How batches (topn <= 128) are served.  AUTO: up to 12 queries as one
 * multi-query pass; 13 and more on shards of >= 65536 rows through the batched
 * path: two passes over the shard per chunk of up to 1024 queries, in which a
 * conservative fp16 pre-filter on the matrix cores (v_mfma_f32_32x32x16_f16 on
 * L2-normalised rows x queries, error bound 1.0e-3 derived in
 * csrc/batched.hip.h) selects a few hundred candidate rows per query that are
 * then scored with the exact fp32 chain — results stay bit-identical to the
 * single-query path.  Queries the bound cannot be claimed for (tiny / huge /
 * non-finite norms, fewer than topn+1 clearly positive groups, more candidates
 * than a query's list holds: about rows / 64, at most 65536) are served by the
 * exact multi-query scan inside the same call.
 * MULTI / MFMA force one path (tests, A/B measurements).  The first batched
 * call allocates the path's scratch (~0.3 GB + the candidate lists, 8 ... 256
 * MB); later calls allocate nothing. */
#define MI355REC_BATCH_AUTO 0
#define MI355REC_BATCH_MULTI 1
#define MI355REC_BATCH_MFMA 2
#define MI355REC_BATCH_HALF 3   /* multi-query passes (<= 32 queries each) with rows from the fp16 replica, whatever the count */
#define MI355REC_BATCH_Q8 4     /* the same passes with rows from the 8-bit replica (integer matrix core, candidates re-checked
                                   against their fp16 rows): half the bytes, but 3.7 us per query of a pass instead of 0.85 —
                                   AUTO takes it for passes of one or two queries only */
/* ... and how much of pass 2 the tile maxima of pass 1 saved in that chunk: the (64-row tile, 32-query block) pairs whose
 * MFMAs pass 2 ran, out of all of them (equal when the chunk ran without tile maxima: fewer than 512 queries, no replica,
 * MI355REC_BATCH_MFMA_NOSKIP).  Synchronises the device. */
int mi355rec_batched_pass2_pairs(mi355rec_t* h, int64_t* pairs_done, int64_t* pairs_total);

#define MI355REC_BATCH_MFMA_NOSKIP 5   /* MFMA, but pass 2 looks at every (tile, query block) pair instead of skipping those the
                                   maxima pass 1 left behind rule out (csrc/batched.hip.h, kTileMax): A/B measurements, tests */
int mi355rec_set_batch_path(mi355rec_t* h, int path);

/* Diagnostics of the LAST chunk (<= 1024 queries) the batched path served on
 * this handle (synchronises the device): special rows listed, queries handed to
 * the exact multi-query scan, and the candidates the pre-filter let through
 * (total and per-query maximum, over the queries it served itself).  Any
 * pointer may be NULL. */
int mi355rec_batched_last_counters(mi355rec_t* h, int32_t* special_rows,
                                   int32_t* queued_queries, int64_t* candidates_total,
                                   int32_t* candidates_max);

/* Merge `n_lists` lists of `list_len` packed keys each (each sorted
 * descending, 0-padded — e.g. the all-gathered per-rank outputs of
 * mi355rec_enqueue_*_keys) into the global best `topn` keys (sorted
 * descending, 0-padded), and optionally unpack them.  out_idx_dev /
 * out_score_dev may be NULL.  Unused idx slots are -1.
 * topn <= MI355REC_MAX_TOPN_FAST here. */
int mi355rec_enqueue_merge_keys(mi355rec_t* h, const mi355rec_key_t* lists_dev,
                                int n_lists, int list_len, int topn,
                                mi355rec_key_t* out_keys_dev,
                                int64_t* out_idx_dev, float* out_score_dev,
                                void* stream);

/* Batched merge: query b's list l starts at lists_dev + b*query_stride +
 * l*list_stride (in keys).  [rank][query][key] data straight out of an
 * all-gather of per-rank batch results is list_stride = batch*topn,
 * query_stride = topn.  Outputs are batch x topn. */
int mi355rec_enqueue_merge_keys_batch(mi355rec_t* h, const mi355rec_key_t* lists_dev,
                                      int n_lists, int list_len, int64_t list_stride,
                                      int64_t query_stride, int batch, int topn,
                                      mi355rec_key_t* out_keys_dev, int64_t* out_idx_dev,
                                      float* out_score_dev, void* stream);

/* Full score vector into device memory (local_row >= 0: query = that row and
 * query12 is ignored; local_row < 0: query12 is used). */
int mi355rec_enqueue_scores(mi355rec_t* h, int64_t local_row,
                            const float* query12, float* out_scores_dev,
                            void* stream);

/* Plain read-only streaming kernel over the same matrix (the achievable-HBM
 * ceiling probe of SURVEY.md §8(d)); writes one checksum word per workgroup
 * to sink_dev (>= compute_units uint32). */
int mi355rec_enqueue_stream_probe(mi355rec_t* h, uint32_t* sink_dev, void* stream);
/* The same plain read over ANOTHER buffer of the handle, so that a kernel's rate can be held against the read ceiling of
 * the buffer it actually streams, in the memory that buffer actually lives in (a 120 MB 8-bit replica sits in the 256 MiB
 * Infinity Cache; the 480 MB fp32 matrix does not): MI355REC_PROBE_FP32_ROWS (what mi355rec_enqueue_stream_probe
 * reads), _FP16_REPLICA (24 B/row: the multi-query and batched passes), _Q8_REPLICA (12 B/row: single queries).
 * MI355REC_ERR_INVALID_ARG when the handle has no such buffer. */
#define MI355REC_PROBE_FP32_ROWS 0
#define MI355REC_PROBE_FP16_REPLICA 1
#define MI355REC_PROBE_Q8_REPLICA 2
int mi355rec_enqueue_stream_probe_of(mi355rec_t* h, int which, uint32_t* sink_dev, void* stream);

/* Brackets the following scan / merge launches with HIP events on their stream
 * so that mi355rec_stats reports last_scan_ms / last_merge_ms (averages over
 * the recorded launches; reading them synchronises the events).  enabled = 0
 * disables, 1 times every launch, k > 1 times every k-th launch of each kind
 * (an event pair costs a few microseconds of stream time). */
int mi355rec_set_timing(mi355rec_t* h, int enabled);

/* The 12 features of one resident row, copied back to the host (48 bytes). */
int mi355rec_fetch_row(mi355rec_t* h, int64_t local_row, float* out12_host);

/* ---- row-sharded catalogue: ONE host process, every GPU of the node ---------
 * Replaces the reference's single-device residency (cudaSetDevice(0),
 * Recommender.cu:124) for a C / C++ host: shard r holds a contiguous block of
 * rows on its own device (balanced split, global row ids), every query runs the
 * fused scan + local merge on all shards concurrently, the per-shard top-N key
 * lists meet on the first device and the same merge kernel produces the global
 * top-N.  The result is independent of the number of shards.
 *
 * The only inter-GPU exchange is topn x batch uint64 keys per shard:
 *   MI355REC_TRANSPORT_PEER  the shards' merge kernels store their keys straight
 *       into the first device's gather buffer through a peer mapping (xGMI
 *       point-to-point) and an event per shard orders the final merge behind
 *       them — no collective, no copy launch.  Default when peer access exists.
 *   MI355REC_TRANSPORT_RCCL  one ncclAllGather per rank per exchange, issued by that rank's worker
 *       thread on its own stream (communicators from ncclCommInitAll; one thread per communicator,
 *       so no ncclGroupStart/End); librccl is opened on first use of this transport.
 * The mi355rec_sharded_query_* calls are synchronous (results in host memory on
 * return); the STREAM calls further down only enqueue.  One host thread at a time
 * per handle.  topn > 1024 is served exactly but slowly (per-shard rounds, the G
 * sorted key lists merged on the host).  With ONE shard the synchronous calls go
 * straight to that shard's single-device handle (no exchange, no second merge). */
typedef struct mi355rec_sharded mi355rec_sharded_t;

#define MI355REC_MAX_SHARDS 64
#define MI355REC_TRANSPORT_PEER 1
#define MI355REC_TRANSPORT_RCCL 2

/* n_devices = 0: the library decides (mi355rec_auto_shards: as many devices as keep at least 4 M rows per
 * shard — a smaller shard is launch-bound and every shard adds to the exchange; 1 device up to 7.9 M rows, 2 at
 * 10 M, all 8 of a node from 32 M rows on); otherwise devices 0 .. n_devices-1.  `feats_host` is the whole
 * row-major n x 12 matrix; each device receives its own block only.
 * (= mi355rec_create_placed(feats, n, dim, NULL, n_devices, MI355REC_PLACEMENT_SHARDED, out).) */
int mi355rec_create_sharded(const float* feats_host, int64_t n, int dim, int n_devices,
                            mi355rec_sharded_t** out);

/* PLACEMENT of the catalogue on the node's devices (replaces cudaSetDevice(0), Recommender.cu:124).
 *   MI355REC_PLACEMENT_SHARDED     rows split into contiguous blocks, one per device (north_star; above).  The only
 *       placement for a catalogue that does not fit one device, and the one that lowers the latency of a query once
 *       a scan is longer than its launches (100 M rows: 207 us on one device).
 *   MI355REC_PLACEMENT_REPLICATED  every device holds ALL rows (and both replicas: 84 B per row, 840 MB at 10 M rows)
 *       and serves whole WINDOWS of the stream by itself, the windows dealt round-robin over the devices: no exchange,
 *       no merge across devices, tickets and mi355rec_sharded_wait unchanged; synchronous calls go to the replicas in
 *       turn.  Queries per second scale with the devices for any catalogue that fits one — which a 10 M-row
 *       catalogue split eight ways does not (1.25 M-row shards are launch-bound at ~12 us per query and every query
 *       pays the exchange).
 *   MI355REC_PLACEMENT_AUTO        = SHARDED (over the device count chosen by the size of the catalogue when
 *       n_devices = 0).
 * devices: an explicit list of n_devices device ordinals (a device may repeat: virtual shards, how the sharded placement
 * is exercised on a one-GPU box; a REPLICATED placement that lists a device more than once gets LANES there —
 * mi355rec_create_lane: one copy of the rows and replicas, several chains of launches that overlap; {0, 0} on one GPU
 * serves a stream at ~57 k queries/s where one handle does 41 k), or NULL: devices 0 .. n_devices-1, n_devices = 0 letting the library
 * choose (SHARDED / AUTO: mi355rec_auto_shards; REPLICATED: every visible device).
 * For a replicated handle mi355rec_sharded_info reports one "shard" of n rows per replica, and the transport is
 * meaningless (mi355rec_sharded_set_transport accepts and ignores it). */
#define MI355REC_PLACEMENT_AUTO 0
#define MI355REC_PLACEMENT_SHARDED 1
#define MI355REC_PLACEMENT_REPLICATED 2
#define MI355REC_PLACEMENT_CPU 3   /* reported, never asked for: see "no HIP device" below */
int mi355rec_create_placed(const float* feats_host, int64_t n, int dim, const int* devices, int n_devices,
                           int placement, mi355rec_sharded_t** out);
int mi355rec_auto_shards(int64_t n, int visible_devices);       /* the size-aware default; 0 without a device */
int mi355rec_sharded_placement(const mi355rec_sharded_t* h);    /* MI355REC_PLACEMENT_SHARDED, _REPLICATED or _CPU */

/* NO HIP DEVICE.  The reference degrades to its CPU loop when it finds no GPU (Recommender.cu:117-127,176-181:
 * "No CUDA devices found. Falling back to CPU", then calculateSimilaritiesCPU, :256-273).  So do
 * mi355rec_create_sharded / mi355rec_create_placed when NO device is visible and the caller left the choice of
 * devices to the library (devices == NULL, n_devices == 0): the handle is then served by the product's own CPU
 * backend (csrc/cpu_backend.cpp: the same sequential fp32 chain over the row-major matrix, rows split over OpenMP
 * threads, a top-N per thread and one merge; canonical order) — every call of this section works, the stream
 * computes a query when it is enqueued, mi355rec_sharded_info reports 0 shards, mi355rec_sharded_placement
 * MI355REC_PLACEMENT_CPU and mi355rec_sharded_note says so.  This is BASELINE configs[0] ("CPU cosine path only").
 * It is never taken on a host with a device, an explicit device list still fails with MI355REC_ERR_NO_DEVICE, and
 * the single-device calls (mi355rec_create*) have no CPU path at all. */

/* Explicit placement: shard r on device devices[r].  A device may appear more
 * than once (virtual shards: several shards of one GPU; how the orchestration is
 * exercised on a one-GPU box) — the RCCL transport then refuses, the peer
 * transport degenerates to stores into local memory. */
int mi355rec_create_sharded_on(const float* feats_host, int64_t n, int dim, const int* devices,
                               int n_shards, mi355rec_sharded_t** out);

void mi355rec_sharded_destroy(mi355rec_sharded_t* h);
const char* mi355rec_sharded_last_error(const mi355rec_sharded_t* h);
int mi355rec_sharded_set_transport(mi355rec_sharded_t* h, int transport);

/* Any out pointer may be NULL; devices_out / shard_rows_out need n_shards slots. */
int mi355rec_sharded_info(const mi355rec_sharded_t* h, int* n_shards, int* transport,
                          int64_t* rows, int* devices_out, int64_t* shard_rows_out);

/* Per-shard diagnostics: mi355rec_set_timing on every shard's engine, and the
 * mi355rec_stats of one shard (kernel event times, grid geometry, replica state). */
int mi355rec_sharded_set_timing(mi355rec_sharded_t* h, int enabled);
int mi355rec_sharded_shard_stats(const mi355rec_sharded_t* h, int shard, mi355rec_stats_t* out);
/* mi355rec_set_replica on every shard (AUTO / OFF / ON); flushes an open stream window first. */
int mi355rec_sharded_set_replica(mi355rec_sharded_t* h, int mode);

/* 1 when queries by row are read by every shard straight from the owning shard's
 * memory (all-pairs peer access, verified against the by-value path when the handle
 * was created), 0 when the row is fetched to the host once per query.  The note says
 * why a fast path was switched off ("" when none was). */
int mi355rec_sharded_rows_by_pointer(const mi355rec_sharded_t* h);
const char* mi355rec_sharded_note(const mi355rec_sharded_t* h);

/* Same contracts as mi355rec_query_row_topn / _query_topn / _query_batch_topn /
 * _scores_row, with GLOBAL row indices. */
int mi355rec_sharded_query_row_topn(mi355rec_sharded_t* h, int64_t global_row, int topn,
                                    int64_t* out_idx, float* out_score, int* out_count);
int mi355rec_sharded_query_topn(mi355rec_sharded_t* h, const float* query12, int64_t exclude_global,
                                int topn, int64_t* out_idx, float* out_score, int* out_count);
int mi355rec_sharded_query_batch_topn(mi355rec_sharded_t* h, const float* queries, int batch,
                                      const int64_t* exclude_global, int topn, int64_t* out_idx,
                                      float* out_score, int* out_count);
int mi355rec_sharded_scores_row(mi355rec_sharded_t* h, int64_t global_row, float* out_host);

/* ---- a STREAM of single queries on the row-sharded catalogue (asynchronous) ------
 * What a serving loop uses (and `bench.py --gpus N`): every call only ENQUEUES —
 * one streamed scan launch per shard (mi355rec_enqueue_*_keys_streamed: the local
 * merge of query k rides in the scan launch of query k + 1) — and returns a TICKET.
 * The per-shard key lists of `window` consecutive queries cross xGMI in ONE exchange
 * (peer stores straight into the first device's gather buffer + one event per shard,
 * or one ncclAllGather per rank) followed by ONE batched merge launch on the first
 * device, whose results (global row ids + scores) the merge kernel stores straight
 * into pinned host memory.  No call but mi355rec_sharded_wait synchronises, nothing
 * is allocated after the first call with a given (topn, window).
 *
 *   enqueue_row / enqueue_query   -> *ticket (consecutive within a window; a flush
 *                                    rounds the next ticket up to a window boundary)
 *   enqueue_flush                 -> closes the open window now (drains the per-shard
 *                                    pipelines: call it when the burst is over)
 *   wait(ticket, ...)             -> blocks until that ticket's window has been merged
 *                                    and copies its result out (flushes first if the
 *                                    window is still open)
 * A window's results stay readable until 3 further windows have been opened
 * (a ring of 4); waiting for an older ticket is MI355REC_ERR_INVALID_ARG.
 * A query by ROW is read by every shard from the owning shard's memory through the
 * peer mapping (mi355rec_enqueue_ptr_keys_streamed) — no 48-byte round trip through
 * the host; where some pair of devices has no peer access the row is fetched once
 * (mi355rec_fetch_row) and passed by value.  topn <= 1024, window in [1, 64]. */
int mi355rec_sharded_set_window(mi355rec_sharded_t* h, int window);   /* default 16; flushes an open window */
/* How a window runs on the shards.  batched = 1 (default): where every shard can take a batch in
 * multi-query passes over its fp16 replica (shards of >= 65536 rows, topn <= 128, window >= 2:
 * mi355rec_batch_pointers_ok) the queries of a window are collected on the host and reach every shard
 * in ONE mi355rec_enqueue_batch_mixed_keys call when the window closes — three launches per shard per
 * WINDOW and one pass over the shard per 32 queries, at the price that a query only starts when its
 * window closes (or at the flush).  batched = 0, or shards that cannot: one streamed scan launch per
 * shard per QUERY, as described above. */
int mi355rec_sharded_set_window_mode(mi355rec_sharded_t* h, int batched);
int mi355rec_sharded_enqueue_row(mi355rec_sharded_t* h, int64_t global_row, int topn, int64_t* ticket);
int mi355rec_sharded_enqueue_query(mi355rec_sharded_t* h, const float* query12, int64_t exclude_global,
                                   int topn, int64_t* ticket);
int mi355rec_sharded_enqueue_flush(mi355rec_sharded_t* h);
int mi355rec_sharded_wait(mi355rec_sharded_t* h, int64_t ticket, int64_t* out_idx, float* out_score,
                          int* out_count);
/* Host-side cost accounting of the stream (cumulative since create): queries
 * enqueued, exchanges issued, and the wall-clock nanoseconds the enqueue / flush
 * calls themselves took on the host thread.  Any pointer may be NULL. */
int mi355rec_sharded_stream_stats(const mi355rec_sharded_t* h, int64_t* queries, int64_t* exchanges,
                                  int64_t* host_ns);

/* TEST HOOK for the cross-workgroup hand-offs of the streamed scans (csrc/replica.hip.h, "hand-offs that fail
 * safe": sample values and cutoffs carry the epoch of their query, arrival counters are never reset).  Simulates
 * what a reader would see if the stores it depends on had not landed; results must stay those of the oracle —
 * only slower.  flags (or-ed):
 *   POISON          now (synchronises the device): every sample buffer and every left-behind cutoff of the handle is
 *                   overwritten with the most hostile values an EARLIER query could have left (a perfect score, a
 *                   cutoff of +1.0) under the epochs of the last queries;
 *   DROP_STORES     the next sampling launch (the seed riders of a streamed launch, or the sample launch of a batch on
 *                   its own / at the head of a stream) does not store the first half of its regions (whoever selects
 *                   the cutoffs then reads whatever was there before);
 *   NO_LAST_RIDER   the next sampling launch is told a wrong arrival count, so none of its workgroups selects a cutoff
 *                   (the pass behind it then finds whatever cutoff was there before).
 * Never needed in production; tests/test_gpu_replica.py and tests/test_gpu_half_multi.py use it. */
#define MI355REC_DEBUG_HANDOFF_POISON 1
#define MI355REC_DEBUG_HANDOFF_DROP_STORES 2
#define MI355REC_DEBUG_HANDOFF_NO_LAST_RIDER 4
int mi355rec_debug_handoff(mi355rec_t* h, int flags);

/* ---- key helpers (host side, no device needed) --------------------------- */
mi355rec_key_t mi355rec_pack_key(float score, int64_t global_row);
float mi355rec_key_score(mi355rec_key_t key);
int64_t mi355rec_key_row(mi355rec_key_t key); /* -1 for the empty key */

#ifdef __cplusplus
}
#endif
#endif /* MI355REC_H */
