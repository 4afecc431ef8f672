/*
 * mi355rec_diag.h — the rest of the C-ABI of libmi355rec.so: what a deployment does not need in order to serve queries.
 *
 * mi355rec.h is the core (create, query, enqueue, stream, lanes, the node handle: 30 entry points).  This header declares,
 * over the same handles and with the same conventions (plain pointers and sizes, 0 / negative codes, caller-owned buffers):
 *   - statistics and what the build is (mi355rec_stats*, _build_flags, _device_count, _last_global_error, _lane_status,
 *     _replica_counters, _batched_*, _sharded_info / _shard_stats / _stream_stats / _note / _rccl_ranks ...);
 *   - controls for A/B measurements and tests (mi355rec_create_ex flags, _set_replica, _rebuild_replica, _set_batch_path,
 *     _set_timing, _sharded_set_transport / _set_window / _set_window_mode / _set_replica);
 *   - the parity hook: the full score vector (mi355rec_scores*, the mirror of the private
 *     Recommender::calculateSimilarities, Recommender.h:114), and the probes (mi355rec_enqueue_stream_probe*);
 *   - the building blocks the multi-GPU layers are made of: queries by device pointer, mixed batches, the merge-only entry
 *     points (mi355rec_enqueue_merge_keys*), mi355rec_fetch_row, the explicit set-ups of the node handle;
 *   - test hooks, compiled in only with -DMI355REC_TEST_HOOKS (spotify_recommender_amd/build.py builds
 *     libmi355rec_testhooks.so for tests/; the product library does not export them).
 */
#ifndef MI355REC_DIAG_H
#define MI355REC_DIAG_H

#include "mi355rec.h"

#ifdef __cplusplus
extern "C" {
#endif

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
#define MI355REC_BUILD_TEST_HOOKS 4    /* mi355rec_debug_handoff is compiled in (libmi355rec_testhooks.so, the experiments build) */
int mi355rec_build_flags(void);

/* Number of visible HIP devices (0 when there is none / no driver). */
int mi355rec_device_count(void);

/* Thread-local text of the last error raised with no handle to attach it to
 * (e.g. a failed create); with a handle, use mi355rec_last_error. */
const char* mi355rec_last_global_error(void);

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
 * single queries scan: signed bytes against a 16-bit query held as two int8 digits, six v_dot4_i32_i8 per
 * row and an integer candidate test; the bound is per query, l1(Q)/(254 S) + sqrt(12)/(2 S) + 3e-5 <= 0.0137, S = 32000).  Both are pre-filters in front of the same
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

/* What mi355rec_create_lane found when it chose the lane's stream: it times a small kernel on the parent's stream alone and
 * on both streams at once, and replaces the lane's stream (each new stream is bound to the next hardware queue) until the pair
 * runs side by side.  *stream_attempts = streams tried (0 on a handle that is not a lane); *overlaps_parent = 1 side by side,
 * 0 none of them did (the lane then buys nothing: use one handle), -1 not testable (under 10 000 rows). */
int mi355rec_lane_status(const mi355rec_t* h, int* stream_attempts, int* overlaps_parent);

int mi355rec_stats(const mi355rec_t* h, mi355rec_stats_t* out);
/* The same for a caller that may have been built against an EARLIER header: mi355rec_stats_t only ever grows at its end, and
 * this copies min(out_size, sizeof(mi355rec_stats_t)) bytes — a shorter struct gets the fields it knows, never an overrun.
 * Pass sizeof(mi355rec_stats_t) of the header you compiled with.  *written (may be NULL) = the bytes copied. */
int mi355rec_stats_sized(const mi355rec_t* h, void* out, size_t out_size, size_t* written);

/* ---- the score vector: the parity hook --------------------------------------- */

/* Replaces the private Recommender::calculateSimilarities(int, float*)
 * (Recommender.h:114, Recommender.cu:184-254): cosine of catalogue row
 * `local_row` against every local row, n floats written to host memory. */
int mi355rec_scores_row(mi355rec_t* h, int64_t local_row, float* out_host);

/* Same for an arbitrary query vector (12 floats, host). */
int mi355rec_scores(mi355rec_t* h, const float* query12, float* out_host);

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

/* ---- the node handle (mi355rec.h: mi355rec_create_placed): set-up variants, controls, statistics ---- */

/* n_devices = 0: the library decides (mi355rec_auto_shards: as many devices as keep at least 4 M rows per
 * shard — a smaller shard is launch-bound and every shard adds to the exchange; 1 device up to 7.9 M rows, 2 at
 * 10 M, all 8 of a node from 32 M rows on); otherwise devices 0 .. n_devices-1.  `feats_host` is the whole
 * row-major n x 12 matrix; each device receives its own block only.
 * (= mi355rec_create_placed(feats, n, dim, NULL, n_devices, MI355REC_PLACEMENT_SHARDED, out).) */
int mi355rec_create_sharded(const float* feats_host, int64_t n, int dim, int n_devices,
                            mi355rec_sharded_t** out);
int mi355rec_auto_shards(int64_t n, int visible_devices);       /* the size-aware default; 0 without a device */
int mi355rec_sharded_placement(const mi355rec_sharded_t* h);    /* MI355REC_PLACEMENT_SHARDED, _REPLICATED or _CPU */

/* Explicit placement: shard r on device devices[r].  A device may appear more
 * than once (virtual shards: several shards of one GPU; how the orchestration is
 * exercised on a one-GPU box) — the RCCL transport then refuses, the peer
 * transport degenerates to stores into local memory. */
int mi355rec_create_sharded_on(const float* feats_host, int64_t n, int dim, const int* devices,
                               int n_shards, mi355rec_sharded_t** out);

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

/* As mi355rec_scores_row, with a GLOBAL row index. */
int mi355rec_sharded_scores_row(mi355rec_sharded_t* h, int64_t global_row, float* out_host);

int mi355rec_sharded_set_window(mi355rec_sharded_t* h, int window);   /* default 16; flushes an open window */
/* How a window runs on the shards.  batched = 1 (default): where every shard can take a batch in
 * multi-query passes over its fp16 replica (shards of >= 65536 rows, topn <= 128, window >= 2:
 * mi355rec_batch_pointers_ok) the queries of a window are collected on the host and reach every shard
 * in ONE mi355rec_enqueue_batch_mixed_keys call when the window closes — three launches per shard per
 * WINDOW and one pass over the shard per 32 queries, at the price that a query only starts when its
 * window closes (or at the flush).  batched = 0, or shards that cannot: one streamed scan launch per
 * shard per QUERY, as described above. */
int mi355rec_sharded_set_window_mode(mi355rec_sharded_t* h, int batched);
/* Host-side cost accounting of the stream (cumulative since create): queries
 * enqueued, exchanges issued, and the wall-clock nanoseconds the enqueue / flush
 * calls themselves took on the host thread.  Any pointer may be NULL. */
int mi355rec_sharded_stream_stats(const mi355rec_sharded_t* h, int64_t* queries, int64_t* exchanges,
                                  int64_t* host_ns);

/* What RCCL itself reports about the communicators of MI355REC_TRANSPORT_RCCL: *comms = communicators the handle holds (one
 * per shard; 0 until that transport has been used), *ranks = ncclCommCount of the first, *ranks_agree = 1 when every
 * communicator reports the same count and its own shard index as its rank (ncclCommUserRank).  bench.py --gpus N puts these
 * in its line, so that a first run on a real node answers "did RCCL see N ranks" by itself.  Any pointer may be NULL. */
int mi355rec_sharded_rccl_ranks(const mi355rec_sharded_t* h, int* comms, int* ranks, int* ranks_agree);

#ifdef MI355REC_TEST_HOOKS
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
 * Never needed in production and NOT in the product library: declared and compiled only with -DMI355REC_TEST_HOOKS
 * (spotify_recommender_amd/build.py: libmi355rec_testhooks.so = the product's sources and flags + that define, the same
 * device code; tests/test_gpu_testhooks.py runs tests/test_gpu_replica.py and tests/test_gpu_half_multi.py against it). */
#define MI355REC_DEBUG_HANDOFF_POISON 1
#define MI355REC_DEBUG_HANDOFF_DROP_STORES 2
#define MI355REC_DEBUG_HANDOFF_NO_LAST_RIDER 4
int mi355rec_debug_handoff(mi355rec_t* h, int flags);
#endif /* MI355REC_TEST_HOOKS */

#ifdef __cplusplus
}
#endif
#endif /* MI355REC_DIAG_H */
