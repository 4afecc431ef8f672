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
 * THIS header is the core: create / destroy, the synchronous queries the C++ shim calls, the asynchronous and streamed
 * queries of a serving loop, lanes, and the node handle over every GPU of a host — 30 entry points.  Everything a
 * deployment does not need to serve queries — statistics, route and replica controls for A/B runs, probes, the score
 * vector (the parity hook), pointer / mixed batches and the merge-only entry points the multi-GPU layers are built from,
 * set-up variants of the node handle, test hooks — is declared in mi355rec_diag.h (same library).
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
 * destroyed in any order (the rows and replicas go with the last one); a lane starts in the parent's replica mode and batch
 * path; mi355rec_rebuild_replica is refused while lanes exist AND the group has replicas — a group without them (shards under
 * 65 536 rows) gets them from whichever member is first told mi355rec_set_replica(ON), the others take them over when they are
 * told the same; a lane of a lane is a lane of the same group.
 * mi355rec_create_lane is a set-up call: it synchronises the DEVICE once (hipDeviceSynchronize, as mi355rec_create_device does
 * for a borrowed matrix), allocates the lane's ~20 MB of lists and sample buffers and times a few small kernels to choose
 * the lane's stream — not something to call between queries of a running stream. */
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

/* Replaces Recommender::~Recommender (Recommender.cu:86-98). */
void mi355rec_destroy(mi355rec_t* h);

const char* mi355rec_last_error(const mi355rec_t* h);

/* ---- synchronous host API (what the C++ Recommender shim calls) ---------- */

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

/* The same for queries that are ALREADY in device memory (batch x 12 floats;
 * exclude_global_dev = batch int64 global row ids or NULL): always the batched
 * matrix-core path below, nothing is staged through the host, the call is
 * hipGraph-capturable after a first (allocating) call.  topn <= 128. */
int mi355rec_enqueue_batch_keys_dev(mi355rec_t* h, const float* queries_dev,
                                    const int64_t* exclude_global_dev, int batch,
                                    int topn, mi355rec_key_t* out_keys_dev, void* stream);

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

void mi355rec_sharded_destroy(mi355rec_sharded_t* h);
const char* mi355rec_sharded_last_error(const mi355rec_sharded_t* h);

/* Same contracts as mi355rec_query_row_topn / _query_topn / _query_batch_topn, with GLOBAL row indices. */
int mi355rec_sharded_query_row_topn(mi355rec_sharded_t* h, int64_t global_row, int topn,
                                    int64_t* out_idx, float* out_score, int* out_count);
int mi355rec_sharded_query_topn(mi355rec_sharded_t* h, const float* query12, int64_t exclude_global,
                                int topn, int64_t* out_idx, float* out_score, int* out_count);
int mi355rec_sharded_query_batch_topn(mi355rec_sharded_t* h, const float* queries, int batch,
                                      const int64_t* exclude_global, int topn, int64_t* out_idx,
                                      float* out_score, int* out_count);

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
int mi355rec_sharded_enqueue_row(mi355rec_sharded_t* h, int64_t global_row, int topn, int64_t* ticket);
int mi355rec_sharded_enqueue_query(mi355rec_sharded_t* h, const float* query12, int64_t exclude_global,
                                   int topn, int64_t* ticket);
int mi355rec_sharded_enqueue_flush(mi355rec_sharded_t* h);
int mi355rec_sharded_wait(mi355rec_sharded_t* h, int64_t ticket, int64_t* out_idx, float* out_score,
                          int* out_count);

/* ---- key helpers (host side, no device needed) --------------------------- */
mi355rec_key_t mi355rec_pack_key(float score, int64_t global_row);
float mi355rec_key_score(mi355rec_key_t key);
int64_t mi355rec_key_row(mi355rec_key_t key); /* -1 for the empty key */

#ifdef __cplusplus
}
#endif
#endif /* MI355REC_H */
