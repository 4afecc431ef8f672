// node_stream.hip.h — the ticketed STREAM of single queries over a node (mi355rec_sharded_enqueue_* / _flush / _wait): one
// streamed scan launch per shard per query — or, with batched windows, one multi-query pass per shard per WINDOW — one
// exchange and one batched merge per window, results in a ring of four windows in pinned host memory.  (Part of
// sharded.hip's translation unit.)
#pragma once

#include "node_state.hip.h"

namespace {

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
