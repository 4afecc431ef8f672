import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from spotify_recommender_amd.engine import CosineEngine
from spotify_recommender_amd.synth import synthetic_catalogue
from spotify_recommender_amd import capi

n = 10_000_000
t = synthetic_catalogue(n, seed=12345)
rows = [(k * 7919) % n for k in range(64)]
q = t[torch.tensor(rows, device="cuda")].cpu().numpy()
with CosineEngine(t) as eng:
    for topn in (100, 10):
        for nb in (1, 2, 4, 8, 12, 16, 24, 32):
            keys = torch.zeros(nb * topn, dtype=torch.int64, device="cuda")
            eng.set_batch_path(capi.BATCH_HALF)
            ex = np.array(rows[:nb], dtype=np.int64)
            eng.enqueue_batch_keys(q[:nb], ex, topn, keys)
            torch.cuda.synchronize()
            b = eng.replica_counters()
            eng.set_timing(1)
            t0 = time.perf_counter()
            for _ in range(20):
                eng.enqueue_batch_keys(q[:nb], ex, topn, keys)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 20
            st = eng.stats()
            eng.set_timing(0)
            a = eng.replica_counters()
            print(f"topn {topn} nq {nb}: {dt*1e6:.1f} us/call, scan kernel {st.last_scan_ms*1e3:.1f} us, merge {st.last_merge_ms*1e3:.1f} us, "
                  f"rescored/pass {(a['rescored_rows']-b['rescored_rows'])/20:.0f} = {(a['rescored_rows']-b['rescored_rows'])/20/nb:.0f} per query", flush=True)

    # a stream of batches: launches per batch 1 (+ head sample, tail merge)
    for topn in (100, 10):
        for nb in (2, 12, 32):
            ring = [torch.zeros(nb * topn, dtype=torch.int64, device="cuda") for _ in range(4)]
            ex = np.array(rows[:nb], dtype=np.int64)
            for k in range(6):
                eng.enqueue_batch_keys_streamed(q[:nb], ex, topn, ring[k % 4])
            eng.enqueue_flush()
            torch.cuda.synchronize()
            eng.set_timing(1)
            t0 = time.perf_counter()
            for k in range(40):
                eng.enqueue_batch_keys_streamed(q[:nb], ex, topn, ring[k % 4])
            eng.enqueue_flush()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 40
            st = eng.stats()
            eng.set_timing(0)
            print(f"STREAMED topn {topn} nq {nb}: {dt*1e6:.1f} us/call = {nb/dt:.0f} q/s, scan kernel {st.last_scan_ms*1e3:.1f} us", flush=True)
