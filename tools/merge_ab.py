#!/usr/bin/env python3
"""merge_kernel on the list shapes scans leave behind, timed with HIP events (200 merges back to back per shape):
  python3 tools/merge_ab.py [--lib another/libmi355rec.so]        (MI355REC_CAPI_LENIENT=1 for a library of an earlier round)
Shapes: `uniform-bound` = 512 lists, 0-4 keys each (a pass under a launch-wide bound), `full` = 768 full lists of 100 (a scan without a
bound), `top10` = 326 lists of 10, `cluster-65` = 65 long lists + 100 strays among 768, `sparse-8` = 8 full lists among 768."""
import argparse, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--clock", action="store_true", help="--lib is a -DMI355REC_PHASE_CLOCK build (tools/phase_build.sh): the phases of ONE merge per shape")
ap.add_argument("--shape", default=None, help="only this shape (under rocprofv3 --kernel-trace --stats: the kernel's own duration, which back-to-back event timing hides below ~7.6 us)")
a = ap.parse_args()
import numpy as np
import torch
from spotify_recommender_amd import CosineEngine, capi
if a.lib:
    capi.LIB_PATH = Path(a.lib).resolve()


def lists_of(rng, n_lists, list_len, fill, lo, hi):
    total = int(np.sum(fill))
    scores = rng.integers(lo, hi, size=total, dtype=np.uint64)
    rows = rng.permutation(total).astype(np.uint64)
    keys = (scores << np.uint64(32)) | ((~rows) & np.uint64(0xFFFFFFFF))
    out = np.zeros((n_lists, list_len), dtype=np.uint64)
    at = 0
    for l, f in enumerate(fill):
        out[l, :f] = np.sort(keys[at:at + f])[::-1]
        at += f
    return out


rng = np.random.default_rng(3)
W = (0x80000001, 0xBF800000)
T = (0xBF7FF000, 0xBF800000)
shapes = {
    "uniform-bound": (512, 100, 100, rng.integers(0, 5, 512), W),
    "full": (768, 100, 100, np.full(768, 100), W),
    "top10": (326, 10, 10, rng.integers(0, 11, 326), W),
    "cluster-65": (768, 100, 100, np.where(np.arange(768) % 11 == 3, rng.integers(30, 101, 768), np.where(np.arange(768) % 7 == 1, 1, 0)), T),
    "sparse-8": (768, 100, 100, np.where(np.arange(768) % 96 == 5, 100, 0), T),
}
out = {"lib": str(capi.LIB_PATH)}
f = np.random.default_rng(1).random((4096, 12), dtype=np.float32)
with CosineEngine(f) as eng:
    for name, (n_lists, list_len, topn, fill, bits) in shapes.items():
        if a.shape and name != a.shape:
            continue
        L = lists_of(rng, n_lists, list_len, np.minimum(fill, list_len), *bits)
        d = torch.from_numpy(L.view(np.int64)).cuda()
        o = torch.zeros(topn, dtype=torch.int64, device="cuda")
        for _ in range(20):
            eng.enqueue_merge_keys(d, n_lists, list_len, topn, o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            eng.enqueue_merge_keys(d, n_lists, list_len, topn, o)
        e1.record()
        torch.cuda.synchronize()
        if a.clock:
            import ctypes
            fn = capi.lib().mi355rec_debug_phase_clock
            fn.restype = ctypes.c_int
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
            buf = np.zeros(1024 * 8, dtype=np.uint64)
            ph = []
            for _ in range(5):
                eng.enqueue_merge_keys(d, n_lists, list_len, topn, o)
                torch.cuda.synchronize()
                assert fn(buf.ctypes.data, buf.size) == 0
                c = buf.reshape(1024, 8)[1023].astype(np.int64)
                ph.append([(c[i] - c[0]) / 100.0 for i in range(5)])
            ph = np.median(np.array(ph), axis=0)
            out.setdefault("phases_us (first chunk + threshold, rounds, final cut, ranked)", {})[name] = [round(float(x), 2) for x in ph[1:]]
        flat = L.reshape(-1)
        want = np.sort(flat[flat != 0])[::-1][:topn]
        got = o.cpu().numpy().view(np.uint64)
        out[name] = {"us_per_merge": round(e0.elapsed_time(e1) * 1000.0 / a.reps, 2), "right": bool(np.array_equal(got[:len(want)], want))}
print(json.dumps(out))
