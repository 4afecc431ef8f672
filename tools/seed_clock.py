#!/usr/bin/env python3
"""Where the sample launch of a batch on its own spends its time (seed_half_multi_kernel; a -DMI355REC_PHASE_CLOCK
build, --lib): python3 tools/seed_clock.py --lib gpurun_out/q8/libmi355rec_phase.so --queries 12"""
import argparse, ctypes, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=12)
ap.add_argument("--lib", required=True)
a = ap.parse_args()
import numpy as np
import torch
from spotify_recommender_amd import CosineEngine, capi
from spotify_recommender_amd.synth import synthetic_catalogue
capi.LIB_PATH = Path(a.lib).resolve()
lib = capi.lib()
fn = lib.mi355rec_debug_phase_clock
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
t = synthetic_catalogue(a.rows, seed=12345)
rows = [(k * 7919) % a.rows for k in range(64)]
q = t[torch.tensor(rows[:a.queries], device="cuda")].cpu().numpy()
ex = np.array(rows[:a.queries], dtype=np.int64)
with CosineEngine(t) as eng:
    eng.set_batch_path(capi.BATCH_HALF)
    keys = torch.zeros(a.queries * 100, dtype=torch.int64, device="cuda")
    # the pass and the merge launch that follow the sample launch overwrite rows of the clock too (they stamp as
    # scanners): the sample launch's 256 workgroups are read from a build whose pass does not run — so instead the
    # seed kernel's stamps are told apart by phase 2 > 0 and phase 3 > 0 with nothing in 5..7
    for _ in range(5):
        eng.enqueue_batch_keys(q, ex, 100, keys)
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 8, dtype=np.uint64)
    assert fn(buf.ctypes.data, buf.size) == 0
c = buf.reshape(1024, 8).astype(np.int64)[:256]
t0 = c[:, 0].min()
us = lambda x: round(float(x) / 100.0, 2)
out = {"queries": a.queries}
for i, nm in enumerate(["entry", "fragment_built", "sampled", "arrived", "selected"]):
    v = c[:, i][c[:, i] > 0] - t0
    if len(v):
        out[nm] = {"first": us(v.min()), "median": us(np.median(v)), "last": us(v.max()), "n": int(len(v))}
print(json.dumps(out))
