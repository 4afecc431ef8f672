#!/usr/bin/env python3
"""Lone synchronous queries at a given size, for a kernel trace: python3 tools/merge_ab.py --rows 1000000 --topn 10 [--lib X]"""
import argparse, sys, time, json
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--topn", type=int, default=10)
ap.add_argument("--queries", type=int, default=400)
ap.add_argument("--lib", default=None)
a = ap.parse_args()
if a.lib:
    from spotify_recommender_amd import capi
    capi.LIB_PATH = Path(a.lib).resolve()
import torch
from spotify_recommender_amd import CosineEngine
from spotify_recommender_amd.synth import synthetic_catalogue
t = synthetic_catalogue(a.rows, seed=12345)
with CosineEngine(t) as eng:
    call = eng.bound_query_row_topn(a.topn)
    for i in range(50):
        call((i * 7919) % a.rows)
    lat = []
    for i in range(a.queries):
        t0 = time.perf_counter()
        call((i * 7919 + 13) % a.rows)
        lat.append((time.perf_counter() - t0) * 1e6)
    lat.sort()
    print(json.dumps({"rows": a.rows, "topn": a.topn, "lib": a.lib or "product", "p50_us": round(lat[len(lat) // 2], 1), "min_us": round(lat[0], 1)}))
