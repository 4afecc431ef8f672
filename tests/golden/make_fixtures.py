"""Regenerates the golden fixtures under tests/golden/.

Run from the repo root in the build container:  python tests/golden/make_fixtures.py

Provenance of each file:
  sample_songs.csv / sample_songs_data.bin / sample_loaded.npz
      written and re-read by the REFERENCE's own DataManager
      (oracle/_ref/libref_dm.so = /root/reference/DataManager.cpp compiled in
      place, OMP_NUM_THREADS=1).  The CSV is the DATASET_INFO.md:43-46 sample
      plus the survey's edge rows (quoted comma, letter key, "Major", a bad
      number, a short row).  Pins the songs_data.bin byte format.
  catalogue4096.npz
      4096 x 12 mt19937(20251017) catalogue with planted duplicates, a zero
      row, an orthogonal pair, a NaN row, a denormal row and a scaled copy;
      score vectors + top-{1,10,100} lists from oracle/liboracle.so, whose
      arithmetic and heap replay are pinned to reference outputs recorded in
      SURVEY.md §8(c) (tests/test_oracle_pins.py).  The reference's
      Recommender.cu itself is not buildable here (needs CUDA header
      stand-ins), so these are oracle outputs, not direct reference outputs.
  seeded_top100.npz
      top-100 rows/scores + CRC32 of the full score vector for 4 query rows of the
      1 M and 10 M mt19937(12345) catalogues (the survey's generator), from the
      pinned oracle; its first three entries for query 0 ARE the reference's
      recorded outputs (survey_pins.json).
  survey_pins.json
      the reference outputs recorded in SURVEY.md §8(c)/§6.2, verbatim.
"""
from __future__ import annotations

import ctypes
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import oracle  # noqa: E402

GOLD = Path(__file__).resolve().parent

SAMPLE_CSV = """\
track_id,track_name,artists,danceability,energy,key,loudness,mode,speechiness,acousticness,instrumentalness,liveness,valence,tempo,track_genre
5SuOikwiRyPMVoIQDJUgSV,Gen Z,Babbu,0.601,0.884,0,-3.803,1,0.182,0.0322,0.0,0.0833,0.368,133.005,dance
4qiyUfNDfud38R8iTJMld2,Lalala,Y2K,0.742,0.69,8,-6.573,0,0.0794,0.101,0.0,0.0971,0.84,130.005,dance
2tHwzyyOLoWSFqYNjeVMzj,Introspection,RAC,0.494,0.595,1,-6.461,0,0.0288,0.0221,0.905,0.166,0.0836,170.018,dance
dupA,"Hello, World",Someone,0.5,0.5,C,-5.0,Major,0.05,0.5,0.0,0.2,0.5,120.0,rock
badnum,Broken,Someone,abc,0.5,1,-5.0,1,0.05,0.5,0.0,0.2,0.5,120.0,rock
short,row
"""


def make_sample_bin() -> None:
    ref = ROOT / "oracle" / "_ref" / "libref_dm.so"
    if not ref.exists():
        print("oracle/_ref/libref_dm.so missing (reference not present) — keeping committed sample fixtures")
        return
    os.environ["OMP_NUM_THREADS"] = "1"
    L = ctypes.CDLL(str(ref))
    L.ref_dm_preprocess.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.ref_dm_load.argtypes = [ctypes.c_char_p]
    L.ref_dm_load.restype = ctypes.c_int64
    L.ref_dm_genre_count.restype = ctypes.c_int64
    L.ref_dm_song_features.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    L.ref_dm_song_string.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_char_p, ctypes.c_int64]
    L.ref_dm_song_string.restype = ctypes.c_int64
    L.ref_dm_genre_name.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_int64]
    L.ref_dm_genre_name.restype = ctypes.c_int64
    csv = GOLD / "sample_songs.csv"
    csv.write_text(SAMPLE_CSV)
    out = GOLD / "sample_songs_data.bin"
    assert L.ref_dm_preprocess(str(csv).encode(), str(out).encode()) == 1
    n = L.ref_dm_load(str(out).encode())
    feats = np.zeros((n, 12), np.float32)
    genre = np.zeros(n, np.int32)
    strings = []
    for i in range(n):
        g = ctypes.c_int(0)
        L.ref_dm_song_features(i, feats[i].ctypes.data, ctypes.byref(g))
        genre[i] = g.value
        row = []
        for which in range(3):
            buf = ctypes.create_string_buffer(4096)
            ln = L.ref_dm_song_string(i, which, buf, 4096)
            row.append(buf.raw[:ln].decode())
        strings.append(row)
    genres = {}
    for gid in range(int(L.ref_dm_genre_count())):
        buf = ctypes.create_string_buffer(4096)
        ln = L.ref_dm_genre_name(gid, buf, 4096)
        genres[gid] = buf.raw[:ln].decode()
    np.savez(GOLD / "sample_loaded.npz", feats=feats, genre=genre,
             strings=np.array(strings), genres=json.dumps(genres),
             sizeof_song=np.int32(L.ref_dm_sizeof_song()))
    print("sample bin:", out.stat().st_size, "bytes,", n, "songs,", genres)


def make_catalogue() -> None:
    n = 4096
    f = oracle.mt19937_uniform(20251017, n)
    f[10] = f[5]; f[11] = f[5]; f[12] = f[5]           # exact duplicates of query row 5
    f[100] = 0.0                                        # zero row -> score exactly 0
    f[7, 1::2] = 0.0                                    # row 7 lives on even features
    f[200, 0::2] = 0.0                                  # row 200 on odd ones: orthogonal to 7
    f[300, 3] = np.nan                                  # NaN feature -> 0 everywhere
    f[400] = np.float32(1e-40) * np.arange(1, 13, dtype=np.float32)  # denormals
    f[500] = f[5] * np.float32(0.5)                     # scaled copy of row 5
    f[600] = f[5]; f[600, 11] = np.nextafter(f[5, 11], np.float32(2))  # 1-ulp neighbour
    queries = np.array([5, 7, 100, 300, 400, 1234, 4095, 0], dtype=np.int32)
    scores = np.stack([oracle.scores(f, f[q]) for q in queries])
    out = {"feats": f, "queries": queries, "scores": scores}
    for k in (1, 10, 100):
        out[f"heap_top{k}"] = np.stack([oracle.topn_heap(scores[i], int(q), k) for i, q in enumerate(queries)])
        out[f"canon_top{k}"] = np.stack([oracle.topn_canonical(scores[i], int(q), k)[0] for i, q in enumerate(queries)])
    np.savez_compressed(GOLD / "catalogue4096.npz", **out)
    print("catalogue4096: scores", scores.shape, "q5 top3", out["heap_top10"][0][:3])


def make_seeded_top100() -> None:
    """SURVEY.md §8(c) fixture (4): for the 1 M and 10 M mt19937(12345) catalogues the
    top-100 (row, score) of a few query rows + a checksum of the whole score
    vector, from the pinned oracle.  A few KB; lets the GPU tests check full-size
    results without trusting anything computed on the GPU box."""
    import zlib
    out = {}
    for rows in (1_000_000, 10_000_000):
        f = oracle.mt19937_uniform(12345, rows)
        qs = np.array([0, 7919, rows // 2, rows - 1], dtype=np.int64)
        out[f"queries_{rows}"] = qs
        for q in qs:
            s = oracle.scores(f, f[q], threads=0)
            idx = oracle.topn_heap(s, int(q), 100)
            out[f"heap_idx_{rows}_{q}"] = idx
            out[f"scores_{rows}_{q}"] = s[idx]
            out[f"crc32_{rows}_{q}"] = np.uint32(zlib.crc32(s.tobytes()))
    np.savez_compressed(GOLD / "seeded_top100.npz", **out)
    print("seeded_top100:", sorted(out)[:4], "...")


def make_pins() -> None:
    pins = {
        "source": "SURVEY.md §8(c) and §6.2 — outputs of the reference CPU path recorded by the survey",
        "generator": "std::mt19937(12345), uniform_real_distribution<float>(0,1), row-major i outer / j inner",
        "1M_q0_top3": [[239849, 0.9900839], [790292, 0.9856242], [536994, 0.9825828]],
        "10M_q0_top3": [[6740478, 0.9951413], [6298018, 0.9907624], [1730600, 0.9903774]],
        "tie_scores": [1, 1, 1, 1, 1, 1, 1, 0.5, 0.7],
        "tie_query": 0,
        "tie_top1": [1],
        "tie_top3": [3, 2, 1],
        "tie_top6": [4, 2, 5, 6, 3, 1],
        "tie_top8": [2, 5, 4, 3, 6, 1, 8, 7],
        "zero_query_top3": [2, 1, 0],
        "sample_valid_songs": 4,
        "sample_genres": 2,
    }
    (GOLD / "survey_pins.json").write_text(json.dumps(pins, indent=1) + "\n")


if __name__ == "__main__":
    make_sample_bin()
    make_catalogue()
    make_seeded_top100()
    make_pins()
