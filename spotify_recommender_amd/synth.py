"""Seeded synthetic catalogues: uniform[0,1) fp32 features.

The reference's features are min-max normalised to [0,1]
(DataManager.cpp:291-299), so uniform[0,1) has the same range.  Data is
generated on the device it will live on; every rank of a sharded run generates
the same full catalogue from the same seed and keeps its own row block.
"""
from __future__ import annotations

import torch

from .capi import DIM


def synthetic_catalogue(n_rows: int, seed: int = 12345, device="cuda") -> torch.Tensor:
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    return torch.rand((int(n_rows), DIM), dtype=torch.float32, device=device, generator=gen)


def clustered_catalogue(n: int, spread: float, seed: int = 777, clusters: int = 3000, contiguous: bool = False,
                        ramp: bool = False, device="cuda") -> torch.Tensor:
    """Rows shaped like the reference's min-max normalised features (Song.h:18-19: danceability, energy, key, loudness,
    mode, speechiness, acousticness, instrumentalness, liveness, valence, tempo, genre_id — DataManager.cpp:286-299):
    cluster centres in [0,1]^12 with key in {0..11}/11, mode in {0,1}, genre in {0..113}/113, Gaussian spread around the
    continuous columns, and 2 % exact duplicates (the same track on several albums).
    contiguous: the rows of a cluster lie next to each other (a catalogue sorted by genre / artist / album): a launch-wide
    bound taken from evenly spaced sample regions then mostly MISSES the query's own cluster.
    ramp (with contiguous): features[11] = cluster / (clusters - 1) instead of a random genre per cluster — what the
    reference's preprocessing makes of a CSV that is grouped by genre: genre ids are handed out in order of first
    appearance (DataManager.cpp:244-250) and end up, divided by G - 1, in the twelfth feature (:299), a ramp along the row
    index."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    centres = torch.rand((clusters, DIM), device=device, generator=g)
    centres[:, 2] = torch.randint(0, 12, (clusters,), device=device, generator=g).float() / 11.0
    centres[:, 4] = torch.randint(0, 2, (clusters,), device=device, generator=g).float()
    centres[:, 7] = centres[:, 7] ** 4          # instrumentalness: mostly near 0
    centres[:, 5] = centres[:, 5] ** 3 * 0.5    # speechiness: small
    centres[:, 11] = torch.randint(0, 114, (clusters,), device=device, generator=g).float() / 113.0
    if ramp:
        centres[:, 11] = torch.arange(clusters, device=device, dtype=torch.float32) / float(max(clusters - 1, 1))
    which = torch.randint(0, clusters, (n,), device=device, generator=g)
    if contiguous:
        which = torch.sort(which).values
    t = centres[which]
    noise = torch.randn((n, DIM), device=device, generator=g) * spread
    noise[:, [2, 4, 11]] = 0.0                  # discrete columns stay on their grid
    t = (t + noise).clamp_(0.0, 1.0)
    dup = torch.randint(0, n, (n // 50,), device=device, generator=g)
    src = torch.randint(0, n, (n // 50,), device=device, generator=g)
    t[dup] = t[src]
    return t.contiguous()


def query_rows(n_rows: int, count: int, stride: int = 7919):
    """Deterministic query rows q_k = (k * 7919) mod N."""
    return [(k * stride) % n_rows for k in range(count)]
