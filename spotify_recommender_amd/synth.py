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


def query_rows(n_rows: int, count: int, stride: int = 7919):
    """Deterministic query rows q_k = (k * 7919) mod N."""
    return [(k * stride) % n_rows for k in range(count)]
