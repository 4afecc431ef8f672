"""MI355X-native cosine top-N engine behind the Spotify_recommender interface.

Layout (hot path only — see DESIGN.md):
  csrc/kernels.hip.h   gfx950 kernels (fused cosine scan + top-N, merge)
  csrc/mi355rec.hip    C-ABI declared in include/mi355rec.h
  csrc/*.cpp           C++ drop-in: Recommender / DataManager / main over the C-ABI
  capi.py, engine.py   ctypes + torch plumbing (device memory, streams, RCCL)
"""
from . import capi  # noqa: F401
from .engine import CosineEngine, NodeEngine, ShardedEngine, shard_bounds, unpack_keys  # noqa: F401

FEATURE_COUNT = capi.DIM  # Song.h:12
