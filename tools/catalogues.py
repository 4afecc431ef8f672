"""Catalogues for the tools that are not uniform noise (tools/run_replica.py, run_batched.py, run_half_multi.py, soak.py:
--catalogue clustered): the generator lives in the package (spotify_recommender_amd/synth.py) since round 5 — the GPU
tests and bench.py use it too."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from spotify_recommender_amd.synth import clustered_catalogue   # noqa: E402,F401
