"""The A/B routes of the MI355REC_EXPERIMENTS build on the GPU: the replica and multi-query test files parametrise on
capi.has_experiments(), so running them against libmi355rec_experiments.so (MI355REC_LIB) covers the fp16 single-query scan
and the 8-bit front end of the multi-query pass with the same oracle checks as the product's routes.  One child process
(a library is chosen when it is loaded)."""
import os
import subprocess
import sys

import pytest

from spotify_recommender_amd import build

pytestmark = pytest.mark.gpu


def test_experiment_routes_match_the_oracle():
    lib = build.LIB_EXPERIMENTS
    assert lib.exists(), f"{lib} is missing: __graft_entry__.build() makes it"
    env = dict(os.environ, MI355REC_LIB=str(lib))
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_replica.py", "tests/test_gpu_half_multi.py", "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=str(build.PKG.parent), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
