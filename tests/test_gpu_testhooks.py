"""The hand-offs that must fail safe, broken on purpose (mi355rec_debug_handoff: POISON / DROP_STORES / NO_LAST_RIDER) — against
libmi355rec_testhooks.so: the product's sources and flags + -DMI355REC_TEST_HOOKS, the same device code (the define only adds a
host function; tests/test_kernel_metadata.py compares the two libraries' kernels).  The product library does not export the
hook, so in the main test process the two tests below are skipped; here they run, with every other test of their files, in
one child process (a library is chosen when it is loaded)."""
import os
import subprocess
import sys

import pytest

from spotify_recommender_amd import build

pytestmark = pytest.mark.gpu


def test_hand_offs_fail_safe_on_the_test_hooks_build():
    lib = build.LIB_TESTHOOKS
    assert lib.exists(), f"{lib} is missing: __graft_entry__.build() makes it"
    env = dict(os.environ, MI355REC_LIB=str(lib))
    r = subprocess.run([sys.executable, "-m", "pytest",
                        "tests/test_gpu_replica.py::test_hand_offs_fail_safe_under_stale_values",
                        "tests/test_gpu_half_multi.py::test_stream_of_batches_fails_safe_under_stale_hand_offs",
                        "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], cwd=str(build.PKG.parent), env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    assert " passed" in last and "skipped" not in last, r.stdout[-500:]   # (both tests RAN: the hook exists in this build)
