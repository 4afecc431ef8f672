import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def experiment_build() -> bool:
    """Is the library the tests will load an MI355REC_EXPERIMENTS build (tools/: the A/B routes exist)?  The product
    library is not; the parametrised GPU tests then leave those routes' cells out instead of skipping them one by one."""
    from spotify_recommender_amd import capi
    try:
        return capi.LIB_PATH.exists() and capi.has_experiments()
    except Exception:
        return False


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture(scope="session")
def engine_lib():
    """The built HIP library; building is part of __graft_entry__.build()."""
    from spotify_recommender_amd import build, capi
    if not capi.LIB_PATH.exists():
        build.build_engine()
    return capi.lib()


@pytest.fixture(autouse=True)
def _gpu_tests_run_on_the_gpu(request):
    """A test marked `gpu` must exercise the HIP path: the library has to see a device, so that the node-level entry
    points cannot hand such a test the CPU backend of device-less hosts (csrc/cpu_backend.cpp) without anyone noticing."""
    if request.node.get_closest_marker("gpu") is not None:
        import torch
        assert torch.cuda.is_available(), "a test marked gpu is running without a GPU"
        from spotify_recommender_amd import capi
        assert capi.lib().mi355rec_device_count() > 0, "libmi355rec.so sees no HIP device on a GPU box"
    yield
