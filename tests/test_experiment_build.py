"""The MI355REC_EXPERIMENTS build of the engine library (spotify_recommender_amd/libmi355rec_experiments.so): the routes that
left the product in round 5 — the fp16 single-query scan, the 8-bit front end of the multi-query pass — and the
environment knobs of tools/ live behind that flag.  They must keep compiling, without scratch (one documented exception in this build), and export the same C-ABI;
tests/test_gpu_experiments.py runs them on the GPU.  No GPU needed here."""
import ctypes

from spotify_recommender_amd import build, capi


def test_experiments_build_compiles_exports_and_reserves_no_scratch():
    lib = build.build_experiments()
    assert lib.exists()
    kernels = build.kernel_metadata(lib)
    assert build.check_no_scratch(lib, build.EXPERIMENTS_TOLERATE) == len(kernels)   # (what is tolerated, and why: build.py)
    assert all(k["scratch"] == 0 for k in kernels if "scan_half_multi_kernel" not in k["name"])
    names = " ".join(k["name"] for k in kernels)
    for fragment in ("scan_half_kernel", "seed_half_kernel", "scan_half_multi_kernel"):
        assert fragment in names, fragment
    product = " ".join(k["name"] for k in build.kernel_metadata(build.LIB_ENGINE))
    assert "16scan_half_kernel" not in product and "seed_half_kernel" not in product      # (… and they are NOT in the product)
    handle = ctypes.CDLL(str(lib))
    for name in capi.SIGNATURES:
        assert hasattr(handle, name), name
    handle.mi355rec_build_flags.restype = ctypes.c_int
    assert handle.mi355rec_build_flags() & capi.BUILD_EXPERIMENTS
    assert not capi.lib().mi355rec_build_flags() & capi.BUILD_EXPERIMENTS
