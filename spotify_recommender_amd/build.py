"""In-tree build of the native pieces (hipcc for gfx950, g++ for the host shim).

Everything lands next to the sources so the built ``.so`` files travel with the
repo snapshot to the GPU box (they are git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
CSRC = PKG / "csrc"
INCLUDE = ROOT / "include"

LIB_ENGINE = PKG / "libmi355rec.so"        # HIP kernels + C-ABI (include/mi355rec.h)
LIB_SHIM = PKG / "librecommender_shim.so"  # C++ Recommender/DataManager drop-in
BIN_CLI = PKG / "recommender"              # drop-in CLI (main.cpp equivalent)

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
OFFLOAD_ARCH = "gfx950"

ENGINE_SOURCES = [CSRC / "mi355rec.hip", CSRC / "sharded.hip"]
ENGINE_DEPS = ENGINE_SOURCES + sorted(CSRC.glob("*.hip.h")) + [INCLUDE / "mi355rec.h"]

# -ffp-contract=off: the parity contract is sequential multiply-then-add with no
# FMA contraction (Recommender.cu:264-269 compiled by the reference Makefile:9).
HIP_FLAGS = [
    f"--offload-arch={OFFLOAD_ARCH}", "-O3", "-std=c++17", "-ffp-contract=off",
    "-fPIC", "-shared", f"-I{INCLUDE}", f"-I{CSRC}", "-ldl",
]


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps if Path(d).exists())


def _run(cmd) -> None:
    proc = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(
            "build command failed:\n  " + " ".join(str(c) for c in cmd) + "\n" + proc.stdout + proc.stderr
        )


def build_engine(force: bool = False) -> Path:
    """Compile the HIP engine for gfx950 (cross-compiles without a GPU)."""
    if force or _stale(LIB_ENGINE, ENGINE_DEPS):
        _run([HIPCC, *HIP_FLAGS, "-o", LIB_ENGINE, *ENGINE_SOURCES])
    return LIB_ENGINE


def shim_sources():
    return [CSRC / "Recommender.cpp", CSRC / "DataManager.cpp"]


def build_shim(force: bool = False) -> Path:
    """Compile the C++ drop-in (Recommender/DataManager over the C-ABI) and CLI."""
    srcs = [s for s in shim_sources() if s.exists()]
    if not srcs:
        return LIB_SHIM
    deps = srcs + [INCLUDE / "Recommender.h", INCLUDE / "Song.h", INCLUDE / "DataManager.h",
                   INCLUDE / "mi355rec.h"]
    build_engine(force)
    common = ["g++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", f"-I{INCLUDE}"]
    link = [f"-L{PKG}", "-lmi355rec", "-Wl,-rpath,$ORIGIN"]
    if force or _stale(LIB_SHIM, deps + [LIB_ENGINE]):
        _run([*common, "-shared", "-o", LIB_SHIM, *srcs, CSRC / "shim_capi.cpp", *link])
    main_cpp = CSRC / "main.cpp"
    if main_cpp.exists() and (force or _stale(BIN_CLI, deps + [main_cpp, LIB_SHIM])):
        _run([*common, "-o", BIN_CLI, main_cpp, *srcs, *link])
    return LIB_SHIM


def build_oracle(force: bool = False) -> Path:
    """Build the CPU oracle (test infrastructure; see oracle/README.md)."""
    odir = ROOT / "oracle"
    target = odir / "liboracle.so"
    if force or _stale(target, [odir / "cosine_oracle.c", odir / "cosine_oracle.h"]):
        _run(["make", "-C", odir, "liboracle.so"])
    if Path("/root/reference/DataManager.cpp").exists():
        ref = odir / "_ref" / "libref_dm.so"
        if force or _stale(ref, [odir / "ref_dm_driver.cpp"]):
            _run(["make", "-C", odir, "ref"])
    return target


def build_all(force: bool = False) -> None:
    build_engine(force)
    build_shim(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force=True)
    print("built:", LIB_ENGINE)
