"""Build recipe for libmatpbr.so (hipcc, gfx950 only, in-tree so the .so travels with the repo snapshot)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libmatpbr.so")
SOURCES = ["matpbr_kernels.hip", "posmlp_kernels.hip", "mesh_host.cpp"]
HEADERS = ["matpbr_device.hpp", "matpbr_shade.hpp", "matpbr_lazy.hpp", "matpbr_pstep.hpp", os.path.join("..", "..", "include", "matpbr.h")]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmatpbr.so can only be built with the ROCm toolchain")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP kernels + C ABI into materialist_amd/libmatpbr.so.  Cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [_hipcc(), *HIPCC_FLAGS, "-o", LIB_PATH, *[os.path.join(CSRC, s) for s in SOURCES]]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
