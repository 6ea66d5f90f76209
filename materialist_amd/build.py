"""Build recipe for libmatpbr.so (hipcc, gfx950 only, in-tree so the .so travels with the repo snapshot)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libmatpbr.so")
SOURCES = ["matpbr_kernels.hip", "posmlp_kernels.hip", "posmlp_chain.hip", "mesh_host.cpp"]
HEADERS = [os.path.join("..", "..", "include", "matpbr_experimental.h"), "posmlp_device.hpp", "matpbr_device.hpp", "matpbr_shade.hpp", "matpbr_lazy.hpp", "matpbr_pstep.hpp", os.path.join("..", "..", "include", "matpbr.h"), os.path.join("..", "..", "include", "matpbr_mlp.h")]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmatpbr.so can only be built with the ROCm toolchain")


def sources_digest() -> str:
    """SHA-256 (16 hex digits) of every kernel source and header: what profiles/pmc_traffic.json is tied to (bench.py `roofline.traffic`)."""
    import hashlib

    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        with open(os.path.normpath(os.path.join(CSRC, name)), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


# per-source flags.  posmlp_chain.hip: its epilogue is interleaved with matrix instructions chunk by chunk; the SLP vectoriser would gather
# the chunks' scalar sines into packed instructions at ONE place of the slot (and v_pk_* next to MFMAs costs more than two scalar ones)
EXTRA_FLAGS = {"posmlp_chain.hip": ["-fno-slp-vectorize"]}
OBJ_DIR = os.path.join(_HERE, "_build")


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP kernels + C ABI into materialist_amd/libmatpbr.so: one object per source (rebuilt when the source or a header is
    newer), then one link.  Cross-compiles without a GPU."""
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.normpath(os.path.join(CSRC, h))) for h in HEADERS)
    flags = [f for f in HIPCC_FLAGS if f != "-shared"]
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(sp), hdr_t, os.path.getmtime(__file__)):
            cmd = [_hipcc(), *flags, *EXTRA_FLAGS.get(src, []), "-c", sp, "-o", obj]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for cmd, pr in procs:                                      # the sources compile side by side
        out, err = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out + err)
        if verbose:
            print(err)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB_PATH, *objs]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc (link) failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
