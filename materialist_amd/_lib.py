"""ctypes binding of libmatpbr.so (include/matpbr.h).  There is NO fallback: if the HIP library is
missing or a call fails, the product path raises."""
from __future__ import annotations

import ctypes
import os
import threading

from . import build as _build

_c_f = ctypes.c_void_p  # device pointers travel as opaque addresses


class MatpbrCamera(ctypes.Structure):
    _fields_ = [("fov_x_deg", ctypes.c_float)]


class MatpbrError(RuntimeError):
    pass


# name -> (restype, argtypes); must list every symbol include/matpbr.h declares
SIGNATURES = {
    "matpbr_version": (ctypes.c_int, []),
    "matpbr_strerror": (ctypes.c_char_p, [ctypes.c_int]),
    "matpbr_shade_fwd": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.POINTER(MatpbrCamera), ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_shade_bwd": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int] + [_c_f] * 6 + [ctypes.c_void_p, ctypes.c_size_t,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(MatpbrCamera),
                                       ctypes.c_uint32, ctypes.c_void_p]),
    "matpbr_shade_bwd_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "matpbr_brdf_loss_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "matpbr_brdf_loss_stats": (ctypes.c_int, [_c_f] * 9 + [ctypes.c_float, _c_f, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_void_p]),
    "matpbr_shade_bwd_brdf_loss": (ctypes.c_int, [_c_f] * 5 + [ctypes.c_int, ctypes.c_int] + [_c_f] * 6 + [ctypes.c_float] + [_c_f] * 7 +
                                   [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(MatpbrCamera), ctypes.c_uint32,
                                    ctypes.c_void_p]),
    "matpbr_adam_step": (ctypes.c_int, [_c_f] * 4 + [ctypes.c_long, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                       ctypes.c_void_p]),
    "matpbr_eval_brdf": (ctypes.c_int, [_c_f] * 8 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_eval_brdf_bwd": (ctypes.c_int, [_c_f] * 11 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_sample_brdf": (ctypes.c_int, [_c_f] * 10 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_sh_eval": (ctypes.c_int, [_c_f] * 3 + [ctypes.c_long, ctypes.c_void_p]),
    "matpbr_normals_from_depth": (ctypes.c_int, [_c_f, _c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(MatpbrCamera),
                                                 ctypes.c_void_p]),
}

_lock = threading.Lock()
_lib = None


def library_path() -> str:
    return _build.LIB_PATH


def load(build_if_missing: bool = False) -> ctypes.CDLL:
    """Load libmatpbr.so and bind every declared symbol.  Raises MatpbrError when it cannot."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            if build_if_missing:
                _build.build_library()
            else:
                raise MatpbrError(
                    f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). matpbr has no CPU or PyTorch fallback.")
        try:
            lib = ctypes.CDLL(path)
        except OSError as e:  # pragma: no cover - depends on the host
            raise MatpbrError(f"cannot load {path}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise MatpbrError(f"{path} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().matpbr_strerror(code)
        raise MatpbrError(f"{what} failed: {msg.decode() if msg else code} ({code})")
