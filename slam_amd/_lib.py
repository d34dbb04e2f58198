"""Loader of libslamfe.so.  There is no CPU fallback: if the library is missing
or does not export the ABI of include/slamfe.h, importing fails loudly."""
from __future__ import annotations

import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLAMFE_LIB names another build of the same library (A/B timing of kernels)
LIB_PATH = os.environ.get("SLAMFE_LIB") or os.path.join(_HERE, "libslamfe.so")

u8p = C.POINTER(C.c_uint8)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
vp = C.c_void_p
ctxp = C.c_void_p

# name -> (restype, argtypes); the complete list of include/slamfe.h
SIGNATURES = {
    "slamfe_version": (C.c_int, []),
    "slamfe_default_params": (C.c_int, [C.POINTER(abi.Params)]),
    "slamfe_create": (C.c_int, [C.POINTER(abi.Params), C.POINTER(ctxp)]),
    "slamfe_destroy": (None, [ctxp]),
    "slamfe_last_error": (C.c_char_p, [ctxp]),
    "slamfe_extract": (C.c_int, [ctxp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int,
                                 i32p, u64p]),
    "slamfe_extract_view": (C.c_int, [ctxp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp),
                                      C.POINTER(vp), i32p, u64p]),
    "slamfe_extract_scaled_bgr": (C.c_int, [ctxp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                            vp, i32p, i32p, vp, vp, C.c_int, i32p, u64p]),
    "slamfe_extract_batch": (C.c_int, [ctxp, C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, vp, vp, C.c_int, i32p, u64p]),
    "slamfe_extract_resident": (C.c_int, [ctxp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                          C.c_int, u64p]),
    "slamfe_frame_download": (C.c_int, [ctxp, C.c_uint64, vp, vp, C.c_int, i32p]),
    "slamfe_match_knn2": (C.c_int, [ctxp, vp, C.c_int, vp, C.c_int, vp]),
    "slamfe_match_frames": (C.c_int, [ctxp, C.c_uint64, C.c_uint64, vp, C.c_int, i32p]),
    "slamfe_match_frames_view": (C.c_int, [ctxp, C.c_uint64, C.c_uint64, C.POINTER(vp), i32p]),
    "slamfe_match_resident": (C.c_int, [ctxp, C.c_uint64, C.c_uint64, vp]),
    "slamfe_match_resident_batch": (C.c_int, [ctxp, C.c_uint64, C.c_uint64, C.c_int, vp]),
    "slamfe_debug_time_stage": (C.c_int, [ctxp, C.c_int, C.c_int, C.c_int, f32p]),
    "slamfe_debug_probe": (C.c_int, [ctxp, C.c_int]),
    "slamfe_debug_probe_read": (C.c_int, [ctxp, f32p, C.POINTER(C.c_int)]),
    "slamfe_match_shard_keys": (C.c_int, [ctxp, vp, C.c_int, vp, C.c_int, C.c_int64, vp]),
    "slamfe_match_shard_keys_static": (C.c_int, [ctxp, vp, C.c_int, vp, C.c_int, C.c_int64, vp]),
    "slamfe_merge_shard_keys": (C.c_int, [ctxp, vp, C.c_int, C.c_int, vp]),
    "slamfe_map_put": (C.c_int, [ctxp, vp, vp, C.c_int]),
    "slamfe_map_put_xyz": (C.c_int, [ctxp, vp, vp, C.c_int]),
    "slamfe_reproject": (C.c_int, [ctxp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_int, vp, vp,
                                   C.c_int, i32p]),
    "slamfe_triangulate": (C.c_int, [ctxp, vp, vp, vp, vp, C.c_int, vp]),
    "slamfe_search_by_projection_visible": (C.c_int, [ctxp, C.c_uint64, vp, C.c_int, C.c_float, C.c_int,
                                                      vp]),
    "slamfe_search_by_projection": (C.c_int, [ctxp, C.c_uint64, vp, C.c_int, vp, vp, C.c_int,
                                              C.c_float, C.c_int, vp]),
    "slamfe_sync": (C.c_int, [ctxp]),
    "slamfe_debug_read_level": (C.c_int, [ctxp, C.c_int, C.c_int, C.c_int, vp, C.c_int, i32p, i32p]),
    "slamfe_frame_device_ptrs": (C.c_int, [ctxp, C.c_uint64, C.POINTER(vp), C.POINTER(vp),
                                           C.POINTER(vp)]),
}

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C slam_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    # PyTorch bundles its own libamdhip64.so.7.  Two HIP runtimes in one process
    # do not work (the second one finds no GPU), so when torch is installed it is
    # imported first: libslamfe.so's dependency on the soname libamdhip64.so.7
    # then binds to the runtime torch has already loaded, and device pointers
    # and streams can be shared with torch tensors.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
