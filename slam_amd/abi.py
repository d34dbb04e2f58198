"""ctypes / numpy views of the PODs declared in include/slamfe.h."""
from __future__ import annotations

import ctypes as C

import numpy as np

SLAMFE_OK = 0
SLAMFE_E_INVALID = -1
SLAMFE_E_CAPACITY = -2
SLAMFE_E_DEVICE = -3
SLAMFE_E_NOMEM = -4
SLAMFE_E_HANDLE = -5

MODE_ORB = 0
MODE_PROVIDED = 1
MATCH_AUTO, MATCH_POPCOUNT, MATCH_MATRIX = 0, 1, 2

# cv::KeyPoint layout (28 bytes)
KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

MATCH2_DTYPE = np.dtype(
    [("idx0", "<i4"), ("idx1", "<i4"), ("d0", "<u2"), ("d1", "<u2"), ("ratio_pass", "u1"),
     ("nfound", "u1"), ("reserved", "u1", (2,))])
assert MATCH2_DTYPE.itemsize == 16


class Params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("mode", C.c_int32),
        ("fast_threshold", C.c_int32),
        ("nfeatures", C.c_int32),
        ("scale_factor", C.c_float),
        ("nlevels", C.c_int32),
        ("edge_threshold", C.c_int32),
        ("patch_size", C.c_int32),
        ("ratio", C.c_float),
        ("device_id", C.c_int32),
        ("max_width", C.c_int32),
        ("max_height", C.c_int32),
        ("max_batch", C.c_int32),
        ("max_keypoints", C.c_int32),
        ("frame_slots", C.c_int32),
        ("map_capacity", C.c_int32),
        ("stream", C.c_void_p),
        ("match_kernel", C.c_int32),
        ("reserved0", C.c_int32),
    ]


def reference_params(**overrides) -> Params:
    """The constants of the reference (Init.cc:64-66, main.cc:11, Init.cc:146)
    without needing the library; slamfe_default_params returns the same."""
    p = Params()
    p.struct_size = C.sizeof(Params)
    p.mode = MODE_ORB
    p.fast_threshold = 30
    p.nfeatures = 10000
    p.scale_factor = 1.2
    p.nlevels = 8
    p.edge_threshold = 31
    p.patch_size = 31
    p.ratio = 0.75
    p.device_id = 0
    p.max_width = 1920
    p.max_height = 1080
    p.max_batch = 1
    p.max_keypoints = 0
    p.frame_slots = 0
    p.map_capacity = 1 << 16
    p.stream = None
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p
