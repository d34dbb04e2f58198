#!/usr/bin/env python3
"""Frames/s of slamfe_extract_batch with frames and results in host memory (PCIe
both ways, pinned staging inside the library): the boundary as the reference's
host code would use it.  Never the headline number of bench.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402

B = 64
frames = list(synth.frame_stream(640, 480, B, seed=1))
with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=640, max_height=480,
              max_batch=B, frame_slots=2 * B, map_capacity=0) as fe:
    for _ in range(3):
        out = fe.extract_batch(frames)
    t = time.perf_counter()
    R = 10
    for _ in range(R):
        out = fe.extract_batch(frames)
    el = (time.perf_counter() - t) / R
    print(f"extract_batch, {B} host frames in, keypoints + descriptors out: {el * 1e3:.3f} ms per batch, "
          f"{B / el:.0f} frames/s ({sum(len(f) for f in out)} keypoints)")

# the same through the C ABI with preallocated outputs (no Python allocations)
import ctypes as C  # noqa: E402
with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=640, max_height=480,
              max_batch=B, frame_slots=2 * B, map_capacity=0) as fe:
    cap = fe.max_keypoints
    kps = np.zeros((B, cap), abi.KEYPOINT_DTYPE)
    desc = np.zeros((B, cap, 32), np.uint8)
    counts = np.zeros(B, np.int32)
    handles = np.zeros(B, np.uint64)
    ptrs = (C.c_void_p * B)(*[f.ctypes.data for f in frames])

    def call():
        rc = fe._lib.slamfe_extract_batch(fe._ctx, ptrs, B, 640, 480, 640, 1, kps.ctypes.data, desc.ctypes.data,
                                          cap, counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                          handles.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0
    for _ in range(3):
        call()
    t = time.perf_counter()
    for _ in range(10):
        call()
    el = (time.perf_counter() - t) / 10
    print(f"C ABI only: {el * 1e3:.3f} ms per batch, {B / el:.0f} frames/s")

    def call_in_only():
        rc = fe._lib.slamfe_extract_batch(fe._ctx, ptrs, B, 640, 480, 640, 1, None, None,
                                          0, counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                          handles.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0
    for _ in range(3):
        call_in_only()
    t = time.perf_counter()
    for _ in range(10):
        call_in_only()
    el = (time.perf_counter() - t) / 10
    print(f"C ABI, results left on the device (counts only): {el * 1e3:.3f} ms per batch, {B / el:.0f} frames/s")
