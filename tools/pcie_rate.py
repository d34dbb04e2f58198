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
