#!/usr/bin/env python3
"""20 single-frame slamfe_extract calls (640x480, ORB mode, 2000 features) and, with
argument `kitti`, 1241x376 in the reference's literal mode: run under
rocprofv3 --kernel-trace to see the launch chain of one call."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402
from bench import HostCall, call_times_ms  # noqa: E402

kitti = len(sys.argv) > 1 and sys.argv[1] == "kitti"
w, h, nf, thr, mode = (1241, 376, 10000, 30, abi.MODE_PROVIDED) if kitti else (640, 480, 2000, 20, abi.MODE_ORB)
img = synth.frame_stream(w, h, 1, seed=3)[0]
with FrontEnd(mode=mode, fast_threshold=thr, nfeatures=nf, max_width=w, max_height=h, map_capacity=0) as fe:
    hc = HostCall(fe)
    med, mean = call_times_ms(lambda: hc.extract(img), 20)
    print(f"{w}x{h} mode {mode}: slamfe_extract median {med:.4f} ms, mean {mean:.4f} ms, {hc.counts[0]} keypoints")
