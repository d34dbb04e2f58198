#!/usr/bin/env python3
"""Time of one pipeline stage alone, launched back to back on the results of a
256-frame VGA extraction (slamfe_debug_time_stage): python tools/time_stage.py <stage>
with stage 1 pyramid, 2 FAST + blur, 3 selection, 4 orientation, 6 descriptors,
7 matching.  SLAMFE_LIB selects a variant build of the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 256
frames = torch.from_numpy(synth.frame_stream(640, 480, B, seed=1)).cuda()
with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=640, max_height=480,
              max_batch=B, frame_slots=2 * B + 2, map_capacity=0) as fe:
    for _ in range(30):
        fe.extract_resident(frames.data_ptr(), B, 640, 480, 640, 640 * 480, 1)
    fe.sync()
    print(os.environ.get("SLAMFE_LIB", "default").split("/")[-1], "stage", stage, "ms per 256 frames:",
          [round(fe.time_stage(stage, B, 20), 4) for _ in range(3)], flush=True)
