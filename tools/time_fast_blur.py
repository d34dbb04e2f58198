import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from slam_amd import abi, synth
from slam_amd.frontend import FrontEnd
B=256
frames = torch.from_numpy(synth.frame_stream(640,480,B,seed=1)).cuda()
with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=640, max_height=480, max_batch=B, frame_slots=2*B+2, map_capacity=0) as fe:
    for _ in range(30): fe.extract_resident(frames.data_ptr(), B, 640, 480, 640, 640*480, 1)
    fe.sync()
    print(os.environ.get("SLAMFE_LIB","default").split("/")[-1], "fast_blur ms", [round(fe.time_stage(2, B, 20),4) for _ in range(3)])
