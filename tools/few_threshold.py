"""Single-call time of slamfe_extract with the launch-poor kernels on / off, per frame size."""
import sys, os, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np
from slam_amd import abi, synth
from slam_amd.frontend import FrontEnd
from bench import HostCall, call_times_ms
w, h, nf, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
frames = synth.frame_stream(w, h, n, seed=3)
with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=nf, max_width=w, max_height=h, max_batch=n, map_capacity=0) as fe:
    import ctypes as C
    hc = HostCall(fe, n)
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    med, mean = call_times_ms(lambda: hc.extract_batch(ptrs, n, w, h), 30)
    print(f"{n} x {w}x{h} few_pixels={os.environ.get('SLAMFE_FEW_PIXELS','default')}: median {med:.4f} ms")
