import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
from slam_amd import abi, synth
from slam_amd.frontend import FrontEnd
from bench import HostCall
img = synth.frame_stream(640, 480, 1, seed=3)[0]
with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=640, max_height=480, map_capacity=0) as fe:
    hc = HostCall(fe)
    for _ in range(10): hc.extract(img)
    out = np.zeros(64, np.uint64)
    fe._lib.slamfe_debug_stamps.argtypes = [C.c_void_p, C.c_int]
    rc = fe._lib.slamfe_debug_stamps(out.ctypes.data, 64)
    st = out.reshape(32, 2).astype(np.int64)
    names = ["start", "ranges", "lvl0 loaded", "lvl0 stored"] + [f"L{l} {w}" for l in range(1, 8) for w in ("computed", "stored")]
    for i, nm in enumerate(names):
        if st[i, 0] == 0: continue
        print(f"{nm:14s} cycles +{st[i,0]-st[max(i-1,0),0]:7d}  wall +{(st[i,1]-st[max(i-1,0),1])*10:6d} ns   total {(st[i,1]-st[0,1])*10} ns")
    print("clock GHz ~", (st[len(names)-1,0]-st[0,0]) / ((st[len(names)-1,1]-st[0,1])*10))
