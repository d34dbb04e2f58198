import sys,time,ctypes as C
import os; sys.path.insert(0, os.getcwd())
import numpy as np, torch
from slam_amd import abi, synth
from slam_amd.frontend import FrontEnd
for (w,h,nf,mode) in ((640,480,2000,abi.MODE_ORB),(1241,376,3000,abi.MODE_ORB),(1241,376,3000,abi.MODE_PROVIDED)):
    img=synth.textured_canvas(w,h,3)
    fe=FrontEnd(mode=mode,fast_threshold=20 if mode==abi.MODE_ORB else 30,nfeatures=nf,max_width=w,max_height=h)
    cap=fe.max_keypoints
    kps=np.zeros(cap,abi.KEYPOINT_DTYPE); desc=np.zeros((cap,32),np.uint8); n=C.c_int32(); hd=C.c_uint64()
    lib=fe._lib
    def call():
        return lib.slamfe_extract(fe.ctx, img.ctypes.data, w,h,w,1, kps.ctypes.data, desc.ctypes.data, cap, C.byref(n), C.byref(hd))
    for _ in range(5): call()
    t=time.perf_counter()
    for _ in range(100): call()
    dt=(time.perf_counter()-t)/100
    print(w,h,mode,'slamfe_extract host call: %.3f ms'%(dt*1e3), 'n=',n.value)
    f=fe.extract_keypoints(img)
    t=time.perf_counter()
    for _ in range(50): fe.extract_keypoints(img)
    print('  python wrapper: %.3f ms'%((time.perf_counter()-t)/50*1e3))
    fe.close()
