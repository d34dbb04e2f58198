import sys, os, time, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from slam_amd import abi, synth
from slam_amd.frontend import FrontEnd
from bench import HostCall, timed
for (w,h,nf,thr,mode) in ((640,480,2000,20,0),(1241,376,3000,20,0),(1241,376,2000,20,0),(1241,376,10000,30,1)):
    a,b = synth.frame_stream(w,h,2,seed=3)
    with FrontEnd(mode=mode, fast_threshold=thr, nfeatures=nf, max_width=w, max_height=h, map_capacity=0) as fe:
        hc = HostCall(fe)
        ha = hc.extract(a); hb = hc.extract(b)
        t_m = timed(lambda: hc.match_frames(ha, hb), 50, fe.sync)
        t_e = timed(lambda: hc.extract(b), 50, fe.sync)
        st = {"prev": hc.extract(a)}
        ts = []
        for i in range(30):
            t0=time.perf_counter(); cur = hc.extract(b); t1=time.perf_counter(); hc.match_frames(st["prev"], cur); t2=time.perf_counter()
            st["prev"]=cur; ts.append((t1-t0, t2-t1))
        ts=np.array(ts[5:])*1e3
        print(w,h,nf,mode,'kp',hc.counts[0],'match alone %.3f extract alone %.3f | in loop: extract %.3f match %.3f'%(t_m*1e3,t_e*1e3,ts[:,0].mean(),ts[:,1].mean()), 'maxkp', fe.max_keypoints)
