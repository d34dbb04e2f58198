#!/usr/bin/env python3
"""Call time of slamfe_search_by_projection: BASELINE config 2 (1241x376, 3000
features, 20 000 map points) and the largest visible set a call accepts (2^19
points), host arrays in / ids out.  Prints one JSON object per case, with the
number of gate tests the 3x3-cell window query makes next to the brute-force
pair count."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402
from bench import window_gate_tests as window_tests  # noqa: E402


def case(npts, ninl_cap, reps):
    w, h, nfeat = 1241, 376, 3000
    frames = synth.frame_stream(w, h, 2, seed=9)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=nfeat, max_width=w, max_height=h,
                  max_batch=2, map_capacity=npts) as fe:
        f0, f1 = fe.extract_batch(list(frames))
        m = fe.knn_match_frames(f0, f1)
        uv, pdesc = synth.local_map(f1.descriptors, f1.pt, npts, w, h, seed=11)
        ids = np.arange(npts, dtype=np.int32)
        fe.map_put(ids, pdesc)
        inl = m["idx0"][m["ratio_pass"] == 1].astype(np.int32)[:ninl_cap]
        matched = np.zeros(len(inl), np.int32)

        def search():
            rc = fe._lib.slamfe_search_by_projection(fe._ctx, C.c_uint64(f1.handle), inl.ctypes.data, len(inl),
                                                     uv.ctypes.data, ids.ctypes.data, npts, C.c_float(32.0), 128,
                                                     matched.ctypes.data)
            assert rc == 0, rc
        for _ in range(3):
            search()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            search()
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        tests = window_tests(f1.pt[inl], uv, 32.0, w, h)
        # the same search over the visible set the device-side reprojection leaves
        pose, K, xyz, th = synth.scene_from_uv(uv, w, h)
        fe.map_put_xyz(ids, xyz)
        tr, tv = [], []
        for it in range(reps + 3):
            t0 = time.perf_counter()
            nvis = fe.reproject(pose, K, th, w, h, npts, download=False)
            t1 = time.perf_counter()
            rc = fe._lib.slamfe_search_by_projection_visible(fe._ctx, C.c_uint64(f1.handle), inl.ctypes.data,
                                                             len(inl), C.c_float(32.0), 128, matched.ctypes.data)
            t2 = time.perf_counter()
            assert rc == 0, rc
            if it >= 3:
                tr.append(t1 - t0)
                tv.append(t2 - t1)
        return {"visible_points": npts, "inliers": int(len(inl)), "matched": int((matched >= 0).sum()),
                "call_ms": round(t * 1e3, 4),
                "visible_form": {"visible_points": int(nvis), "reproject_call_ms": round(float(np.median(tr)) * 1e3, 4),
                                 "search_call_ms": round(float(np.median(tv)) * 1e3, 4)},
                "brute_force_pairs": int(len(inl)) * npts,
                "window_gate_tests": tests, "gate_tests_per_s": round(tests / t, 1),
                "brute_force_pairs_per_s_equivalent": round(len(inl) * npts / t, 1)}


if __name__ == "__main__":
    # usage: time_projection.py [visible_points [max_inliers [reps]]] (default: config 2, then a dense map)
    if len(sys.argv) > 1:
        a = [int(x) for x in sys.argv[1:]] + [100000, 20][len(sys.argv) - 2:]
        print(json.dumps(case(a[0], a[1], a[2])))
    else:
        print(json.dumps(case(20000, 100000, 50)))
        print(json.dumps(case(1 << 17, 200, 5)))
