#!/usr/bin/env python3
"""Side benchmarks for the other BASELINE configs (not the driver's bench.py):
  2  1241x376 stream, 3000 features, search-by-projection against 20k map points
  3  1920x1080 batch, 5000 features
  5  3840x2160, 10000 features
Prints one JSON object per config.  Host-pointer calls are timed end to end
(H2D + kernels + D2H); resident calls are timed with frames in HBM."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402


def timeit(fn, reps, sync):
    fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def extract_config(w, h, nfeat, batch, reps, seed, cpu):
    frames_np = synth.frame_stream(w, h, batch, seed=seed)
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(frames_np).to(dev)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=nfeat, max_width=w, max_height=h,
                  max_batch=batch, frame_slots=2 * batch + 2, map_capacity=0) as fe:
        t = timeit(lambda: fe.extract_resident(frames.data_ptr(), batch, w, h, w, w * h, 1), reps,
                   fe.sync)
        stages = {n: fe.time_stage(s, batch, 5) for n, s in
                  (("pyramid", 1), ("fast_blur", 2), ("select", 3), ("orient", 4), ("describe", 6))}
    out = {"config": f"{w}x{h}, {nfeat} features, batch {batch}", "frames_per_s": batch / t,
           "ms_per_batch": t * 1e3, "stage_ms": stages}
    if cpu:
        import oracle_lib as O
        p = abi.reference_params(fast_threshold=20, nfeatures=nfeat)
        t0 = time.perf_counter()
        O.extract(p, frames_np[0])
        out["cpu_port_ms_per_frame"] = (time.perf_counter() - t0) * 1e3
    return out


def config2(cpu):
    w, h, nfeat, npts = 1241, 376, 3000, 20000
    frames = synth.frame_stream(w, h, 2, seed=9)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=nfeat, max_width=w, max_height=h,
                  max_batch=2, map_capacity=npts) as fe:
        f0, f1 = fe.extract_batch(list(frames))
        m = fe.knn_match_frames(f0, f1)
        uv, pdesc = synth.local_map(f1.descriptors, f1.pt, npts, w, h, seed=11)
        ids = np.arange(npts, dtype=np.int32)
        fe.map_put(ids, pdesc)
        inl = m["idx0"][m["ratio_pass"] == 1].astype(np.int32)
        t_match = timeit(lambda: fe.knn_match_frames(f0, f1), 20, fe.sync)
        t_proj = timeit(lambda: fe.search_by_projection(f1, inl, uv, ids), 20, fe.sync)
        got = fe.search_by_projection(f1, inl, uv, ids)
        t_ext = timeit(lambda: fe.extract_keypoints(frames[1]), 20, fe.sync)  # recycles the slots
    out = {"config": "1241x376 stream, 3000 features, search-by-projection vs 20k map points",
           "keypoints": len(f1), "inliers": int(len(inl)), "matched_points": int((got >= 0).sum()),
           "extract_host_call_ms": t_ext * 1e3, "match_frames_call_ms": t_match * 1e3,
           "search_by_projection_call_ms": t_proj * 1e3,
           "pair_tests_per_s": len(inl) * npts / t_proj,
           "frames_per_s_extract_match_search": 1.0 / (t_ext + t_match + t_proj)}
    if cpu:
        import oracle_lib as O
        t0 = time.perf_counter()
        want = O.search_by_projection(f1.pt, f1.descriptors, inl, uv, pdesc, ids)
        out["cpu_port_search_ms"] = (time.perf_counter() - t0) * 1e3
        out["identical_to_cpu_port"] = bool(np.array_equal(got, want))
    return out


if __name__ == "__main__":
    cpu = "--no-cpu" not in sys.argv
    print(json.dumps(config2(cpu)))
    print(json.dumps(extract_config(1920, 1080, 5000, 16, 10, 100, cpu)))
    print(json.dumps(extract_config(3840, 2160, 10000, 4, 10, 5, cpu)))
