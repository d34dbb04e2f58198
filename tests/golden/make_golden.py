"""Regenerates tests/golden/front_end_golden.npz from the CPU checker on seeded
synthetic inputs.  The reference ships no vectors for this path (its hot path is
inside an un-vendored OpenCV), so these fixtures pin the checker against drift
and give the GPU tests a file-based target; they are not OpenCV outputs.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from slam_amd import abi, synth  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def build():
    out = {}
    small = synth.textured_canvas(200, 160, 23)
    p_orb = abi.reference_params(fast_threshold=20, nfeatures=60)
    p_ref = abi.reference_params(mode=abi.MODE_PROVIDED, fast_threshold=30)
    k, d = O.extract(p_orb, small)
    out["small_orb_kps"], out["small_orb_desc"] = k, d
    k, d = O.extract(p_ref, small)
    out["small_ref_kps"], out["small_ref_desc"] = k, d
    q, t = synth.descriptor_sets(64, 200, seed=3, dup_percent=5)
    out["match_small"] = O.match_knn2(q, t)
    # full-size cases as hashes
    a, b = synth.frame_pair(640, 480)
    p = abi.reference_params(fast_threshold=20, nfeatures=2000)
    ka, da = O.extract(p, a)
    kb, db = O.extract(p, b)
    m = O.match_knn2(da, db)
    out["pair640_counts"] = np.array([len(ka), len(kb), int(m["ratio_pass"].sum())])
    out["pair640_sha"] = np.array([sha(ka), sha(da), sha(kb), sha(db),
                                   sha(np.stack([m["idx0"], m["idx1"]])), sha(m["ratio_pass"])])
    kitti = synth.textured_canvas(1241, 376, 44)
    kr, dr = O.extract(abi.reference_params(mode=abi.MODE_PROVIDED, fast_threshold=30), kitti)
    out["kitti_ref_count"] = np.array([len(kr)])
    out["kitti_ref_sha"] = np.array([sha(kr), sha(dr)])
    # frame ingest (resize 0.99 on BGR), reprojection and the projection search
    g = synth.textured_canvas(200, 160, 31)
    bgr = np.ascontiguousarray(np.stack([np.roll(g, 1, 0), g, np.roll(g, 2, 1)], axis=2))
    small_resized = O.resize_linear_8u(bgr, 0.99, 0.99)
    out["resize_small"] = small_resized
    k, d = O.extract(p_orb, small_resized)
    out["resize_small_kps"], out["resize_small_desc"] = k, d
    pose, K, xyz, th = synth.camera_scene(400, 640, 480, 8)
    uv, idx = O.reproject(pose, K, xyz, th, 640, 480)
    out["reproject_uv"], out["reproject_idx"] = uv, idx
    xy = np.stack([ka["x"], ka["y"]], 1)
    muv, mdesc = synth.local_map(da, xy, 500, 640, 480, seed=4)
    kp_idx = np.arange(0, len(ka), 3, dtype=np.int32)
    out["search_hits"] = O.search_by_projection(xy, da, kp_idx, muv, mdesc, np.arange(500, dtype=np.int32))
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "front_end_golden.npz"), **build())
    print("written", os.path.join(HERE, "front_end_golden.npz"))
