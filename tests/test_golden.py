"""Golden fixtures (tests/golden/front_end_golden.npz, made by make_golden.py).
CPU: the checker still produces them.  GPU: the HIP path produces them."""
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as O
from slam_amd import abi, synth

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "front_end_golden.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_checker_reproduces_golden_small_cases():
    small = synth.textured_canvas(200, 160, 23)
    k, d = O.extract(abi.reference_params(fast_threshold=20, nfeatures=60), small)
    assert np.array_equal(k, G["small_orb_kps"]) and np.array_equal(d, G["small_orb_desc"])
    k, d = O.extract(abi.reference_params(mode=abi.MODE_PROVIDED, fast_threshold=30), small)
    assert np.array_equal(k, G["small_ref_kps"]) and np.array_equal(d, G["small_ref_desc"])
    q, t = synth.descriptor_sets(64, 200, seed=3, dup_percent=5)
    assert np.array_equal(O.match_knn2(q, t), G["match_small"])


def test_checker_reproduces_golden_hashes():
    a, b = synth.frame_pair(640, 480)
    p = abi.reference_params(fast_threshold=20, nfeatures=2000)
    ka, da = O.extract(p, a)
    kb, db = O.extract(p, b)
    m = O.match_knn2(da, db)
    assert list(G["pair640_counts"]) == [len(ka), len(kb), int(m["ratio_pass"].sum())]
    assert list(G["pair640_sha"]) == [sha(ka), sha(da), sha(kb), sha(db),
                                      sha(np.stack([m["idx0"], m["idx1"]])), sha(m["ratio_pass"])]


def _new_row_inputs():
    g = synth.textured_canvas(200, 160, 31)
    bgr = np.ascontiguousarray(np.stack([np.roll(g, 1, 0), g, np.roll(g, 2, 1)], axis=2))
    scene = synth.camera_scene(400, 640, 480, 8)
    return bgr, scene


def test_checker_reproduces_golden_ingest_reprojection_search():
    bgr, (pose, K, xyz, th) = _new_row_inputs()
    resized = O.resize_linear_8u(bgr, 0.99, 0.99)
    assert np.array_equal(resized, G["resize_small"])
    k, d = O.extract(abi.reference_params(fast_threshold=20, nfeatures=60), resized)
    assert np.array_equal(k, G["resize_small_kps"]) and np.array_equal(d, G["resize_small_desc"])
    uv, idx = O.reproject(pose, K, xyz, th, 640, 480)
    assert np.array_equal(uv, G["reproject_uv"]) and np.array_equal(idx, G["reproject_idx"])
    a, _ = synth.frame_pair(640, 480)
    ka, da = O.extract(abi.reference_params(fast_threshold=20, nfeatures=2000), a)
    xy = np.stack([ka["x"], ka["y"]], 1)
    muv, mdesc = synth.local_map(da, xy, 500, 640, 480, seed=4)
    kp_idx = np.arange(0, len(ka), 3, dtype=np.int32)
    hits = O.search_by_projection(xy, da, kp_idx, muv, mdesc, np.arange(500, dtype=np.int32))
    assert np.array_equal(hits, G["search_hits"]) and (hits >= 0).sum() > 50


@pytest.mark.gpu
def test_gpu_reproduces_golden_ingest_reprojection_search():
    from slam_amd.frontend import FrontEnd
    bgr, (pose, K, xyz, th) = _new_row_inputs()
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=60, max_width=640, max_height=480,
                  map_capacity=1024) as fe:
        f, resized = fe.extract_scaled_bgr(bgr, 0.99, 0.99)
        assert np.array_equal(resized, G["resize_small"])
        assert np.array_equal(f.keypoints, G["resize_small_kps"])
        assert np.array_equal(f.descriptors, G["resize_small_desc"])
        fe.map_put_xyz(np.arange(len(xyz)), xyz)
        uv, idx = fe.reproject(pose, K, th, 640, 480, len(xyz))
        assert np.array_equal(uv, G["reproject_uv"]) and np.array_equal(idx, G["reproject_idx"])
    a, _ = synth.frame_pair(640, 480)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=640, max_height=480,
                  map_capacity=1024) as fe:
        fa = fe.extract_keypoints(a)
        muv, mdesc = synth.local_map(fa.descriptors, fa.pt, 500, 640, 480, seed=4)
        fe.map_put(np.arange(500), mdesc)
        kp_idx = np.arange(0, len(fa), 3, dtype=np.int32)
        hits = fe.search_by_projection(fa, kp_idx, muv, np.arange(500, dtype=np.int32))
        assert np.array_equal(hits, G["search_hits"])


@pytest.mark.gpu
def test_gpu_reproduces_golden():
    from slam_amd.frontend import FrontEnd
    small = synth.textured_canvas(200, 160, 23)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=60, max_width=640,
                  max_height=480) as fe:
        f = fe.extract_keypoints(small)
        assert np.array_equal(f.keypoints, G["small_orb_kps"])
        assert np.array_equal(f.descriptors, G["small_orb_desc"])
        q, t = synth.descriptor_sets(64, 200, seed=3, dup_percent=5)
        assert np.array_equal(fe.knn_match(q, t), G["match_small"])
    with FrontEnd(mode=abi.MODE_PROVIDED, fast_threshold=30, max_width=1241, max_height=376) as fe:
        f = fe.extract_keypoints(small)
        assert np.array_equal(f.keypoints, G["small_ref_kps"])
        assert np.array_equal(f.descriptors, G["small_ref_desc"])
        f = fe.extract_keypoints(synth.textured_canvas(1241, 376, 44))
        assert [len(f)] == list(G["kitti_ref_count"])
        assert [sha(f.keypoints), sha(f.descriptors)] == list(G["kitti_ref_sha"])
    a, b = synth.frame_pair(640, 480)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=640,
                  max_height=480, max_batch=2) as fe:
        fa, fb = fe.extract_batch([a, b])
        m = fe.knn_match_frames(fa, fb)
        assert list(G["pair640_counts"]) == [len(fa), len(fb), int(m["ratio_pass"].sum())]
        assert list(G["pair640_sha"]) == [sha(fa.keypoints), sha(fa.descriptors), sha(fb.keypoints),
                                          sha(fb.descriptors),
                                          sha(np.stack([m["idx0"], m["idx1"]])), sha(m["ratio_pass"])]


@pytest.mark.gpu
def test_cxx_host_adapter_against_checker(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "test_host_adapter")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    a, b = synth.frame_pair(640, 480)
    pa, pb = tmp_path / "a.raw", tmp_path / "b.raw"
    a.tofile(pa)
    b.tofile(pb)
    r = subprocess.run([exe, str(pa), str(pb), "640", "480"], capture_output=True, timeout=110)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b"host adapter ok" in r.stdout


@pytest.mark.gpu
def test_cxx_tracking_loop_replay_against_checker(tmp_path):
    # tests/cpp/test_tracking_loop.cpp: 40 frames of the reference's per-frame loop
    # (Init.cc:118-328) through the C++ host layer with a growing map, rewritten
    # coordinates every 10 frames and a wrapping frame-slot ring; every per-frame
    # product equals the CPU checker's
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "test_tracking_loop")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    n, w, h = 40, 320, 240
    frames = synth.frame_stream(w, h, n, seed=15)  # pans 3 px right, 2 px down per frame
    path = tmp_path / "frames.raw"
    frames.tofile(path)
    r = subprocess.run([exe, str(path), str(w), str(h), str(n), "3", "2"], capture_output=True, timeout=110)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b"tracking loop ok" in r.stdout


@pytest.mark.gpu
def test_c_caller_of_batch_entry_points():
    # tests/cpp/abi_batch_check.c: batch, per-frame, download and match calls agree
    # bit for bit and the error paths answer as the header documents
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "abi_batch_check")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, timeout=110)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b"abi batch check ok" in r.stdout
