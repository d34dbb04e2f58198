#!/usr/bin/env python3
"""Benchmark of the feature front-end on MI355X.

Metric (BASELINE.json): frames/s of FAST+ORB extraction + brute-force Hamming
matching on 640x480 frames with 2000 features (8-level x1.2 pyramid, FAST
threshold 20), plus G Hamming comparisons/s.

One "step" = one batch of BATCH synthetic frames, already resident in HBM,
through the whole hot path: pyramid, FAST+NMS, selection, orientation, blur,
rBRIEF, then one k=2 ratio-test match per frame against its predecessor (the
per-frame loop of Init.cc:130-131).  Nothing is skipped or cached between
steps.  With --gpus N every rank runs the same per-GPU workload on its own
frames (frame batches shard with no collective): weak scaling, value = total
frames/s.

Also reported in the same JSON line (never part of `value`):
  roofline        the dominant kernel against the HBM roofline (algorithmic bytes
                  per launch / HIP-event time on the launch stream) and the
                  instruction-issue bound that actually limits it
  parity_checked_frames   frames of the last timed step whose keypoints,
                  descriptors and match rows were compared with the CPU checker
  host_path       what the reference's host code can call (Init.cc:124-131):
                  host frames in, results back -- batch rate and single-call times
  configs         one leg per other BASELINE config (KITTI stream + search by
                  projection, 256 x 1080p split over the ranks, 3840x2160)
  cpu_baseline    the scalar CPU port of the same path, timed on this host
  hamming         G Hamming comparisons/s of the match kernel; with N > 1 also
                  the sharded relocalisation case (2000 queries x 1M map rows,
                  map split over the ranks, RCCL all-gather of per-shard top-2)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from slam_amd import abi, sharding, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402

W, H, NFEAT, THR = 640, 480, 2000, 20
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TOPS = 41.0  # measured: tools/ubench_valu.hip, v_xor + accumulating v_bcnt
VALU_ISSUE_PER_S = 1024 * 2.4e9 / 4  # wave-instructions/s: 1024 SIMDs, one per 4 cycles, 2.4 GHz
MFMA_FP4_PEAK_TOPS = 4096 * 1024 * 2.4e9 / 1e12  # dense FP4 MFMA peak (twice int8 / fp8), 10066 TOP/s
# k_match_mfma, per 32 x 32 tile of comparisons and wave: 4 matrix instructions (each
# holds the SIMD's vector issue for 8 cycles = 2 slots) and 28 vector instructions (20
# of the top-2 update, 8 of key shifting, row expansion and addressing: 112 per stage
# of four tiles, counted in the compiled stage loop)
MATCH_ISSUE_SLOTS_PER_TILE = 4 * 2 + 28

HOST_GROUP = None  # gloo group of the ranks, for barriers between host processes


def host_barrier():
    dist.barrier(group=HOST_GROUP)


STAGES = {"pyramid": 1, "fast_blur": 2, "select": 3, "orient": 4, "describe": 6, "match": 7}
MATCH_FIELDS = ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound")


def level_sizes(w, h, nlevels=8, sf=1.2):
    out = []
    for l in range(nlevels):
        scale = np.float32(np.float64(np.float32(sf)) ** l)
        inv = np.float32(1.0) / scale
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def algorithmic_bytes_per_frame(w, h):
    """Compulsory u8 traffic of each streaming stage for one frame."""
    px = [a * b for a, b in level_sizes(w, h)]
    return {
        "pyramid": sum(px[:-1]) + sum(px[1:]),  # each resize reads level l-1, writes level l
        # the fused FAST + blur kernel reads every level once and writes the
        # blurred level once (as separate passes the two stages would move
        # 3 * sum(px): FAST read, blur read, blur write)
        "fast_blur": 2 * sum(px),
    }


def checker():
    """The scalar CPU port under oracle/ (test infrastructure).  Used here only for
    the cpu_baseline leg and to check outputs after the timed regions."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    return O


def host_threads():
    return max(1, min(16, len(os.sched_getaffinity(0))))


def cpu_baseline(frames, budget_s=12.0):
    """The scalar CPU port on a bounded sample of the same workload: extract +
    match against the previous frame, on one thread and on all host threads
    (frames dealt to the threads; the port itself is single-threaded, as
    OpenCV's FAST/ORB path is per image)."""
    from concurrent.futures import ThreadPoolExecutor
    O = checker()
    p = abi.reference_params(fast_threshold=THR, nfeatures=NFEAT)
    prev = None
    n = 0
    t0 = time.perf_counter()
    while True:
        img = frames[n % len(frames)]
        _, desc = O.extract(p, img)
        O.match_knn2(prev if prev is not None else desc, desc)
        prev = desc
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 4 * len(frames):
            break
    out = {"value": n / el, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{n} frames of the same 640x480 stream, extract + 1 match each, "
                     f"{el:.1f} s, scalar C -O3, 1 thread"}
    nthr = host_threads()

    def chain(ix):  # one thread's share: consecutive frames, each matched with its predecessor
        prev = None
        for i in ix:
            _, d = O.extract(p, frames[i % len(frames)])
            O.match_knn2(prev if prev is not None else d, d)
            prev = d
        return len(ix)
    per = max(2, int(out["value"] * 6.0))  # about 6 s of work per thread
    with ThreadPoolExecutor(nthr) as ex:
        t0 = time.perf_counter()
        done = sum(ex.map(chain, [range(t * per, (t + 1) * per) for t in range(nthr)]))
        par = time.perf_counter() - t0
    out["all_threads"] = {"value": round(done / par, 2), "unit": "frames/s", "cores": nthr,
                          "sample": f"{done} frames, {par:.1f} s, {nthr} threads"}
    # matching alone: one thread, and query rows split over the host threads
    # (OpenCV's batchDistance runs under parallel_for_)
    q = synth.random_descriptors(2000, 3, 5)
    t = synth.random_descriptors(100000, 4, 6)
    t0 = time.perf_counter()
    O.match_knn2(q, t)
    one = time.perf_counter() - t0
    parts = [np.ascontiguousarray(q[ix]) for ix in np.array_split(np.arange(len(q)), nthr)]
    with ThreadPoolExecutor(nthr) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda a: O.match_knn2(a, t), parts))
        par = time.perf_counter() - t0
    cm = len(q) * len(t) / 1e9
    out["hamming_gcmps_1_thread"] = round(cm / one, 3)
    out["hamming_gcmps_threads"] = round(cm / par, 3)
    out["hamming_threads"] = nthr
    return out


def hbm_copy_ceiling(dev, nbytes=1 << 30, reps=10):
    """Device-to-device copy rate (read + write bytes per second): the practical
    ceiling an HBM-streaming kernel can be held against."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2 * nbytes / (ms * 1e-3) / 1e9


def pmc_profile():
    """Counter readings of the committed rocprofv3 --pmc passes (they cannot be
    collected inside this process), with their provenance: `source` names the file
    and the commit they were measured on, `binary_matches` says whether the library
    loaded by this run is byte for byte the one the counters were collected on
    (tools/update_pmc_traffic.py records its SHA-256).  A mismatch is reported on
    standard error: the counters then describe other kernels than the ones timed."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        prof = json.load(open(tpath))
    except Exception:
        return {}
    on = prof.get("measured_on", {})
    prof["source"] = "profiles/pmc_traffic.json @ " + on.get("commit", "unknown commit")
    want = on.get("libslamfe_sha256")
    try:
        import hashlib
        from slam_amd._lib import LIB_PATH
        have = hashlib.sha256(open(LIB_PATH, "rb").read()).hexdigest()
    except Exception:
        have = None
    prof["binary_matches"] = bool(want) and want == have
    if not prof["binary_matches"]:
        print(f"bench.py: warning: the counters in {tpath} were collected on another build of libslamfe.so "
              f"(recorded {str(want)[:12]}, loaded {str(have)[:12]}); roofline.traffic and limiter.wave_insts_per_launch "
              "are not of the kernels timed here -- re-run tools/collect_profiles.sh and tools/update_pmc_traffic.py",
              file=sys.stderr)
    return prof


def hbm_roofline(kernel, alg_bytes, ms, extra=None):
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    r = {"kernel": kernel, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(ms, 5)}
    if extra:
        r.update(extra)
    return r


def timed(fn, reps, sync, warm=2):
    for _ in range(warm):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    sync()
    return (time.perf_counter() - t0) / reps


def call_times_ms(fn, reps, warm=5):
    """(median, mean) milliseconds of one synchronous call: the median is the
    call's latency, the mean also carries the occasional host hiccup."""
    for _ in range(warm):
        fn()
    ts = np.empty(reps)
    for i in range(reps):
        t0 = time.perf_counter()
        fn()
        ts[i] = time.perf_counter() - t0
    return float(np.median(ts)) * 1e3, float(ts.mean()) * 1e3


class HostCall:
    """slamfe_extract / slamfe_extract_batch / slamfe_match_frames through the C ABI
    with preallocated outputs: what a C++ host pays, without Python allocations."""

    def __init__(self, fe, nframes=1):
        self.fe = fe
        cap = fe.max_keypoints
        self.cap = cap
        self.kps = np.zeros((nframes, cap), abi.KEYPOINT_DTYPE)
        self.desc = np.zeros((nframes, cap, 32), np.uint8)
        self.rows = np.zeros(cap, abi.MATCH2_DTYPE)
        self.counts = np.zeros(nframes, np.int32)
        self.handles = np.zeros(nframes, np.uint64)
        self.n = C.c_int32()

    def extract(self, img):
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else 3
        rc = self.fe._lib.slamfe_extract(self.fe._ctx, img.ctypes.data, w, h, w * ch, ch, self.kps.ctypes.data,
                                         self.desc.ctypes.data, self.cap,
                                         self.counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                         self.handles.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0, rc
        return int(self.handles[0])

    def extract_batch(self, ptrs, n, w, h):
        rc = self.fe._lib.slamfe_extract_batch(self.fe._ctx, ptrs, n, w, h, w, 1, self.kps.ctypes.data,
                                               self.desc.ctypes.data, self.cap,
                                               self.counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                               self.handles.ctypes.data_as(C.POINTER(C.c_uint64)))
        assert rc == 0, rc

    def match_frames(self, hq, ht):
        rc = self.fe._lib.slamfe_match_frames(self.fe._ctx, C.c_uint64(hq), C.c_uint64(ht),
                                              self.rows.ctypes.data, self.cap, C.byref(self.n))
        assert rc == 0, rc
        return self.n.value


def leg_host_path(local_rank):
    """The path the reference's tracking thread can call today (Init.cc:124-131):
    frames in pageable host memory, keypoints / descriptors / match rows back in
    host memory, one context, one host thread."""
    out = {}
    B = 64
    frames = list(synth.frame_stream(W, H, B, seed=1))
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=THR, nfeatures=NFEAT, max_width=W, max_height=H,
                  max_batch=B, frame_slots=2 * B, device_id=local_rank, map_capacity=0) as fe:
        hc = HostCall(fe, B)
        ptrs = (C.c_void_p * B)(*[f.ctypes.data for f in frames])
        t = timed(lambda: hc.extract_batch(ptrs, B, W, H), 10, fe.sync)
        out["host_batch_fps"] = round(B / t, 1)
        out["host_batch"] = {"frames_per_call": B, "ms_per_call": round(t * 1e3, 4),
                             "what": "slamfe_extract_batch: 64 pageable 640x480 frames in, keypoints + "
                                     "descriptors back to host memory (PCIe both ways)"}
    single = {}
    native = os.path.join(ROOT, "tools", "single_call_bench")
    native_cxx = os.path.join(ROOT, "tools", "cxx_adapter_bench")
    cases = (("vga_orb_2000_thr20", W, H, NFEAT, THR, abi.MODE_ORB),
             ("kitti_1241x376_provided_thr30", 1241, 376, 10000, 30, abi.MODE_PROVIDED),
             ("kitti_1241x376_orb_3000_thr20", 1241, 376, 3000, 20, abi.MODE_ORB))
    for name, w, h, nf, thr, mode in cases:
        a, b = synth.frame_stream(w, h, 2, seed=3)
        with FrontEnd(mode=mode, fast_threshold=thr, nfeatures=nf, max_width=w, max_height=h,
                      device_id=local_rank, map_capacity=0) as fe:
            hc = HostCall(fe)
            ext_med, ext_mean = call_times_ms(lambda: hc.extract(a), 100)
            st = {"prev": hc.extract(a)}

            def loop():  # Init.cc:130-131: extract the new frame, match the previous one against it
                cur = hc.extract(b)
                hc.match_frames(st["prev"], cur)
                st["prev"] = cur
            loop_med, loop_mean = call_times_ms(loop, 100)
            single[name] = {"extract_ms": round(ext_med, 4), "extract_mean_ms": round(ext_mean, 4),
                            "extract_plus_match_ms": round(loop_med, 4),
                            "extract_plus_match_mean_ms": round(loop_mean, 4),
                            "frames_per_s": round(1e3 / loop_mean, 1), "keypoints": int(hc.counts[0]),
                            "caller": "python (ctypes)"}
        # the same two loops from a C program (no interpreter between the calls): what
        # the reference's C++ host would see
        if os.path.exists(native):
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                fa, fb = os.path.join(td, "a.raw"), os.path.join(td, "b.raw")
                a.tofile(fa)
                b.tofile(fb)
                r = subprocess.run([native, fa, fb, str(w), str(h), str(mode), str(thr), str(nf), "200"],
                                   capture_output=True, text=True, timeout=120)
                r2 = None
                if os.path.exists(native_cxx):
                    r2 = subprocess.run([native_cxx, fa, fb, str(w), str(h), str(mode), str(thr), str(nf), "200"],
                                        capture_output=True, text=True, timeout=120)
            if r.returncode == 0:
                c = json.loads(r.stdout.strip().splitlines()[-1])
                c["frames_per_s"] = round(1e3 / c["extract_plus_match_mean_ms"], 1)
                c["caller"] = "C (tools/single_call_bench.c)"
                single[name + "_c_caller"] = c
            else:
                single[name + "_c_caller"] = {"error": r.stderr[-300:]}
            # ... and through the C++ host layer the patched Init.cc binds: containers filled,
            # knnMatch on the two frames' handles, the hand-over of Init.cc:322-323
            if r2 is not None and r2.returncode == 0:
                c = json.loads(r2.stdout.strip().splitlines()[-1])
                c["frames_per_s"] = round(1e3 / c["extract_plus_match_mean_ms"], 1)
                c["caller"] = "C++14 slamfe::FrontEnd (tools/cxx_adapter_bench.cpp)"
                cc = single.get(name + "_c_caller", {})
                if "extract_plus_match_ms" in cc:
                    c["over_c_caller"] = round(c["extract_plus_match_ms"] / cc["extract_plus_match_ms"], 3)
                single[name + "_cxx_adapter"] = c
            elif r2 is not None:
                single[name + "_cxx_adapter"] = {"error": r2.stderr[-300:]}
    out["single_call_ms"] = single
    return out


def window_gate_tests(xy, uv, radius, max_w, max_h):
    """Pixel-gate tests the grid query of slamfe_search_by_projection makes: the points
    in the 3 x 3 cells around each inlier keypoint (cells a little wider than the
    radius, at most 128 per axis: the library's rule, restated)."""
    cell = max(radius * 1.0001 + 1e-3, max(max_w, max_h) / 128.0)
    gx = int(min(128, np.floor(max_w / cell) + 1))
    gy = int(min(128, np.floor(max_h / cell) + 1))
    cx = np.clip(np.floor(uv[:, 0].astype(np.float64) / cell), 0, gx - 1).astype(np.int64)
    cy = np.clip(np.floor(uv[:, 1].astype(np.float64) / cell), 0, gy - 1).astype(np.int64)
    cnt = np.zeros((gy + 2, gx + 2), np.int64)
    np.add.at(cnt, (cy + 1, cx + 1), 1)
    win = sum(cnt[1 + dy:gy + 1 + dy, 1 + dx:gx + 1 + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1))
    kx = np.clip(np.floor(xy[:, 0].astype(np.float64) / cell), 0, gx - 1).astype(np.int64)
    ky = np.clip(np.floor(xy[:, 1].astype(np.float64) / cell), 0, gy - 1).astype(np.int64)
    return int(win[ky, kx].sum())


def leg_two_streams(frames, B, dev, local_rank, steps):
    """The headline workload from two contexts on two HIP streams at once (each its
    own 256-frame batches): kernels bound by different units (vector issue in FAST +
    blur, L1 lines in orientation, matrix pipe in matching) share the chip.  Reported
    beside the headline, which stays one stream so that its per-kernel times are those
    of kernels that have the chip to themselves."""
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    fes = [FrontEnd(mode=abi.MODE_ORB, fast_threshold=THR, nfeatures=NFEAT, max_width=W, max_height=H, max_batch=B,
                    frame_slots=2 * B + 2, device_id=local_rank, stream=st.cuda_stream, map_capacity=0)
           for st in streams]
    outs = [torch.empty((B, fes[0].max_keypoints, 16), dtype=torch.uint8, device=dev) for _ in fes]
    last = [None, None]

    def step():
        for i, f in enumerate(fes):
            h = f.extract_resident(frames.data_ptr(), B, W, H, W, W * H, 1)
            first = int(h[0])
            if last[i] is not None:
                f.match_resident_batch(first - 1, first, B, outs[i].data_ptr())
            last[i] = first + B - 1

    try:
        for _ in range(30):
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
    finally:
        for f in fes:
            f.close()
    return {"what": f"two contexts / HIP streams, {B} frames per stream and step", "frames_per_s": round(2 * B * steps / el, 1),
            "frames_per_step": 2 * B, "ms_per_step": round(el / steps * 1e3, 4)}


def leg_config2(local_rank, check):
    """BASELINE config 2: KITTI-size 1241x376 stream, 3000 features per frame,
    search by projection against 20 000 local map points; one frame at a time
    through the host-pointer calls, as Init.cc:124-259 would drive it."""
    w, h, nfeat, npts = 1241, 376, 3000, 20000
    frames = synth.frame_stream(w, h, 2, seed=9)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=THR, nfeatures=nfeat, max_width=w, max_height=h,
                  max_batch=2, device_id=local_rank, map_capacity=npts) as fe:
        f0, f1 = fe.extract_batch(list(frames))
        m = fe.knn_match_frames(f0, f1)
        uv, pdesc = synth.local_map(f1.descriptors, f1.pt, npts, w, h, seed=11)
        ids = np.arange(npts, dtype=np.int32)
        fe.map_put(ids, pdesc)
        inl = m["idx0"][m["ratio_pass"] == 1].astype(np.int32)
        hc = HostCall(fe)
        t_ext = timed(lambda: hc.extract(frames[1]), 30, fe.sync)
        fe.probe_stage(STAGES["fast_blur"])
        for _ in range(8):
            hc.extract(frames[1])
        fb_ms, _ = fe.probe_read()
        fe.probe_stage(0)
        f0, f1 = fe.extract_batch(list(frames))
        t_match = timed(lambda: hc.match_frames(f0.handle, f1.handle), 30, fe.sync)
        matched = np.zeros(len(inl), np.int32)

        def search():
            rc = fe._lib.slamfe_search_by_projection(fe._ctx, C.c_uint64(f1.handle), inl.ctypes.data, len(inl),
                                                     uv.ctypes.data, ids.ctypes.data, npts, C.c_float(32.0), 128,
                                                     matched.ctypes.data)
            assert rc == 0, rc
        t_proj = timed(search, 30, fe.sync)
        got = matched.copy()
        # The same search fed by the device-side reprojection (Init.cc:203-218), as the
        # patched loop runs it: slamfe_reproject leaves the visible set on the device and
        # queues the search grid behind it; the search call is candidates + resolve.
        pose, Kc, xyz, th = synth.scene_from_uv(uv, w, h)
        fe.map_put_xyz(ids, xyz)
        nvis = C.c_int32()
        pose_c, K_c = np.ascontiguousarray(pose), np.ascontiguousarray(Kc)
        matched_v = np.zeros(len(inl), np.int32)

        def reproj():
            rc = fe._lib.slamfe_reproject(fe._ctx, pose_c.ctypes.data, K_c.ctypes.data, C.c_float(th),
                                          C.c_float(w), C.c_float(h), npts, None, None, 0, C.byref(nvis))
            assert rc == 0, rc

        def search_vis():
            rc = fe._lib.slamfe_search_by_projection_visible(fe._ctx, C.c_uint64(f1.handle), inl.ctypes.data,
                                                             len(inl), C.c_float(32.0), 128, matched_v.ctypes.data)
            assert rc == 0, rc
        for _ in range(5):
            reproj()
            search_vis()
        tr, tv = [], []
        for _ in range(40):
            a = time.perf_counter()
            reproj()
            b = time.perf_counter()
            search_vis()
            tv.append(time.perf_counter() - b)
            tr.append(b - a)
        t_reproj, t_vis = float(np.median(tr)), float(np.median(tv))
        vis_uv, vis_idx = fe.reproject(pose, Kc, th, w, h, npts)
        search_vis()
        got_v = matched_v.copy()
    alg = algorithmic_bytes_per_frame(w, h)["fast_blur"]
    tests = window_gate_tests(f1.pt[inl], uv, 32.0, w, h)
    out = {"workload": "1241x376 stream, 3000 features, search by projection vs 20k map points, one frame "
                       "per call, host frames in / results out",
           "keypoints": len(f1), "inliers": int(len(inl)), "matched_points": int((got >= 0).sum()),
           "extract_call_ms": round(t_ext * 1e3, 4), "match_call_ms": round(t_match * 1e3, 4),
           "search_by_projection_call_ms": round(t_proj * 1e3, 4),
           "device_reprojection": {"visible_points": int(nvis.value), "reproject_call_ms": round(t_reproj * 1e3, 4),
                                   "search_visible_call_ms": round(t_vis * 1e3, 4),
                                   "matched_points": int((got_v >= 0).sum())},
           "frames_per_s": round(1.0 / (t_ext + t_match + t_reproj + t_vis), 1),
           "search": {"brute_force_pairs": int(len(inl)) * npts,
                      "window_gate_tests": tests,
                      "gate_tests_per_inlier": round(tests / max(1, len(inl)), 1),
                      "gate_tests_per_s": round(tests / t_proj, 1),
                      "brute_force_pairs_per_s_equivalent": round(len(inl) * npts / t_proj, 1)},
           "roofline": hbm_roofline("k_fast_blur (one 1241x376 frame per launch: latency-bound)", alg, fb_ms)}
    if check:
        O = checker()
        t0 = time.perf_counter()
        want = O.search_by_projection(f1.pt, f1.descriptors, inl, uv, pdesc, ids)
        out["cpu_port_search_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
        assert np.array_equal(got, want), "search by projection differs from the CPU checker"
        want_v = O.search_by_projection(f1.pt, f1.descriptors, inl, vis_uv, pdesc[vis_idx], vis_idx)
        assert np.array_equal(got_v, want_v), "search over the device-side visible set differs from the CPU checker"
        out["identical_to_cpu_port"] = True
    return out


def leg_extract_config(name, w, h, nfeat, total_frames, distinct, dev, local_rank, world, rank, multi, check,
                       include_h2d):
    """A fixed batch of `total_frames` frames split over the ranks (strong scaling,
    no collective on the data path): each rank uploads its share from pinned
    host memory (when include_h2d) and extracts it; results stay on the device."""
    lo, hi = sharding.frame_bounds(total_frames, world, rank)
    mine = hi - lo
    uniq = synth.frame_stream(w, h, min(distinct, total_frames), seed=100)
    idx = [i % len(uniq) for i in range(lo, hi)]
    host = np.ascontiguousarray(uniq[idx]) if mine else np.zeros((0, h, w), np.uint8)
    res = {"workload": f"{total_frames} frames of {w}x{h}, {nfeat} features each, split over {world} rank(s) "
                       f"({len(uniq)} distinct frames repeated)", "frames_this_rank": mine}
    B = max(1, mine)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=THR, nfeatures=nfeat, max_width=w, max_height=h,
                  max_batch=B, frame_slots=B + 2, device_id=local_rank, map_capacity=0) as fe:
        d = torch.empty((B, h, w), dtype=torch.uint8, device=dev)
        # the frames wait in pinned host memory, where a capture pipeline would put them
        host_t = torch.from_numpy(host).pin_memory() if mine else None
        state = {}

        def run(h2d):
            if mine == 0:
                return
            if h2d:
                d[:mine].copy_(host_t, non_blocking=True)  # host memory -> HBM over PCIe
                torch.cuda.synchronize(dev)  # torch's stream; the library runs on its own
            state["hd"] = fe.extract_resident(d.data_ptr(), mine, w, h, w, w * h, 1)

        def measure(h2d, reps):
            run(h2d)
            fe.sync()
            if multi:
                host_barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                run(h2d)
            fe.sync()
            if multi:
                host_barrier()
            torch.cuda.synchronize(dev)
            el = (time.perf_counter() - t0) / reps
            if multi:
                t = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el
        if mine:
            d[:mine].copy_(host_t)
        torch.cuda.synchronize(dev)
        t_res = measure(False, 5)
        res["resident_fps"] = round(total_frames / t_res, 1)
        res["resident_ms_per_batch"] = round(t_res * 1e3, 3)
        if include_h2d:
            t_h2d = measure(True, 3)
            res["with_h2d_fps"] = round(total_frames / t_h2d, 1)
            res["with_h2d_ms_per_batch"] = round(t_h2d * 1e3, 3)
        if mine:
            fe.probe_stage(STAGES["fast_blur"])
            for _ in range(4):
                run(False)
            fb_ms, _ = fe.probe_read()
            fe.probe_stage(0)
            res["roofline"] = hbm_roofline("k_fast_blur", algorithmic_bytes_per_frame(w, h)["fast_blur"] * mine, fb_ms,
                                           {"frames_per_launch": mine})
            stage = {}
            for sname, sid in STAGES.items():
                if sname != "match":
                    stage[sname] = round(fe.time_stage(sid, mine, 3), 4)
            res["stage_ms_back_to_back"] = stage
        if check and mine:
            O = checker()
            p = abi.reference_params(fast_threshold=THR, nfeatures=nfeat)
            checked = []
            for k in sorted({0, min(7, mine - 1), min(8, mine - 1), mine - 1}):
                f = fe.frame_download(int(state["hd"][k]))
                t0 = time.perf_counter()
                kps, desc = O.extract(p, host[k])
                res["cpu_port_ms_per_frame"] = round((time.perf_counter() - t0) * 1e3, 1)
                assert np.array_equal(f.keypoints, kps) and np.array_equal(f.descriptors, desc), \
                    f"{name}: frame {k} differs from the CPU checker"
                checked.append(lo + k)
            res["parity_checked_frames"] = checked
    res["scaling"] = "strong"
    return res


def main():
    # Standard output carries the one JSON line and nothing else: whatever libraries
    # print there meanwhile (RCCL's version banner at the first collective, for one)
    # is sent to standard error.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024,
                    help="frames per step (one launch sequence): the fixed cost of a launch sequence is "
                         "about 4 %% of a 256-frame step and 1 %% of a 1024-frame one")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed load before the warm-up steps so that the GPU clock has settled")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent frame streams per GPU, each with its own context and HIP stream")
    ap.add_argument("--no-cpu", action="store_true",
                    help="skip everything that runs the CPU checker (baseline leg and output checks)")
    ap.add_argument("--no-reloc", action="store_true", help="skip the 1M-row matching leg")
    ap.add_argument("--no-legs", action="store_true", help="skip the host-path and other-config legs")
    ap.add_argument("--reloc-sets-per-collective", type=int, default=1,
                    help="query sets of 2000 matched per all-gather in the sharded relocalisation leg "
                         "(1, 4 and 16 are always reported; this adds another value)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (process group, barriers, all-gather) even "
                         "with one rank: a rehearsal of the N > 1 run on a one-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=dev)
        # The barriers that bracket the timed steps run between host processes (each
        # rank synchronises its GPU first), over a gloo group: an RCCL barrier is a
        # collective launch plus a device synchronisation of its own, about 2 ms here --
        # 10 % of twenty 1 ms steps that is not the workload's.
        global HOST_GROUP
        HOST_GROUP = dist.new_group(backend="gloo")

    B = args.batch
    # each rank gets its own frames (different pan offset / noise per rank)
    frames_np = synth.frame_stream(W, H, B, seed=1 + 17 * rank)
    frames = torch.from_numpy(frames_np).to(dev)
    S = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    fes = [FrontEnd(mode=abi.MODE_ORB, fast_threshold=THR, nfeatures=NFEAT, max_width=W,
                    max_height=H, max_batch=B, frame_slots=2 * B + 2, device_id=local_rank,
                    stream=st.cuda_stream, map_capacity=0) for st in streams]
    fe, stream = fes[0], streams[0]
    cap = fe.max_keypoints
    match_outs = [torch.empty((B, cap, 16), dtype=torch.uint8, device=dev) for _ in range(S)]
    states = [{"last": None} for _ in range(S)]

    def step():
        hd = None
        for f, mo, state in zip(fes, match_outs, states):
            h = f.extract_resident(frames.data_ptr(), B, W, H, W, W * H, 1)
            first = int(h[0])
            # frame i is matched against frame i-1; frame 0 against the last frame
            # of the previous step (on the very first step: against itself)
            # (handles are consecutive, so from the second step on the previous
            # step's last frame is handle first - 1 and all B pairs go in one call)
            if state["last"] is not None and state["last"] == first - 1:
                f.match_resident_batch(first - 1, first, B, mo.data_ptr())
            else:
                f.match_resident_batch(first, first, 1, mo.data_ptr())
                if B > 1:
                    f.match_resident_batch(first, first + 1, B - 1, mo[1:].data_ptr())
            state["last"] = first + B - 1
            hd = hd if hd is not None else h
        return hd

    with torch.cuda.stream(stream):
        # Untimed, before the W warm-up steps: keep the GPU under this load for a
        # fraction of a second.  The chip raises its clock only after tens of
        # milliseconds of work, and a short run (few steps of ~1.3 ms) would
        # otherwise be timed partly at the idle clock.
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            for _ in range(10):
                step()
            torch.cuda.synchronize(dev)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(dev)
        if multi:
            host_barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hd = step()
        torch.cuda.synchronize(dev)
        if multi:
            host_barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # The outputs of the last timed step, kept for the check against the CPU
    # checker further down: frames 0, 7, 8 and B-1 (both sides of the 8-frame
    # grouping of the launch grids, and the last frame), each with the frame it
    # was matched against and its match rows.
    check_ix = sorted({0, min(7, B - 1), min(8, B - 1), B - 1})
    need = sorted({i for k in check_ix for i in (k, (k - 1) % B)})
    kept = {i: fe.frame_download(int(hd[i])) for i in need}
    kept_rows = {k: match_outs[0][k].cpu().numpy().reshape(-1).view(abi.MATCH2_DTYPE) for k in check_ix}
    all_counts = np.array([len(fe.frame_download(int(h))) for h in hd], dtype=np.int64)

    # per-stage kernel time inside the running pipeline: HIP events around the
    # stage on the stream the kernels are launched on, over extra (untimed) steps
    # of the same loop -- the same context the rocprofv3 averages come from.
    # (Launched back to back in isolation the dominant kernel runs up to 15 %
    # slower: kept as avg_launch_ms_back_to_back.)
    stage_ms = {}
    with torch.cuda.stream(stream):
        for name, sid in STAGES.items():
            fe.probe_stage(sid)
            for _ in range(8):
                step()
            stage_ms[name], nprobe = fe.probe_read()
            assert nprobe == 8, (name, nprobe)
        fe.probe_stage(0)
    isolated_ms = fe.time_stage(STAGES["fast_blur"], B, 10)
    alg = algorithmic_bytes_per_frame(W, H)
    hbm_stage = max(("pyramid", "fast_blur"), key=lambda k: stage_ms[k])
    dominant = max(stage_ms, key=lambda k: stage_ms[k])
    # the roofline object is given for the slowest HBM-streaming kernel
    # (the pyramid stage is seven launches; FAST + blur is one fused launch)
    roof_bytes = alg[hbm_stage] * B
    prof = pmc_profile()
    roofline = hbm_roofline("k_" + hbm_stage if hbm_stage != "pyramid" else "k_resize (7 launches)",
                            roof_bytes, stage_ms[hbm_stage],
                            {"avg_launch_ms_back_to_back": round(isolated_ms, 5),
                             "measured_copy_ceiling_GBps": round(hbm_copy_ceiling(dev), 1)})
    traffic = prof.get(hbm_stage)
    if traffic is not None:
        # measured on launches of prof["frames_per_launch"] frames; per frame-proportional
        roofline["traffic"] = int(traffic * B / prof.get("frames_per_launch", B))
        roofline["traffic_source"] = {"source": prof.get("source"), "binary_matches": prof.get("binary_matches")}
    winsts = prof.get(hbm_stage + "_wave_insts")
    if winsts is not None:
        # What actually bounds the kernel: vector-instruction issue.  One wave
        # instruction per 4 cycles per SIMD; the count comes from the committed
        # SQ_INSTS_* counter pass, the time from this run.
        winsts = winsts * B / prof.get("frames_per_launch", B)
        issue_ms = winsts / VALU_ISSUE_PER_S * 1e3
        roofline["limiter"] = {"bound": "vector instruction issue", "wave_insts_per_launch": int(winsts),
                               "peak_wave_insts_per_s": VALU_ISSUE_PER_S, "issue_time_ms": round(issue_ms, 5),
                               "frac": round(issue_ms / stage_ms[hbm_stage], 4),
                               "source": prof.get("source"), "binary_matches": prof.get("binary_matches")}
        minsts = prof.get(hbm_stage + "_mfma_insts")
        if minsts is not None:
            roofline["limiter"]["mfma_insts_per_launch"] = int(minsts * B / prof.get("frames_per_launch", B))

    # Hamming comparisons/s of the match kernel inside the frame pipeline
    # (a step matches B pairs in one call: frame i with frame i-1, frame 0 with
    # the last frame of the previous step)
    cmps = float((all_counts[:-1] * all_counts[1:]).sum() + all_counts[0] * all_counts[-1])
    gc = cmps / (stage_ms["match"] * 1e-3) / 1e9
    # The match kernel runs on the matrix cores: one 256-bit comparison = 256
    # FP4 multiply-accumulates = 512 operations of v_mfma_f32_32x32x64_f8f6f4, whose
    # dense peak is 4096 op/clk/SIMD x 1024 SIMDs x 2.4 GHz.  At that instruction's
    # rate the kernel is bound by vector issue (the top-2 update, 2 operations per
    # comparison and lane), which "vector_issue" prices: issue slots of 4 cycles per
    # SIMD.  (The integer-VALU formulation, 16 lane-ops per comparison, tops out at
    # 41.0 T lane-ops/s = 2.56 T cmp/s: measured by tools/ubench_valu.hip.)
    def match_roofline(gcmps, gpus=1):
        return {"bound": "mfma", "operands": "fp4 (E2M1), f32 accumulate", "ops_per_cmp": 512,
                "achieved": round(gcmps * 512 / 1e3, 1), "peak": MFMA_FP4_PEAK_TOPS * gpus, "unit": "TOP/s",
                "frac": round(gcmps * 512 / 1e3 / (MFMA_FP4_PEAK_TOPS * gpus), 4),
                "vector_issue": {"slots_per_1024_cmps": MATCH_ISSUE_SLOTS_PER_TILE,
                                 "frac": round(gcmps * 1e9 / 1024 * MATCH_ISSUE_SLOTS_PER_TILE
                                               / (VALU_ISSUE_PER_S * gpus), 4)}}

    hamming = {"frame_pairs_gcmps": round(gc, 2), "mfma_roofline": match_roofline(gc),
               "valu_formulation_ceiling_gcmps": round(VALU_PEAK_TOPS * 1e3 / 16, 1)}

    # relocalisation: 2000 queries against a 1M-row map sharded over the ranks
    if not args.no_reloc:
        try:
            NQ, NT = 2000, 1_000_000
            lo, hi = sharding.shard_bounds(NT, world, rank)
            shard = torch.from_numpy(synth.random_descriptors(hi - lo, 3 + 101 * rank, 6)).to(dev)
            R = 50

            def reloc_job(K):
                """K query sets per collective: one match launch for K x 2000 queries, one
                all-gather of K x 32 KB per rank, one merge launch.  Returns the functions
                of the three phases and the whole sequence."""
                dq = torch.from_numpy(synth.random_descriptors(K * NQ, 3, 5)).to(dev)
                keys = torch.empty((K * NQ, 2), dtype=torch.int64, device=dev)
                gathered = torch.empty((world, K * NQ, 2), dtype=torch.int64, device=dev)
                out = torch.empty((K * NQ, 16), dtype=torch.uint8, device=dev)

                def match(k=keys):
                    fe.match_shard_keys(dq.data_ptr(), K * NQ, shard.data_ptr(), hi - lo, lo, k.data_ptr())

                def gather():
                    if multi:
                        dist.all_gather_into_tensor(gathered, keys)

                def merge(g=None):
                    if multi:
                        fe.merge_shard_keys((gathered if g is None else g).data_ptr(), world, K * NQ, out.data_ptr())
                    else:
                        fe.merge_shard_keys(keys.data_ptr(), 1, K * NQ, out.data_ptr())

                def whole():
                    match()
                    gather()
                    merge()
                return {"match": match, "gather": gather, "merge": merge, "whole": whole, "keys": keys,
                        "gathered": gathered}

            def max_over_ranks(x):
                if not multi:
                    return x
                t = torch.tensor([x], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return float(t.item())

            def time_reloc(K):
                job = reloc_job(K)
                res = {}
                with torch.cuda.stream(stream):
                    for _ in range(50 if K == 1 else 10):  # (the chip takes some tens of launches to reach its clock)
                        job["whole"]()
                    torch.cuda.synchronize(dev)
                    if multi:
                        host_barrier()
                    torch.cuda.synchronize(dev)
                    r0 = time.perf_counter()
                    for _ in range(R):
                        job["whole"]()
                    torch.cuda.synchronize(dev)
                    if multi:
                        host_barrier()
                    rel = max_over_ranks(time.perf_counter() - r0)
                    res["gcmps"] = round(K * NQ * NT * R / rel / 1e9, 2)
                    res["ms_per_query_set"] = round(rel / R / K * 1e3, 4)
                    # the three phases between HIP events on the stream they are queued on
                    # (a synchronous collective makes this stream wait for its end)
                    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(R)]
                    for e in ev:
                        e[0].record()
                        job["match"]()
                        e[1].record()
                        job["gather"]()
                        e[2].record()
                        job["merge"]()
                        e[3].record()
                    torch.cuda.synchronize(dev)
                    ph = [float(np.mean([e[i].elapsed_time(e[i + 1]) for e in ev])) for i in range(3)]
                    res["phases_ms"] = {"match_kernel": round(max_over_ranks(ph[0]), 4),
                                        "all_gather": round(max_over_ranks(ph[1]), 4) if multi else None,
                                        "merge": round(max_over_ranks(ph[2]), 4)}
                    if multi:
                        # successive collectives pipelined: the all-gather of step i (on RCCL's
                        # stream) overlaps the match kernel of step i+1
                        keys2 = [job["keys"], torch.empty_like(job["keys"])]
                        gath2 = [job["gathered"], torch.empty_like(job["gathered"])]

                        def pipelined(count):
                            works = [None, None]
                            for it in range(count + 2):
                                b = it & 1
                                if works[b] is not None:
                                    works[b].wait()
                                    job["merge"](gath2[b])
                                    works[b] = None
                                if it < count:
                                    job["match"](keys2[b])
                                    works[b] = dist.all_gather_into_tensor(gath2[b], keys2[b], async_op=True)

                        pipelined(4)
                        torch.cuda.synchronize(dev)
                        host_barrier()
                        torch.cuda.synchronize(dev)
                        p0 = time.perf_counter()
                        pipelined(R)
                        torch.cuda.synchronize(dev)
                        host_barrier()
                        pl = max_over_ranks(time.perf_counter() - p0)
                        res["pipelined_gcmps"] = round(K * NQ * NT * R / pl / 1e9, 2)
                return res, rel

            ks = sorted({1, 4, 16, max(1, args.reloc_sets_per_collective)})
            per_k = {}
            rel = None
            for K in ks:
                per_k[str(K)], rel_k = time_reloc(K)
                if K == 1:
                    rel = rel_k
            hamming["reloc_2000x1M_gcmps"] = per_k["1"]["gcmps"]
            hamming["reloc_phases_ms"] = per_k["1"]["phases_ms"]
            if multi:
                hamming["reloc_pipelined_gcmps"] = {k: v["pipelined_gcmps"] for k, v in per_k.items()}
            # K query sets per collective: at 8 ranks a rank's share of one set is some tens of
            # microseconds of kernel against one latency-bound all-gather; K sets share one
            hamming["reloc_sets_per_collective"] = per_k
            hamming["reloc_sets_per_collective_selected"] = max(1, args.reloc_sets_per_collective)
            dq = torch.from_numpy(synth.random_descriptors(NQ, 3, 5)).to(dev)
            keys = torch.empty((NQ, 2), dtype=torch.int64, device=dev)
            if world == 1 and not multi:
                # the same job on the vector-ALU popcount kernel (v_xor + v_bcnt), against
                # the measured integer-VALU ceiling: 16 lane-ops per comparison
                with FrontEnd(mode=abi.MODE_ORB, nfeatures=NFEAT, max_width=W, max_height=H, max_batch=1,
                              device_id=local_rank, stream=stream.cuda_stream, map_capacity=0,
                              match_kernel=abi.MATCH_POPCOUNT) as fp:
                    with torch.cuda.stream(stream):
                        for _ in range(3):
                            fp.match_shard_keys(dq.data_ptr(), NQ, shard.data_ptr(), hi - lo, lo, keys.data_ptr())
                        torch.cuda.synchronize(dev)
                        p0 = time.perf_counter()
                        for _ in range(10):
                            fp.match_shard_keys(dq.data_ptr(), NQ, shard.data_ptr(), hi - lo, lo, keys.data_ptr())
                        torch.cuda.synchronize(dev)
                        pc = NQ * NT * 10 / (time.perf_counter() - p0) / 1e9
                hamming["reloc_popcount_kernel_gcmps"] = round(pc, 2)
                hamming["popcount_kernel_valu_roofline"] = {
                    "bound": "integer VALU", "lane_ops_per_cmp": 16, "achieved_Tops": round(pc * 16 / 1e3, 2),
                    "peak_Tops": VALU_PEAK_TOPS, "frac": round(pc * 16 / 1e3 / VALU_PEAK_TOPS, 4)}
            hamming["reloc_mfma_roofline"] = match_roofline(NQ * NT * R / rel / 1e9, world)
            hamming["reloc_ms_per_query_set"] = round(rel / R * 1e3, 4)
            hamming["reloc_scaling"] = "strong (map rows split over ranks, RCCL all-gather of top-2 keys)"
            del shard
        except Exception as e:  # the headline number must survive a failure of this leg
            hamming["reloc_error"] = repr(e)

    total_frames = B * S * args.steps * world
    value = total_frames / elapsed
    result = {
        "metric": "frames/sec FAST+ORB extract+match (640x480, 2k feats)",
        "value": round(value, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "640x480 textured frame stream, FAST thr=20, 8-level x1.2 pyramid, "
                               "2000 ORB features, brute-force Hamming k=2 + ratio test per frame",
                   "frames_per_step": B * S, "frame_streams": S, "mode": "ORB (pyramid + quota + orientation)",
                   "keypoints_per_frame_mean": round(float(all_counts.mean()), 1),
                   "input": "gray u8 frames resident in HBM"},
        "roofline": roofline,
        "stage_ms_per_batch": {k: round(v, 5) for k, v in stage_ms.items()},
        "stage_ms_per_256_frames": {k: round(v * 256.0 / B, 5) for k, v in stage_ms.items()},
        "dominant_stage": dominant,
        "hamming": hamming,
    }
    for f in fes:
        f.close()
    del match_outs
    if S == 1 and not args.no_legs and world == 1 and not multi:
        try:
            result["two_streams"] = leg_two_streams(frames, B, dev, local_rank, max(args.steps, 20))
        except Exception as e:  # the headline number must survive a failure of this leg
            result["two_streams"] = {"error": repr(e)}
    del frames
    check = not args.no_cpu

    # The timed batch against the CPU checker (rank 0's frames): keypoints,
    # descriptors and the match rows of the checked frames must be identical.
    if check and rank == 0:
        O = checker()
        p = abi.reference_params(fast_threshold=THR, nfeatures=NFEAT)
        want = {i: O.extract(p, frames_np[i]) for i in need}
        for i in need:
            assert np.array_equal(kept[i].keypoints, want[i][0]), f"bench frame {i}: keypoints differ from the CPU checker"
            assert np.array_equal(kept[i].descriptors, want[i][1]), f"bench frame {i}: descriptors differ"
        for k in check_ix:
            ref = O.match_knn2(want[(k - 1) % B][1], want[k][1])  # query = previous frame, train = frame k
            got = kept_rows[k][:len(ref)]
            for name in MATCH_FIELDS:
                assert np.array_equal(got[name], ref[name]), f"bench frame {k}: match rows differ ({name})"
        result["parity_checked_frames"] = check_ix

    # Legs outside the headline: the host-pointer path and the other BASELINE
    # configs.  Each is fenced: the headline line survives a failure of any of them.
    if not args.no_legs:
        legs = {}
        if world == 1 and not multi:
            for name, fn in (("host_path", lambda: leg_host_path(local_rank)),
                             ("config2_kitti_search_by_projection", lambda: leg_config2(local_rank, check)),
                             ("config5_3840x2160_10k_features",
                              lambda: leg_extract_config("config5", 3840, 2160, 10000, 4, 4, dev, local_rank, 1, 0,
                                                         False, check, False))):
                try:
                    legs[name] = fn()
                except Exception as e:
                    legs[name] = {"error": repr(e)}
        try:
            # config 3: the fixed batch of 256 x 1080p frames split over the ranks
            legs["config3_256x1080p_5k_features"] = leg_extract_config(
                "config3", 1920, 1080, 5000, 256, 32, dev, local_rank, world, rank, multi, check and rank == 0, True)
        except Exception as e:
            legs["config3_256x1080p_5k_features"] = {"error": repr(e)}
        if "host_path" in legs:
            result["host_path"] = legs.pop("host_path")
        result["configs"] = legs

    if rank == 0 and world == 1 and check:
        result["cpu_baseline"] = cpu_baseline(frames_np)
        result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
    if multi:
        host_barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
