"""Host-side mirror of the reference's front-end calls, over the C ABI.

The reference drives its front-end through three OpenCV objects created in
``Init::Init`` (Init.cc:64-66) and four calls (Init.cc:76, 78, 131, 240).  This
module exposes the same operations, with the same argument meaning, on top of
``libslamfe.so``:

================================  =============================================
reference (Init.cc)               here
================================  =============================================
``Init::extractKeyPoints`` :74    :meth:`FrontEnd.extract_keypoints`
``matcher->knnMatch(..., 2)`` :131  :meth:`FrontEnd.knn_match`
ratio loop :144-153               ``ratio_pass`` column of the match rows
search by projection :229-259     :meth:`FrontEnd.search_by_projection`
``new Point(.., desc2.row(idx))``  :meth:`FrontEnd.map_put`
================================  =============================================

Errors: the reference lets ``cv::Exception`` terminate the process; here every
non-zero status of the C ABI raises :class:`SlamfeError` carrying the library's
message.  Nothing in this module computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import abi
from ._lib import load


class SlamfeError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"slamfe error {code}: {message}")
        self.code = code


class Frame:
    """Result of one extraction: host copies plus the device-resident handle."""

    def __init__(self, handle, keypoints, descriptors):
        self.handle = int(handle)
        self.keypoints = keypoints      # structured array, cv::KeyPoint layout
        self.descriptors = descriptors  # [n, 32] uint8 (continuous CV_8U Mat)

    def __len__(self):
        return len(self.keypoints)

    @property
    def pt(self):
        """[n, 2] float32 view of KeyPoint.pt."""
        return np.stack([self.keypoints["x"], self.keypoints["y"]], axis=1)


class FrontEnd:
    """One context = one GPU, one stream; not re-entrant (Init.cc:83 calls the
    front-end from a single thread)."""

    def __init__(self, **params):
        self._lib = load()
        p = abi.Params()
        self._lib.slamfe_default_params(C.byref(p))
        for k, v in params.items():
            if not hasattr(p, k):
                raise AttributeError(f"unknown parameter {k}")
            setattr(p, k, v)
        self.params = p
        ctx = C.c_void_p()
        rc = self._lib.slamfe_create(C.byref(p), C.byref(ctx))
        self._ctx = ctx
        if rc != 0:
            msg = self._lib.slamfe_last_error(ctx).decode() if ctx else "create failed"
            if ctx:
                self._lib.slamfe_destroy(ctx)
            self._ctx = None
            raise SlamfeError(rc, msg or "slamfe_create failed (no usable HIP device?)")
        self.mode = p.mode
        mk = p.max_keypoints
        if mk <= 0:
            mk = max(2 * p.nfeatures, 4096) if p.mode == abi.MODE_ORB else 32768
        self.max_keypoints = mk

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.slamfe_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise SlamfeError(rc, self._lib.slamfe_last_error(self._ctx).decode())

    @property
    def ctx(self):
        return self._ctx

    # -- extraction -------------------------------------------------------
    def extract_keypoints(self, frame: np.ndarray) -> Frame:
        """``Init::extractKeyPoints(frame, keypoints, descriptors)``.  frame is
        [h, w] gray or [h, w, 3] BGR, uint8."""
        return self.extract_batch([frame])[0]

    def extract_batch(self, frames) -> list:
        frames = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
        n = len(frames)
        f0 = frames[0]
        ch = 1 if f0.ndim == 2 else f0.shape[2]
        h, w = f0.shape[:2]
        for f in frames:
            if f.shape != f0.shape:
                raise ValueError("all frames of a batch must have one shape")
        cap = self.max_keypoints
        # the landing buffers are kept between calls: fresh ones would be paged
        # in anew every batch, which costs more than the transfer itself
        held = getattr(self, "_batch_out", None)
        if held is None or held[0].shape[0] < n:
            held = (np.empty((n, cap), abi.KEYPOINT_DTYPE), np.empty((n, cap, 32), np.uint8))
            self._batch_out = held
        kps, desc = held[0][:n], held[1][:n]
        counts = np.zeros(n, np.int32)
        handles = np.zeros(n, np.uint64)
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
        rc = self._lib.slamfe_extract_batch(
            self._ctx, ptrs, n, w, h, w * ch, ch, kps.ctypes.data, desc.ctypes.data, cap,
            counts.ctypes.data_as(C.POINTER(C.c_int32)),
            handles.ctypes.data_as(C.POINTER(C.c_uint64)))
        self._check(rc)
        return [Frame(handles[i], kps[i, :counts[i]].copy(), desc[i, :counts[i]].copy())
                for i in range(n)]

    def extract_scaled_bgr(self, frame: np.ndarray, fx: float = 0.99, fy: float = 0.99):
        """``cv::resize(frame, frame, Size(), fx, fy)`` (Init.cc:103,126) followed by
        ``extractKeyPoints``.  Returns (Frame, resized BGR frame)."""
        frame = np.ascontiguousarray(frame, dtype=np.uint8)
        h, w, ch = frame.shape
        if ch != 3:
            raise ValueError("a BGR frame is expected")
        cap = self.max_keypoints
        kps = np.zeros(cap, abi.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        resized = np.zeros(h * w * 3, np.uint8)
        n, dw, dh = C.c_int32(), C.c_int32(), C.c_int32()
        hd = C.c_uint64()
        self._check(self._lib.slamfe_extract_scaled_bgr(
            self._ctx, frame.ctypes.data, w, h, w * 3, fx, fy, resized.ctypes.data, C.byref(dw),
            C.byref(dh), kps.ctypes.data, desc.ctypes.data, cap, C.byref(n), C.byref(hd)))
        out = resized[:dw.value * dh.value * 3].reshape(dh.value, dw.value, 3).copy()
        return Frame(hd.value, kps[:n.value].copy(), desc[:n.value].copy()), out

    def extract_resident(self, d_ptr: int, n: int, w: int, h: int, stride: int,
                         frame_stride: int, channels: int = 1):
        """Asynchronous extraction of frames already in device memory."""
        handles = np.zeros(n, np.uint64)
        self._check(self._lib.slamfe_extract_resident(
            self._ctx, C.c_void_p(d_ptr), n, w, h, stride, frame_stride, channels,
            handles.ctypes.data_as(C.POINTER(C.c_uint64))))
        return handles

    def frame_download(self, handle) -> Frame:
        cap = self.max_keypoints
        kps = np.zeros(cap, abi.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int32()
        self._check(self._lib.slamfe_frame_download(self._ctx, C.c_uint64(int(handle)),
                                                    kps.ctypes.data, desc.ctypes.data, cap,
                                                    C.byref(n)))
        return Frame(handle, kps[:n.value].copy(), desc[:n.value].copy())

    # -- matching ---------------------------------------------------------
    def knn_match(self, query: np.ndarray, train: np.ndarray) -> np.ndarray:
        """``matcher->knnMatch(desc1, desc2, matches, 2)`` + ratio flag; rows of
        MATCH2_DTYPE, one per query."""
        query = np.ascontiguousarray(query, np.uint8).reshape(-1, 32)
        train = np.ascontiguousarray(train, np.uint8).reshape(-1, 32)
        out = np.zeros(max(len(query), 1), abi.MATCH2_DTYPE)
        self._check(self._lib.slamfe_match_knn2(self._ctx, query.ctypes.data, len(query),
                                                train.ctypes.data, len(train), out.ctypes.data))
        return out[:len(query)]

    def knn_match_frames(self, query: Frame, train: Frame) -> np.ndarray:
        cap = self.max_keypoints
        out = np.zeros(cap, abi.MATCH2_DTYPE)
        n = C.c_int32()
        self._check(self._lib.slamfe_match_frames(self._ctx, C.c_uint64(int(query.handle)),
                                                  C.c_uint64(int(train.handle)), out.ctypes.data,
                                                  cap, C.byref(n)))
        return out[:n.value].copy()

    def match_resident(self, query_handle, train_handle, d_out: int):
        self._check(self._lib.slamfe_match_resident(self._ctx, C.c_uint64(int(query_handle)),
                                                    C.c_uint64(int(train_handle)),
                                                    C.c_void_p(d_out)))

    def match_resident_batch(self, first_query, first_train, npairs: int, d_out: int):
        self._check(self._lib.slamfe_match_resident_batch(
            self._ctx, C.c_uint64(int(first_query)), C.c_uint64(int(first_train)), npairs,
            C.c_void_p(d_out)))

    def time_stage(self, stage: int, nframes: int, reps: int) -> float:
        """Measurement hook: average milliseconds of one stage, HIP events on
        the context's stream."""
        ms = C.c_float()
        self._check(self._lib.slamfe_debug_time_stage(self._ctx, stage, nframes, reps,
                                                      C.byref(ms)))
        return ms.value

    def probe_stage(self, stage: int):
        """Measurement hook: record HIP events around `stage` of the calls that
        follow (0 switches it off)."""
        self._check(self._lib.slamfe_debug_probe(self._ctx, stage))

    def probe_read(self):
        """(mean milliseconds, number of intervals) recorded since probe_stage."""
        ms, n = C.c_float(), C.c_int()
        self._check(self._lib.slamfe_debug_probe_read(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def match_shard_keys(self, d_query: int, nq: int, d_train: int, nt: int, train_base: int,
                         d_keys: int):
        self._check(self._lib.slamfe_match_shard_keys(self._ctx, C.c_void_p(d_query), nq,
                                                      C.c_void_p(d_train), nt,
                                                      C.c_int64(train_base), C.c_void_p(d_keys)))

    def match_shard_keys_static(self, d_query: int, nq: int, d_train: int, nt: int, train_base: int,
                                d_keys: int):
        """``match_shard_keys`` without any device allocation (fixed scratch array)."""
        self._check(self._lib.slamfe_match_shard_keys_static(self._ctx, C.c_void_p(d_query), nq,
                                                             C.c_void_p(d_train), nt,
                                                             C.c_int64(train_base), C.c_void_p(d_keys)))

    def merge_shard_keys(self, d_keys_all: int, nshards: int, nq: int, d_out: int):
        self._check(self._lib.slamfe_merge_shard_keys(self._ctx, C.c_void_p(d_keys_all), nshards, nq,
                                                      C.c_void_p(d_out)))

    # -- map --------------------------------------------------------------
    def map_put(self, point_ids, desc32):
        ids = np.ascontiguousarray(point_ids, np.int32)
        d = np.ascontiguousarray(desc32, np.uint8).reshape(-1, 32)
        if len(ids) != len(d):
            raise ValueError("one descriptor row per point id")
        self._check(self._lib.slamfe_map_put(self._ctx, ids.ctypes.data, d.ctypes.data, len(ids)))

    def search_by_projection(self, frame: Frame, kp_idx, uv, point_ids, radius: float = 32.0,
                             max_hamming: int = 128) -> np.ndarray:
        """The search of Init.cc:229-259: for inlier i (``kp_idx[i]`` =
        ``goodMatches[i].trainIdx``) the id of the matched map point or -1."""
        kp_idx = np.ascontiguousarray(kp_idx, np.int32)
        uv = np.ascontiguousarray(uv, np.float32).reshape(-1, 2)
        ids = np.ascontiguousarray(point_ids, np.int32)
        if len(uv) != len(ids):
            raise ValueError("one id per projected point")
        out = np.full(max(len(kp_idx), 1), -2, np.int32)
        self._check(self._lib.slamfe_search_by_projection(
            self._ctx, C.c_uint64(int(frame.handle)), kp_idx.ctypes.data, len(kp_idx),
            uv.ctypes.data, ids.ctypes.data, len(ids), C.c_float(radius), max_hamming,
            out.ctypes.data))
        return out[:len(kp_idx)]

    def map_put_xyz(self, point_ids, xyz):
        """Mirror of ``Point::getCoords`` (keyframe.cc:6, 15-25) by point id."""
        ids = np.ascontiguousarray(point_ids, np.int32)
        x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        if len(ids) != len(x):
            raise ValueError("one coordinate triple per point id")
        self._check(self._lib.slamfe_map_put_xyz(self._ctx, ids.ctypes.data, x.ctypes.data, len(ids)))

    def reproject(self, pose, K, th: float, W: float, H: float, npoints: int, download: bool = True):
        """The reprojection loop of Init.cc:203-218 over map points 0..npoints-1.
        Returns (uv [R,2] f32, point index [R] i32) in ascending order, or just R
        when ``download`` is false; the visible set stays on the device."""
        pose = np.ascontiguousarray(pose, np.float32).reshape(4, 4)
        K = np.ascontiguousarray(K, np.float32).reshape(3, 3)
        n = C.c_int32()
        if not download:
            self._check(self._lib.slamfe_reproject(self._ctx, pose.ctypes.data, K.ctypes.data,
                                                   C.c_float(th), C.c_float(W), C.c_float(H), npoints,
                                                   None, None, 0, C.byref(n)))
            return n.value
        uv = np.zeros((max(npoints, 1), 2), np.float32)
        idx = np.zeros(max(npoints, 1), np.int32)
        self._check(self._lib.slamfe_reproject(self._ctx, pose.ctypes.data, K.ctypes.data,
                                               C.c_float(th), C.c_float(W), C.c_float(H), npoints,
                                               uv.ctypes.data, idx.ctypes.data, npoints, C.byref(n)))
        return uv[:n.value].copy(), idx[:n.value].copy()

    def search_by_projection_visible(self, frame: Frame, kp_idx, radius: float = 32.0,
                                     max_hamming: int = 128) -> np.ndarray:
        """Init.cc:229-259 over the visible set left on the device by ``reproject``."""
        kp_idx = np.ascontiguousarray(kp_idx, np.int32)
        out = np.full(max(len(kp_idx), 1), -2, np.int32)
        self._check(self._lib.slamfe_search_by_projection_visible(
            self._ctx, C.c_uint64(int(frame.handle)), kp_idx.ctypes.data, len(kp_idx),
            C.c_float(radius), max_hamming, out.ctypes.data))
        return out[:len(kp_idx)]

    def triangulate(self, P1, P2, pts1, pts2) -> np.ndarray:
        """``Init::triangulate`` (Init.cc:5-31) for point pairs: [n, 3] float32."""
        P1 = np.ascontiguousarray(P1, np.float32).reshape(3, 4)
        P2 = np.ascontiguousarray(P2, np.float32).reshape(3, 4)
        a = np.ascontiguousarray(pts1, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(pts2, np.float32).reshape(-1, 2)
        if len(a) != len(b):
            raise ValueError("one point of each view per pair")
        out = np.zeros((max(len(a), 1), 3), np.float32)
        self._check(self._lib.slamfe_triangulate(self._ctx, P1.ctypes.data, P2.ctypes.data, a.ctypes.data,
                                                 b.ctypes.data, len(a), out.ctypes.data))
        return out[:len(a)]

    def debug_read_level(self, frame_in_batch: int, level: int, blurred: bool) -> np.ndarray:
        """Test hook: a pyramid level of the most recent extraction."""
        cap = self.params.max_width * self.params.max_height
        buf = np.zeros(cap, np.uint8)
        w = C.c_int32()
        h = C.c_int32()
        self._check(self._lib.slamfe_debug_read_level(self._ctx, frame_in_batch, level,
                                                      int(blurred), buf.ctypes.data, cap,
                                                      C.byref(w), C.byref(h)))
        return buf[:w.value * h.value].reshape(h.value, w.value).copy()

    def sync(self):
        self._check(self._lib.slamfe_sync(self._ctx))
