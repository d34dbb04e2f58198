"""GPU parity: the HIP path, called through the C ABI, against the CPU checker
on the same seeded inputs.  Bit-exact everywhere (integer / byte / index work;
the two float routines are shared source with -ffp-contract=off).
"""
import numpy as np
import pytest

import oracle_lib as O
from slam_amd import abi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe_orb():
    from slam_amd.frontend import FrontEnd
    fe = FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=1280,
                  max_height=720, max_batch=4, map_capacity=1 << 15)
    yield fe
    fe.close()


@pytest.fixture(scope="module")
def fe_ref():
    from slam_amd.frontend import FrontEnd
    fe = FrontEnd(mode=abi.MODE_PROVIDED, fast_threshold=30, max_width=1280, max_height=720,
                  max_batch=2)
    yield fe
    fe.close()


def oracle_params(fe):
    p = abi.Params()
    for name, _ in abi.Params._fields_:
        setattr(p, name, getattr(fe.params, name))
    return p


def assert_frame_equal(frame, kps, desc):
    assert len(frame) == len(kps), (len(frame), len(kps))
    for name in abi.KEYPOINT_DTYPE.names:
        bad = np.nonzero(frame.keypoints[name] != kps[name])[0]
        assert len(bad) == 0, (name, bad[:5], frame.keypoints[bad[:5]], kps[bad[:5]])
    bad = np.nonzero((frame.descriptors != desc).any(axis=1))[0]
    assert len(bad) == 0, (bad[:5], len(bad))


# ------------------------------------------------------------ stage parity
def test_pyramid_and_blur_levels(fe_orb):
    img, _ = synth.frame_pair(640, 480)
    fe_orb.extract_keypoints(img)
    levels = O.build_pyramid(img)
    for l, want in enumerate(levels):
        got = fe_orb.debug_read_level(0, l, False)
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"pyramid level {l}"
        assert np.array_equal(fe_orb.debug_read_level(0, l, True), O.blur7(want)), f"blur {l}"


@pytest.mark.parametrize("shape", [(333, 277), (200, 160), (97, 131)])
def test_pyramid_ragged_sizes(fe_orb, shape):
    w, h = shape
    img = synth.textured_canvas(w, h, 41)
    fe_orb.extract_keypoints(img)
    for l, want in enumerate(O.build_pyramid(img)):
        assert np.array_equal(fe_orb.debug_read_level(0, l, False), want)
        assert np.array_equal(fe_orb.debug_read_level(0, l, True), O.blur7(want))


@pytest.mark.parametrize("kind", ["white", "black", "binary", "checker", "noise", "steps"])
def test_blur_extreme_images(fe_orb, kind):
    # the ends of the blur's integer range: the tap sum is 257, so an all-white
    # neighbourhood gives (257 * 257 * 255 + 2^15) >> 16 = 257 and must saturate to 255;
    # black / white patterns drive every row sum and its 8-bit halves (the blur splits a
    # row sum into a high and a low byte for the matrix cores) to their extremes
    w, h = 335, 251  # ragged: partial tiles on both edges, a last tile group of few rows
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w]
    img = {
        "white": np.full((h, w), 255, np.uint8),
        "black": np.zeros((h, w), np.uint8),
        "binary": (rng.integers(0, 2, (h, w)) * 255).astype(np.uint8),
        "checker": (((xx // 3 + yy // 5) & 1) * 255).astype(np.uint8),
        "noise": rng.integers(0, 256, (h, w), dtype=np.uint8),
        "steps": np.where(xx < w // 2, 254, 255).astype(np.uint8) - (yy % 2).astype(np.uint8),
    }[kind]
    from slam_amd.frontend import SlamfeError
    try:
        fe_orb.extract_keypoints(img)
    except SlamfeError as e:  # thousands of tied corners on the patterns: the levels are made all the same
        assert e.code == abi.SLAMFE_E_CAPACITY
    for l, want in enumerate(O.build_pyramid(img)):
        assert np.array_equal(fe_orb.debug_read_level(0, l, False), want), (kind, l)
        assert np.array_equal(fe_orb.debug_read_level(0, l, True), O.blur7(want)), (kind, l)


# --------------------------------------------------------------- extraction
def test_extract_orb_640x480_pair(fe_orb):
    a, b = synth.frame_pair(640, 480)
    p = oracle_params(fe_orb)
    for img in (a, b):
        f = fe_orb.extract_keypoints(img)
        kps, desc = O.extract(p, img)
        assert len(kps) >= 2000
        assert_frame_equal(f, kps, desc)
        ang = np.radians(f.keypoints["angle"].astype(np.float64) - kps["angle"])
        assert np.abs(ang).max() <= 1e-3  # the stated orientation tolerance (here: exact)


@pytest.mark.parametrize("shape,seed", [((333, 277), 42), ((200, 160), 43), ((1241, 376), 44),
                                        ((1280, 720), 45), ((70, 70), 46), ((64, 64), 47)])
def test_extract_orb_other_sizes(fe_orb, shape, seed):
    w, h = shape
    img = synth.textured_canvas(w, h, seed)
    f = fe_orb.extract_keypoints(img)
    kps, desc = O.extract(oracle_params(fe_orb), img)
    assert_frame_equal(f, kps, desc)


def test_extract_flat_image_has_no_keypoints(fe_orb, fe_ref):
    img = np.full((240, 320), 99, np.uint8)
    assert len(fe_orb.extract_keypoints(img)) == 0
    assert len(fe_ref.extract_keypoints(img)) == 0


def test_extract_reference_mode(fe_ref):
    p = oracle_params(fe_ref)
    for (w, h), seed in (((640, 480), 1), ((1241, 376), 2), ((333, 277), 3)):
        img = synth.textured_canvas(w, h, seed)
        f = fe_ref.extract_keypoints(img)
        kps, desc = O.extract(p, img)
        assert len(kps) > 100
        assert_frame_equal(f, kps, desc)
        assert np.all(f.keypoints["angle"] == -1) and np.all(f.keypoints["size"] == 7)
        key = f.keypoints["y"].astype(np.int64) * 10000 + f.keypoints["x"].astype(np.int64)
        assert np.all(np.diff(key) > 0)  # FAST's row-major order


@pytest.mark.parametrize("thr", [1, 20, 127, 254])
def test_fast_scores_at_extreme_contrast(thr):
    # The score network runs on pixel values taken as 16-bit float bit patterns, two
    # survivors per lane: isolated 0 / 255 dots and small blobs on grounds of 0, 1, 127,
    # 254 and 255 give margins from 1 to 255 (scores 0 .. 254), ties between neighbours
    # in the non-maximum suppression, and survivor counts of either parity in a tile
    # group.  FAST-only mode returns the raw responses in row-major order.
    from slam_amd.frontend import FrontEnd
    w, h = 397, 283
    rng = np.random.default_rng(100 + thr)
    img = np.zeros((h, w), np.uint8)
    for x0, ground in ((0, 0), (80, 1), (160, 127), (240, 254), (320, 255)):
        img[:, x0:x0 + 80] = ground
    for _ in range(900):  # dots
        img[rng.integers(0, h), rng.integers(0, w)] = rng.choice([0, 1, 126, 128, 254, 255])
    for _ in range(120):  # blobs of 2 x 2 .. 4 x 4 with one value
        y, x, n = rng.integers(0, h - 4), rng.integers(0, w - 4), rng.integers(2, 5)
        img[y:y + n, x:x + n] = rng.choice([0, 255, 64, 192])
    with FrontEnd(mode=abi.MODE_PROVIDED, fast_threshold=thr, max_width=w, max_height=h) as fe:
        f = fe.extract_keypoints(img)
        kps, desc = O.extract(oracle_params(fe), img)
        assert_frame_equal(f, kps, desc)
        assert len(kps) > 20 and kps["response"].max() == 254


def test_extract_bgr_input(fe_orb, fe_ref):
    g = synth.textured_canvas(320, 240, 48)
    bgr = np.stack([np.roll(g, 1, 0), g, np.roll(g, 2, 1)], axis=2)
    for fe in (fe_orb, fe_ref):
        f = fe.extract_keypoints(bgr)
        kps, desc = O.extract(oracle_params(fe), bgr)
        assert len(kps) > 20
        assert_frame_equal(f, kps, desc)


@pytest.mark.parametrize("shape,seed", [((640, 480), 61), ((1241, 376), 62), ((333, 277), 63),
                                        ((1280, 720), 64)])
def test_extract_scaled_bgr_frame(fe_orb, fe_ref, shape, seed):
    # Init.cc:103/126 resize(0.99) + Init.cc:74-79, against checker resize -> checker extract
    w, h = shape
    g = synth.textured_canvas(w, h, seed)
    bgr = np.ascontiguousarray(np.stack([np.roll(g, 1, 0), g, np.roll(g, 2, 1)], axis=2))
    want_frame = O.resize_linear_8u(bgr, 0.99, 0.99)
    for fe in (fe_orb, fe_ref):
        f, resized = fe.extract_scaled_bgr(bgr, 0.99, 0.99)
        assert resized.shape == want_frame.shape
        assert np.array_equal(resized, want_frame)
        assert np.array_equal(fe.debug_read_level(0, 0, False), O.bgr2gray(want_frame))
        kps, desc = O.extract(oracle_params(fe), want_frame)
        assert len(kps) > 50
        assert_frame_equal(f, kps, desc)


def test_extract_scaled_bgr_other_factors_and_errors(fe_orb):
    from slam_amd.frontend import SlamfeError
    g = synth.textured_canvas(320, 240, 65)
    bgr = np.ascontiguousarray(np.stack([g, np.roll(g, 3, 1), np.roll(g, 1, 0)], axis=2))
    for fx, fy in ((1.0, 1.0), (0.75, 0.9), (0.6, 0.6)):
        f, resized = fe_orb.extract_scaled_bgr(bgr, fx, fy)
        want = O.resize_linear_8u(bgr, fx, fy)
        assert np.array_equal(resized, want)
        kps, desc = O.extract(oracle_params(fe_orb), want)
        assert_frame_equal(f, kps, desc)
    with pytest.raises(SlamfeError):
        fe_orb.extract_scaled_bgr(bgr, 0.5, 0.5)   # exact halving is OpenCV's area path: not offered
    with pytest.raises(SlamfeError):
        fe_orb.extract_scaled_bgr(bgr, 1.5, 1.0)


def test_extract_batch_equals_single_frames(fe_orb):
    frames = synth.frame_stream(640, 480, 4, seed=5)
    batch = fe_orb.extract_batch(list(frames))
    p = oracle_params(fe_orb)
    for img, f in zip(frames, batch):
        kps, desc = O.extract(p, img)
        assert_frame_equal(f, kps, desc)
    again = fe_orb.frame_download(batch[2].handle)
    assert_frame_equal(again, batch[2].keypoints, batch[2].descriptors)


def test_extract_capacity_error():
    from slam_amd.frontend import FrontEnd, SlamfeError
    img, _ = synth.frame_pair(640, 480)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_keypoints=500,
                  max_width=640, max_height=480) as fe:
        with pytest.raises(SlamfeError) as e:
            fe.extract_keypoints(img)
        assert e.value.code == abi.SLAMFE_E_CAPACITY


def test_frame_handle_expires(fe_orb):
    from slam_amd.frontend import SlamfeError
    img = synth.textured_canvas(200, 160, 49)
    first = fe_orb.extract_keypoints(img)
    for _ in range(fe_orb.params.frame_slots if fe_orb.params.frame_slots else 8):
        fe_orb.extract_keypoints(img)
    with pytest.raises(SlamfeError) as e:
        fe_orb.frame_download(first.handle)
    assert e.value.code == abi.SLAMFE_E_HANDLE


@pytest.mark.parametrize("nframes", [8, 11, 19])
def test_extract_and_match_large_batches(nframes):
    # eight frames and more take the grid that keeps a frame's workgroups on one
    # XCD (frame = blockIdx.z * 8 + blockIdx.x); 11 and 19 leave the last group ragged
    import torch
    from slam_amd.frontend import FrontEnd
    frames = synth.frame_stream(320, 240, nframes, seed=21 + nframes)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=500, max_width=320, max_height=240,
                  max_batch=nframes, frame_slots=2 * nframes, map_capacity=0) as fe:
        got = fe.extract_batch(list(frames))
        p = oracle_params(fe)
        want = [O.extract(p, fr) for fr in frames]
        for f, (kps, desc) in zip(got, want):
            assert len(kps) > 300
            assert_frame_equal(f, kps, desc)
        cap = fe.max_keypoints
        out = torch.zeros((nframes - 1, cap, 16), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the fill runs on torch's stream, the library on its own
        fe.match_resident_batch(got[1].handle, got[0].handle, nframes - 1, out.data_ptr())
        fe.sync()
        rows = out.cpu().numpy()
        for i in range(nframes - 1):
            m = np.frombuffer(rows[i].tobytes(), dtype=abi.MATCH2_DTYPE)[:len(got[i + 1])]
            ref = O.match_knn2(want[i + 1][1], want[i][1])
            for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
                assert np.array_equal(m[name], ref[name]), (i, name)


@pytest.mark.parametrize("channels", [1, 3])
def test_extract_resident_and_padded_rows(channels):
    # frames already in device memory with padded rows and a gap between frames,
    # and host frames with padded rows through the C ABI: the row stride is honoured
    import ctypes as C
    import torch
    from slam_amd.frontend import FrontEnd
    n, w, h = 9, 333, 201
    gray = synth.frame_stream(w, h, n, seed=77)
    frames = gray if channels == 1 else np.stack([gray, np.roll(gray, 2, 2), np.roll(gray, 1, 1)], axis=3)
    row = w * channels
    stride, fstride = row + 37, (row + 37) * h + 1000
    buf = np.full(n * fstride, 0xAB, np.uint8)
    for i in range(n):
        view = buf[i * fstride:i * fstride + stride * h].reshape(h, stride)
        view[:, :row] = frames[i].reshape(h, row)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=400, max_width=w, max_height=h,
                  max_batch=n, frame_slots=2 * n, map_capacity=0) as fe:
        p = oracle_params(fe)
        want = [O.extract(p, np.ascontiguousarray(fr)) for fr in frames]
        d = torch.from_numpy(buf).cuda()
        hd = fe.extract_resident(d.data_ptr(), n, w, h, stride, fstride, channels)
        for i in range(n):
            assert_frame_equal(fe.frame_download(int(hd[i])), *want[i])
        # host frame with padded rows
        cap = fe.max_keypoints
        kps = np.zeros(cap, abi.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        cnt, handle = C.c_int32(), C.c_uint64()
        rc = fe._lib.slamfe_extract(fe._ctx, buf[3 * fstride:].ctypes.data, w, h, stride, channels,
                                    kps.ctypes.data, desc.ctypes.data, cap, C.byref(cnt), C.byref(handle))
        assert rc == 0
        assert np.array_equal(kps[:cnt.value], want[3][0]) and np.array_equal(desc[:cnt.value], want[3][1])


def test_extract_fuzz_parameters():
    # seeded random frame sizes and ORB parameters (levels, scale factor, threshold,
    # quota, border), both modes, against the checker
    from slam_amd.frontend import FrontEnd
    rng = np.random.default_rng(2024)
    checked = 0
    for case in range(24):
        w, h = int(rng.integers(72, 700)), int(rng.integers(72, 520))
        kw = dict(mode=int(case % 3 == 2), fast_threshold=int(rng.integers(5, 60)),
                  nfeatures=int(rng.integers(40, 3000)), nlevels=int(rng.integers(1, 9)),
                  scale_factor=float(rng.choice([1.1, 1.2, 1.25, 1.5, 1.7])),
                  edge_threshold=int(rng.integers(22, 40)))
        img = synth.textured_canvas(w, h, 100 + case)
        if case % 4 == 1:  # a BGR frame
            img = np.ascontiguousarray(np.stack([img, np.roll(img, 1, 1), np.roll(img, 1, 0)], axis=2))
        with FrontEnd(max_width=w, max_height=h, map_capacity=0, **kw) as fe:
            f = fe.extract_keypoints(img)
            kps, desc = O.extract(oracle_params(fe), img)
            assert len(f) == len(kps), (case, w, h, kw, len(f), len(kps))
            assert_frame_equal(f, kps, desc)
            checked += len(kps)
    assert checked > 5000


@pytest.mark.parametrize("nlevels,scale", [(1, 1.2), (2, 1.2), (8, 1.1), (8, 2.0)])
@pytest.mark.parametrize("nframes", [1, 3])
def test_extract_frames_usable_in_place(nlevels, scale, nframes):
    # gray frames whose width is a multiple of 64 already have level 0's layout and are
    # read where they lie (no copy into the pyramid buffer): every kernel that reads
    # level 0 must take it from there, in calls of a few frames too, with one level
    # (no pyramid kernel at all) and with geometries the one-launch pyramid cannot hold
    import torch
    from slam_amd.frontend import FrontEnd
    w, h = 320, 240
    frames = synth.frame_stream(w, h, nframes, seed=40 + nlevels)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=500, nlevels=nlevels, scale_factor=scale,
                  max_width=w, max_height=h, max_batch=4, map_capacity=0) as fe:
        want = [O.extract(oracle_params(fe), fr) for fr in frames]
        for f, (kps, desc) in zip(fe.extract_batch(list(frames)), want):  # host frames
            assert_frame_equal(f, kps, desc)
        d = torch.from_numpy(np.ascontiguousarray(frames)).cuda()        # resident frames, in place
        torch.cuda.synchronize()
        handles = fe.extract_resident(d.data_ptr(), nframes, w, h, w, w * h, 1)
        for hd, (kps, desc) in zip(handles, want):
            assert_frame_equal(fe.frame_download(int(hd)), kps, desc)
        assert sum(len(k) for k, _ in want) > 100


def test_extract_in_place_from_a_buffer_with_nothing_behind_it():
    # frames read in place from a device buffer of exactly n * w * h bytes taken straight
    # from hipMalloc (no allocator padding behind the last row; 640 x 480 frames are whole
    # 4 KiB pages): no kernel may read past the last byte of the last frame.  The level-1
    # resize loads 12-byte windows and used to run 8 bytes over the end of every row.
    import ctypes as C
    from slam_amd.frontend import FrontEnd
    import torch  # noqa: F401  (the process's HIP runtime is the one torch brought in)
    with open("/proc/self/maps") as maps:
        hip_path = next(line.split()[-1] for line in maps if "libamdhip64" in line)
    hip = C.CDLL(hip_path)  # the runtime already loaded, not a second copy
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    w, h, n = 640, 480, 9
    frames = np.ascontiguousarray(synth.frame_stream(w, h, n, seed=77))
    assert frames.nbytes == n * w * h and frames.nbytes % 4096 == 0
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=1000, max_width=w, max_height=h,
                  max_batch=n, map_capacity=0) as fe:
        want = [O.extract(oracle_params(fe), fr) for fr in frames]
        d = C.c_void_p()
        assert hip.hipMalloc(C.byref(d), frames.nbytes) == 0
        try:
            assert hip.hipMemcpy(d, frames.ctypes.data, frames.nbytes, 1) == 0  # host to device
            handles = fe.extract_resident(d.value, n, w, h, w, w * h, 1)
            fe.sync()
            for hd, (kps, desc) in zip(handles, want):
                assert_frame_equal(fe.frame_download(int(hd)), kps, desc)
        finally:
            fe.sync()
            hip.hipFree(d)


@pytest.mark.parametrize("kernel", [abi.MATCH_POPCOUNT, abi.MATCH_MATRIX])
def test_match_extreme_descriptors_among_random_ones(kernel):
    # all-zero and all-one rows among random ones: distances 0 and 256 and queries of
    # popcount 0 and 256, the ends of the range the matrix-core keys have to hold
    # (key = 16 * (distance + 256 - popcount(query)) + row fraction); many exact ties
    from slam_amd.frontend import FrontEnd
    rng = np.random.default_rng(11)
    nq, nt = 300, 2100
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    t = rng.integers(0, 256, (nt, 32), dtype=np.uint8)
    q[::7] = 0
    q[3::7] = 255
    t[5::11] = 0
    t[6::11] = 255
    t[40:45] = q[10]        # exact copies: distance 0, ties broken by the lower row
    t[2050:2052] = q[10]
    with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0,
                  match_kernel=kernel) as fe:
        for qq, tt in ((q, t), (q, t[:33]), (q[:129], np.full((70, 32), 255, np.uint8)),
                       (np.zeros((130, 32), np.uint8), np.full((64, 32), 255, np.uint8))):
            got = fe.knn_match(qq, tt)
            want = O.match_knn2(qq, tt)
            for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
                assert np.array_equal(got[name], want[name]), (kernel, name, len(qq), len(tt))
        assert want["d0"].max() == 256


# ------------------------------------------------------------------ matching
@pytest.mark.parametrize("nq,nt", [(2000, 2000), (1, 1), (1, 2), (257, 63), (64, 1000), (5, 0),
                                   (3001, 2999), (100, 20000),
                                   # 128 queries and more take the matrix-core kernel: empty and
                                   # one-row train sets, ragged tiles (rows % 32), chunk borders
                                   (300, 0), (129, 1), (128, 2), (256, 33), (255, 1025), (513, 2049),
                                   (130, 40000)])
def test_match_knn2(fe_orb, nq, nt):
    q, t = synth.descriptor_sets(nq, max(nt, 1), seed=3 + nq, dup_percent=3)
    t = t[:nt]
    got = fe_orb.knn_match(q, t)
    want = O.match_knn2(q, t)
    for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
        assert np.array_equal(got[name], want[name]), name
    if nt >= 1000:
        assert (got["d0"] == got["d1"]).sum() > 0  # ties were exercised


def test_match_extreme_descriptors(fe_orb):
    # all-zero, all-one and complementary rows: distances 0 and 256, every tie broken by index
    q = np.zeros((200, 32), np.uint8)
    q[1::2] = 0xFF
    t = np.zeros((100, 32), np.uint8)
    t[50:] = 0xFF
    t[10, 0] = 1
    got = fe_orb.knn_match(q, t)
    want = O.match_knn2(q, t)
    for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
        assert np.array_equal(got[name], want[name]), name
    assert got["d0"].min() == 0 and got["idx0"][0] == 0 and got["idx1"][0] == 1
    far = fe_orb.knn_match(q[:150], t[50:52] ^ 0)   # even queries are 256 away from both rows
    assert far["d0"][0] == 256 and far["d1"][0] == 256 and far["idx0"][0] == 0


@pytest.mark.parametrize("kernel", [abi.MATCH_POPCOUNT, abi.MATCH_MATRIX])
def test_match_either_kernel_forced(kernel):
    # the vector-ALU popcount kernel and the matrix-core kernel, each forced for
    # query counts on both sides of the automatic switch
    from slam_amd.frontend import FrontEnd
    with FrontEnd(mode=abi.MODE_ORB, nfeatures=500, max_width=320, max_height=240, map_capacity=0,
                  match_kernel=kernel) as fe:
        for nq, nt in ((40, 3000), (2000, 2000), (700, 1), (257, 4099)):
            q, t = synth.descriptor_sets(nq, nt, seed=11 + nq, dup_percent=3)
            got = fe.knn_match(q, t)
            want = O.match_knn2(q, t)
            for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
                assert np.array_equal(got[name], want[name]), (kernel, nq, nt, name)


def test_match_resident_frames_with_few_or_no_keypoints():
    # the plan is made for the slot capacity, the kernels work from the counts on the
    # device: frames with a handful of keypoints, and a flat frame with none
    from slam_amd.frontend import FrontEnd
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=40, max_width=320, max_height=240,
                  max_batch=3, map_capacity=0) as fe:
        a = synth.textured_canvas(320, 240, 91)
        b = synth.textured_canvas(320, 240, 92)
        flat = np.full((240, 320), 100, np.uint8)
        fa, fb, ff = fe.extract_batch([a, b, flat])
        assert 0 < len(fa) < 128 and 0 < len(fb) < 128 and len(ff) == 0
        for q, t in ((fa, fb), (fb, fa), (fa, ff), (ff, fa), (fa, fa)):
            got = fe.knn_match_frames(q, t)
            want = O.match_knn2(q.descriptors, t.descriptors)
            assert len(got) == len(want)
            for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
                assert np.array_equal(got[name], want[name]), name


def test_match_empty_query(fe_orb):
    assert len(fe_orb.knn_match(np.zeros((0, 32), np.uint8), np.zeros((4, 32), np.uint8))) == 0


def test_match_resident_frames(fe_orb):
    a, b = synth.frame_pair(640, 480)
    fa, fb = fe_orb.extract_batch([a, b])
    got = fe_orb.knn_match_frames(fa, fb)
    want = O.match_knn2(fa.descriptors, fb.descriptors)
    assert len(got) == len(fa)
    for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
        assert np.array_equal(got[name], want[name]), name
    good = got[got["ratio_pass"] == 1]
    assert len(good) > 300
    dx = fb.keypoints["x"][good["idx0"]] - fa.keypoints["x"][got["ratio_pass"] == 1]
    assert abs(np.median(dx) - 7) < 1.0  # the known (+7, +3) motion


def test_match_sharded_keys_merge_equals_single(fe_orb):
    import torch
    nq, nt, shards = 300, 5000, 4
    q, t = synth.descriptor_sets(nq, nt, seed=9, dup_percent=2)
    dq = torch.from_numpy(q).cuda()
    dt = torch.from_numpy(t).cuda()
    keys = torch.empty((shards, nq, 2), dtype=torch.int64, device="cuda")
    out = torch.zeros(nq * 16, dtype=torch.uint8, device="cuda")
    bounds = [0, 1000, 1001, 3500, nt]  # uneven shards, one with a single row
    torch.cuda.synchronize()  # the fill runs on torch's stream, the library on its own
    for s in range(shards):
        lo, hi = bounds[s], bounds[s + 1]
        fe_orb.match_shard_keys(dq.data_ptr(), nq, dt[lo:hi].data_ptr(), hi - lo, lo,
                                keys[s].data_ptr())
    fe_orb.merge_shard_keys(keys.data_ptr(), shards, nq, out.data_ptr())
    fe_orb.sync()
    got = out.cpu().numpy().view(abi.MATCH2_DTYPE)
    want = O.match_knn2(q, t)
    for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
        assert np.array_equal(got[name], want[name]), name


def test_stacked_query_sets_share_one_match_and_one_merge(fe_orb):
    # K query sets per collective in the sharded relocalisation: one match call over the
    # K stacked sets per shard, one merge call over [shard][K * nq][2] keys -- the rows
    # must be those of K separate match + merge sequences, and of the CPU checker
    import torch
    K, nq, nt, shards = 5, 211, 9000, 3
    q, t = synth.descriptor_sets(K * nq, nt, seed=19, dup_percent=2)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    bounds = [0, 4097, 4100, nt]
    stacked = torch.empty((shards, K * nq, 2), dtype=torch.int64, device="cuda")
    out_stacked = torch.zeros(K * nq * 16, dtype=torch.uint8, device="cuda")
    single = torch.empty((shards, nq, 2), dtype=torch.int64, device="cuda")
    out_single = torch.zeros(nq * 16, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for s in range(shards):
        lo, hi = bounds[s], bounds[s + 1]
        fe_orb.match_shard_keys(dq.data_ptr(), K * nq, dt[lo:hi].data_ptr(), hi - lo, lo, stacked[s].data_ptr())
    fe_orb.merge_shard_keys(stacked.data_ptr(), shards, K * nq, out_stacked.data_ptr())
    fe_orb.sync()
    got = out_stacked.cpu().numpy().view(abi.MATCH2_DTYPE)
    want = O.match_knn2(q, t)
    for k in range(K):
        for s in range(shards):
            lo, hi = bounds[s], bounds[s + 1]
            fe_orb.match_shard_keys(dq[k * nq:(k + 1) * nq].data_ptr(), nq, dt[lo:hi].data_ptr(), hi - lo, lo,
                                    single[s].data_ptr())
        fe_orb.merge_shard_keys(single.data_ptr(), shards, nq, out_single.data_ptr())
        fe_orb.sync()
        one = out_single.cpu().numpy().view(abi.MATCH2_DTYPE)
        for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
            assert np.array_equal(got[name][k * nq:(k + 1) * nq], one[name]), (k, name)
            assert np.array_equal(one[name], want[name][k * nq:(k + 1) * nq]), (k, name)


# ------------------------------------------------------ search by projection
def test_search_by_projection(fe_orb):
    img = synth.textured_canvas(1241, 376, 50)
    f = fe_orb.extract_keypoints(img)
    npts = 20000
    uv, pdesc = synth.local_map(f.descriptors, f.pt, npts, 1241, 376, seed=7)
    ids = np.arange(npts, dtype=np.int32)
    fe_orb.map_put(ids, pdesc)
    # visible subset, ascending, like rPointsIdx
    vis = np.nonzero(synth.rand_below(8, 1, np.arange(npts), 10) < 8)[0].astype(np.int32)
    inl = synth.rand_below(8, 2, np.arange(1500), len(f)).astype(np.int32)
    got = fe_orb.search_by_projection(f, inl, uv[vis], vis)
    want = O.search_by_projection(f.pt, f.descriptors, inl, uv[vis], pdesc[vis], vis)
    assert np.array_equal(got, want)
    assert (got >= 0).sum() > 200
    hit = got[got >= 0]
    assert len(np.unique(hit)) == len(hit)


def test_search_by_projection_dense_cluster(fe_orb):
    # more candidates per keypoint than the per-inlier list holds, and far more
    # inliers than points: exercises list overflow, rescans and exhaustion
    img = synth.textured_canvas(320, 240, 51)
    f = fe_orb.extract_keypoints(img)
    assert len(f) > 50
    k0 = 10
    n = 3000
    uv = np.tile(f.pt[k0], (n, 1)).astype(np.float32)
    uv += (synth.rand_below(9, 1, np.arange(2 * n), 41).reshape(n, 2) - 20).astype(np.float32)
    pdesc = np.tile(f.descriptors[k0], (n, 1))
    flips = synth.rand_below(9, 2, np.arange(n), 32)
    for i in range(n):
        pdesc[i, flips[i]] ^= 1 + (i % 7)
    ids = np.arange(100, 100 + n, dtype=np.int32)
    fe_orb.map_put(ids, pdesc)
    inl = np.full(200, k0, np.int32)
    inl[::3] = 11
    got = fe_orb.search_by_projection(f, inl, uv, ids)
    want = O.search_by_projection(f.pt, f.descriptors, inl, uv, pdesc, ids)
    assert np.array_equal(got, want)
    assert (got >= 0).sum() >= 100


def _claim_slot(j):
    # the hash slot k_proj_resolve enters a proposed position in
    return ((int(j) * 2654435761) & 0xffffffff) >> 20


@pytest.mark.parametrize("nA,nB", [(32, 32), (40, 0), (32, 31), (64, 0), (1, 63)])
def test_search_by_projection_serial_chain_of_64(fe_orb, nA, nB):
    # A chunk of 64 inliers that resolves one lane per round, 64 rounds: 32 inliers
    # on keypoint A contest A's candidates one after the other, then 32 inliers on
    # keypoint B do the same with B's.  The last proposal (round 64) goes to a
    # hash slot that still holds a proposal of round 1.  Lists longer than the 32
    # stored candidates (nA = 40, 64) run dry and are re-scanned.
    img = synth.textured_canvas(640, 480, 57)
    f = fe_orb.extract_keypoints(img)
    xy = f.pt
    kA = 5
    far = np.nonzero(np.hypot(xy[:, 0] - xy[kA, 0], xy[:, 1] - xy[kA, 1]) > 150)[0]
    kB = int(far[0])
    R = 30000
    uv = np.stack([3000 + np.arange(R, dtype=np.float32), np.full(R, 3000, np.float32)], 1)  # filler, far away
    desc = synth.random_descriptors(R, 23, 2)
    a_idx = [100 + 3 * t for t in range(nA)]
    last = next(j for j in range(2000, R) if _claim_slot(j) == _claim_slot(a_idx[0]))
    b_idx = [1000 + 5 * t for t in range(max(nB - 1, 0))] + ([last] if nB else [])
    for k, idxs in ((kA, a_idx), (kB, b_idx)):
        bits = np.unpackbits(f.descriptors[k], bitorder="little")
        for t, j in enumerate(idxs):
            b = bits.copy()
            b[:t] ^= 1  # candidate t lies t bits away: the list order is the order of idxs
            desc[j] = np.packbits(b, bitorder="little")
            uv[j] = xy[k] + np.array([(t % 7) - 3, (t % 5) - 2], np.float32)
    ids = np.arange(R, dtype=np.int32)
    fe_orb.map_put(ids, desc)
    inl = np.array([kA] * 32 + [kB] * 32, np.int32)
    got = fe_orb.search_by_projection(f, inl, uv, ids)
    want = O.search_by_projection(xy, f.descriptors, inl, uv, desc, ids)
    assert np.array_equal(got, want)
    assert list(got[:min(nA, 32)]) == a_idx[:32] and list(got[32:32 + nB]) == b_idx[:32]
    # the same chain not aligned with a chunk of 64: 10 unrelated inliers in front
    pre = np.arange(20, 30, dtype=np.int32)
    inl2 = np.concatenate([pre, inl])
    assert np.array_equal(fe_orb.search_by_projection(f, inl2, uv, ids),
                          O.search_by_projection(xy, f.descriptors, inl2, uv, desc, ids))


def test_search_by_projection_points_outside_the_image(fe_orb):
    # projected positions outside the frame, far away, infinite and NaN: they fall
    # into the clamped border cells of the grid and are judged by the gate alone
    img = synth.textured_canvas(640, 480, 58)
    f = fe_orb.extract_keypoints(img)
    xy = f.pt
    order = np.argsort(xy[:, 0] + xy[:, 1])
    corner = order[:40]  # keypoints nearest the top-left corner ...
    other = np.argsort(-(xy[:, 0] + xy[:, 1]))[:40]  # ... and the bottom-right one
    ks = np.concatenate([corner, other])
    n = len(ks)
    uv = np.empty((4 * n, 2), np.float32)
    desc = np.empty((4 * n, 32), np.uint8)
    for i, k in enumerate(ks):
        sgn = -1.0 if i < 40 else 1.0
        uv[4 * i] = xy[k] + sgn * np.array([60.0, 45.0], np.float32)      # within 80 px, outside the image
        uv[4 * i + 1] = xy[k] + sgn * np.array([79.0, 15.0], np.float32)  # just outside the radius
        uv[4 * i + 2] = [np.inf, xy[k][1]] if i % 2 else [np.nan, np.nan]
        uv[4 * i + 3] = sgn * np.array([1e30, -1e30], np.float32)
        desc[4 * i:4 * i + 4] = f.descriptors[k]
    assert (uv[0::4] < 0).any() and (uv[0::4, 0] > 640).any()
    ids = np.arange(4 * n, dtype=np.int32)
    fe_orb.map_put(ids, desc)
    inl = ks.astype(np.int32)
    got = fe_orb.search_by_projection(f, inl, uv, ids, radius=80.0)
    want = O.search_by_projection(xy, f.descriptors, inl, uv, desc, ids, radius=80.0)
    assert np.array_equal(got, want)
    assert (got >= 0).sum() >= n // 2


@pytest.mark.parametrize("w,h,n,seed", [(1241, 376, 20000, 7), (640, 480, 3000, 8), (640, 480, 1, 9),
                                        (640, 480, 257, 10)])
def test_reproject_map_points(fe_orb, w, h, n, seed):
    pose, K, xyz, th = synth.camera_scene(n, w, h, seed)
    fe_orb.map_put_xyz(np.arange(n), xyz)
    uv, idx = fe_orb.reproject(pose, K, th, w, h, n)
    want_uv, want_idx = O.reproject(pose, K, xyz, th, w, h)
    assert np.array_equal(idx, want_idx)
    assert np.array_equal(uv.view(np.uint32), want_uv.view(np.uint32))
    assert fe_orb.reproject(pose, K, th, w, h, n, download=False) == len(want_idx)
    assert fe_orb.reproject(pose, K, th, w, h, 0) [1].size == 0


def test_reproject_singular_pose_is_an_error(fe_orb):
    from slam_amd.frontend import SlamfeError
    with pytest.raises(SlamfeError):
        fe_orb.reproject(np.ones((4, 4), np.float32), np.eye(3, dtype=np.float32), 0.0, 640, 480, 10)


def test_reproject_then_search_on_device(fe_orb):
    # Init.cc:203-259 end to end: reprojection feeds the search without a host round trip
    w, h, n = 640, 480, 6000
    img = synth.textured_canvas(w, h, 71)
    f = fe_orb.extract_keypoints(img)
    pose, K, xyz, th = synth.camera_scene(n, w, h, 12)
    want_uv, want_idx = O.reproject(pose, K, xyz, th, w, h)
    assert len(want_idx) > 1000
    # descriptors: visible point j looks like the keypoint nearest to its projection (some perturbed)
    xy = np.stack([f.keypoints["x"], f.keypoints["y"]], 1)
    rng = np.random.default_rng(3)
    desc = synth.random_descriptors(n, 13, 1)
    for j, (p, i) in enumerate(zip(want_uv, want_idx)):
        if j % 3:
            k = int(np.argmin(((xy - p) ** 2).sum(1)))
            d = f.descriptors[k].copy()
            d[rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
            desc[i] = d
    fe_orb.map_put(np.arange(n), desc)
    fe_orb.map_put_xyz(np.arange(n), xyz)
    kp_idx = rng.permutation(len(f))[:1500].astype(np.int32)
    assert fe_orb.reproject(pose, K, th, w, h, n, download=False) == len(want_idx)
    got = fe_orb.search_by_projection_visible(f, kp_idx)
    want = O.search_by_projection(xy, f.descriptors, kp_idx, want_uv, desc[want_idx], want_idx)
    assert np.array_equal(got, want)
    assert (got >= 0).sum() > 100
    # the host-round-trip form gives the same answer
    assert np.array_equal(fe_orb.search_by_projection(f, kp_idx, want_uv, want_idx), want)


@pytest.mark.parametrize("n", [6000, 90_000])
def test_visible_set_grid_is_reused_and_rebuilt(n):
    # The grid over the visible set is built by the reprojection call (for the radius of
    # the most recent search) and reused by the searches that follow; a search with
    # another radius, a host-supplied visible set in between, new coordinates or a new
    # pose must each lead to a grid that fits.  90 000 points: the visible set is beyond
    # what the one-workgroup grid kernel holds (count / scan / scatter launches).
    from slam_amd.frontend import FrontEnd
    w, h = 640, 480
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=2000, max_width=w, max_height=h,
                  max_batch=1, map_capacity=n) as fe:
        f = fe.extract_keypoints(synth.textured_canvas(w, h, 72))
        xy = np.stack([f.keypoints["x"], f.keypoints["y"]], 1)
        pose, K, xyz, th = synth.camera_scene(n, w, h, 14)
        rng = np.random.default_rng(8)
        desc = synth.random_descriptors(n, 17, 1)
        near = rng.integers(0, len(f), n)
        take = rng.random(n) < 0.6  # these look like some keypoint of the frame, a few bits off
        desc[take] = f.descriptors[near[take]]
        desc[take, rng.integers(0, 32, take.sum())] ^= np.uint8(4)
        fe.map_put(np.arange(n), desc)
        fe.map_put_xyz(np.arange(n), xyz)
        kp_idx = rng.permutation(len(f))[:700].astype(np.int32)

        def check(pose_, xyz_, radius, label):
            uv, idx = O.reproject(pose_, K, xyz_, th, w, h)
            want = O.search_by_projection(xy, f.descriptors, kp_idx, uv, desc[idx], idx, radius=radius)
            got = fe.search_by_projection_visible(f, kp_idx, radius=radius)
            assert np.array_equal(got, want), label
            return uv, idx, want

        assert fe.reproject(pose, K, th, w, h, n, download=False) > (1000 if n < 10_000 else SLAMFE_ONE_WG)
        uv, idx, want32 = check(pose, xyz, 32.0, "grid from the reprojection call")
        assert (want32 >= 0).sum() > 50
        check(pose, xyz, 32.0, "same grid again")
        check(pose, xyz, 9.5, "other radius: rebuilt")
        check(pose, xyz, 9.5, "and kept")
        # a host-supplied visible set uses the same buffers
        sub = slice(0, len(idx) // 2)
        got = fe.search_by_projection(f, kp_idx, uv[sub], idx[sub], radius=20.0)
        assert np.array_equal(got, O.search_by_projection(xy, f.descriptors, kp_idx, uv[sub], desc[idx[sub]],
                                                           idx[sub], radius=20.0))
        check(pose, xyz, 20.0, "visible set after a host-supplied one")
        # new pose: the next reprojection builds the grid for the last radius (20)
        pose2 = pose.copy()
        pose2[0, 3] += 0.3
        fe.reproject(pose2, K, th, w, h, n, download=False)
        check(pose2, xyz, 20.0, "new pose, last radius")
        check(pose2, xyz, 32.0, "new pose, other radius")
        # new coordinates drop the visible set altogether
        xyz2 = xyz.copy()
        xyz2[: n // 3, 0] += 0.2
        fe.map_put_xyz(np.arange(n), xyz2)
        from slam_amd.frontend import SlamfeError
        with pytest.raises(SlamfeError):
            fe.search_by_projection_visible(f, kp_idx)
        fe.reproject(pose2, K, th, w, h, n, download=False)
        check(pose2, xyz2, 32.0, "after new coordinates")


SLAMFE_ONE_WG = 20480  # points the one-workgroup grid kernel takes


@pytest.mark.parametrize("radius", [32.0, 31.7, 0.0, 1.0, 5.000001, 1e9, -1.0])
def test_search_by_projection_radius_boundary(fe_orb, radius):
    # points placed on and next to the circle of the given radius around their
    # keypoint, identical descriptors: only the pixel gate (float differences,
    # double squares, root narrowed to float, Init.cc:389-391) decides
    img = synth.textured_canvas(640, 480, 53)
    f = fe_orb.extract_keypoints(img)
    n = 300
    kp_idx = np.arange(n, dtype=np.int32)
    xy = f.pt[:n]
    r32 = np.float32(abs(radius) if abs(radius) < 1e6 else 32.0)
    ang = np.linspace(0, 2 * np.pi, n, endpoint=False)
    base = np.stack([np.cos(ang), np.sin(ang)], 1)
    uv = (xy.astype(np.float64) + base * float(r32)).astype(np.float32)
    # nudge a third of the points outwards and a third inwards by one float step
    uv[0::3] = np.nextafter(uv[0::3], np.float32(1e9) * np.sign(base[0::3]).astype(np.float32))
    uv[1::3] = np.nextafter(uv[1::3], xy[1::3])
    uv[::7] = xy[::7] + np.array([r32, 0], np.float32)   # exactly on the axis at distance r
    ids = np.arange(n, dtype=np.int32)
    fe_orb.map_put(ids, f.descriptors[:n])
    got = fe_orb.search_by_projection(f, kp_idx, uv, ids, radius=radius)
    want = O.search_by_projection(f.pt, f.descriptors, kp_idx, uv, f.descriptors[:n], ids, radius=radius)
    assert np.array_equal(got, want)
    if radius in (32.0, 31.7, 1.0, 5.000001):
        assert 0 < (got >= 0).sum()
    if radius < 0:
        assert (got == -1).all()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 3000])
def test_triangulate_point_pairs(fe_orb, n):
    # Init.cc:260 with Init.cc:5-17 for all pairs at once; the float contract is one
    # source compiled for host and device, so the bits agree
    P1, P2, p1, p2 = synth.two_view_scene(n, 640, 480, 30 + n)
    got = fe_orb.triangulate(P1, P2, p1, p2)
    want = O.triangulate(P1, P2, p1, p2)
    assert got.shape == (n, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert len(fe_orb.triangulate(P1, P2, np.zeros((0, 2), np.float32), np.zeros((0, 2), np.float32))) == 0


def test_search_by_projection_no_points(fe_orb):
    img = synth.textured_canvas(200, 160, 52)
    f = fe_orb.extract_keypoints(img)
    got = fe_orb.search_by_projection(f, [0, 1, 2], np.zeros((0, 2), np.float32),
                                      np.zeros(0, np.int32))
    assert list(got) == [-1, -1, -1]


def test_match_resident_batch(fe_orb):
    import torch
    frames = synth.frame_stream(640, 480, 4, seed=6)
    fs = fe_orb.extract_batch(list(frames))
    cap = fe_orb.max_keypoints
    out = torch.zeros((3, cap, 16), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    fe_orb.match_resident_batch(fs[0].handle, fs[1].handle, 3, out.data_ptr())
    fe_orb.sync()
    got = out.cpu().numpy()
    for i in range(3):
        want = O.match_knn2(fs[i].descriptors, fs[i + 1].descriptors)
        g = got[i].reshape(-1).view(abi.MATCH2_DTYPE)[:len(want)]
        for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
            assert np.array_equal(g[name], want[name]), (i, name)


# ------------------------------------------------- BASELINE configs at full size
def _full_size_case(w, h, nfeat, seed, batch=1):
    from slam_amd.frontend import FrontEnd
    img = synth.textured_canvas(w, h, seed)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=nfeat, max_width=w, max_height=h,
                  max_batch=batch) as fe:
        f = fe.extract_keypoints(img)
        p = oracle_params(fe)
        kps, desc = O.extract(p, img)
        assert len(kps) >= nfeat
        assert_frame_equal(f, kps, desc)
        # every level's quota is saturated on the benchmark texture
        assert np.all(np.bincount(kps["octave"], minlength=8) >= O.level_quotas(nfeat))
    return f


def test_config3_1920x1080_5000_features():
    _full_size_case(1920, 1080, 5000, 100)


@pytest.mark.parametrize("nframes", [9])
def test_config3_1080p_batch_through_the_resident_calls(nframes):
    # BASELINE config 3 as the benchmark drives it: a batch of 1920x1080 frames, 5000
    # features each, resident in device memory, through slamfe_extract_resident and
    # slamfe_match_resident_batch.  Nine frames leave the second XCD group ragged
    # (frame = blockIdx.z * 8 + blockIdx.x) and k_resize's four-frames-per-wave
    # layout with an incomplete group.  Every frame and every pair is compared.
    import torch
    from slam_amd.frontend import FrontEnd
    w, h, nfeat = 1920, 1080, 5000
    frames = synth.frame_stream(w, h, nframes, seed=33)
    d = torch.from_numpy(frames).cuda()
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=nfeat, max_width=w, max_height=h,
                  max_batch=nframes, frame_slots=2 * nframes, map_capacity=0) as fe:
        cap = fe.max_keypoints
        out = torch.zeros((nframes - 1, cap, 16), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()  # the fills run on torch's stream, the library on its own
        hd = fe.extract_resident(d.data_ptr(), nframes, w, h, w, w * h, 1)
        fe.match_resident_batch(int(hd[0]), int(hd[1]), nframes - 1, out.data_ptr())
        fe.sync()
        p = oracle_params(fe)
        want = [O.extract(p, fr) for fr in frames]
        quotas = O.level_quotas(nfeat)
        for i in range(nframes):
            kps, desc = want[i]
            assert len(kps) >= nfeat and np.all(np.bincount(kps["octave"], minlength=8) >= quotas)
            assert_frame_equal(fe.frame_download(int(hd[i])), kps, desc)
        rows = out.cpu().numpy()
        for i in range(nframes - 1):
            m = np.frombuffer(rows[i].tobytes(), dtype=abi.MATCH2_DTYPE)[:len(want[i][0])]
            ref = O.match_knn2(want[i][1], want[i + 1][1])
            for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
                assert np.array_equal(m[name], ref[name]), (i, name)


def test_config5_3840x2160_10000_features():
    # 60 tiles per band: the selection sort needs the >64 KiB dynamic-LDS path
    _full_size_case(3840, 2160, 10000, 5)


def test_config2_kitti_stream_with_20k_map():
    from slam_amd.frontend import FrontEnd
    frames = synth.frame_stream(1241, 376, 2, seed=9)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=3000, max_width=1241,
                  max_height=376, max_batch=2, map_capacity=20000) as fe:
        f0, f1 = fe.extract_batch(list(frames))
        p = oracle_params(fe)
        for img, f in zip(frames, (f0, f1)):
            kps, desc = O.extract(p, img)
            assert len(kps) >= 3000
            assert_frame_equal(f, kps, desc)
        m = fe.knn_match_frames(f0, f1)
        want = O.match_knn2(f0.descriptors, f1.descriptors)
        for name in ("idx0", "idx1", "d0", "d1", "ratio_pass"):
            assert np.array_equal(m[name], want[name]), name
        uv, pdesc = synth.local_map(f1.descriptors, f1.pt, 20000, 1241, 376, seed=11)
        ids = np.arange(20000, dtype=np.int32)
        fe.map_put(ids, pdesc)
        inl = m["idx0"][m["ratio_pass"] == 1].astype(np.int32)  # goodMatches[i].trainIdx
        got = fe.search_by_projection(f1, inl, uv, ids)
        assert np.array_equal(got, O.search_by_projection(f1.pt, f1.descriptors, inl, uv, pdesc, ids))
        assert (got >= 0).sum() > 500


def test_extract_all_dark_and_all_bright_frames(fe_orb):
    for v in (0, 255):
        assert len(fe_orb.extract_keypoints(np.full((480, 640), v, np.uint8))) == 0


def test_extract_checkerboard_ties(fe_orb):
    # a regular pattern produces many equal scores: NMS ties and a cut-off bin
    # holding far more keypoints than the quota
    yy, xx = np.mgrid[0:480, 0:640]
    img = (((xx // 8) + (yy // 8)) % 2 * 200 + 20).astype(np.uint8)
    img[::7, ::5] += 9
    f = fe_orb.extract_keypoints(img) if True else None
    kps, desc = O.extract(oracle_params(fe_orb), img)
    assert_frame_equal(f, kps, desc)


# ------------------------------------------- relocalisation at its full size
def test_reloc_2000_by_1M_properties():
    # 2000 queries against 1,000,000 map rows (2 * 10^9 comparisons): too many for
    # the CPU checker, so the result is pinned by properties that do not depend on
    # the size -- planted rows with known distances, the two independent kernels
    # (matrix cores and vector-ALU popcount) agreeing bit for bit, the 8-way
    # sharded merge equal to the single call, and a CPU check of the reported
    # distances for a sample of queries.
    import torch
    from slam_amd.frontend import FrontEnd
    NQ, NT, SH = 2000, 1_000_000, 8
    train = synth.random_descriptors(NT, 3, 6)
    query = synth.random_descriptors(NQ, 3, 5)
    # plant: query i (i < 500) is train row p(i) with i % 20 bits flipped, and a
    # second copy of that row with 21 + i % 5 flips sits at row q(i) > p(i)
    rows = synth.rand_below(77, 1, np.arange(500), NT // 2 - 1)
    rows2 = NT // 2 + synth.rand_below(77, 2, np.arange(500), NT // 2)
    assert len(np.unique(rows)) + len(np.unique(rows2)) > 990
    flips1 = np.arange(500) % 20
    flips2 = 21 + np.arange(500) % 5
    keep = np.ones(500, bool)
    seen = set()
    for i in range(500):
        if rows[i] in seen or rows2[i] in seen:
            keep[i] = False
            continue
        seen.add(int(rows[i]))
        seen.add(int(rows2[i]))
        bits = np.unpackbits(train[rows[i]], bitorder="little")
        b1 = bits.copy()
        pos = np.argsort(synth.hash_u64(79, 4, i * 256 + np.arange(256)))
        b1[pos[:flips1[i]]] ^= 1
        query[i] = np.packbits(b1, bitorder="little")
        b2 = bits.copy()
        b2[pos[40:40 + flips2[i]]] ^= 1
        train[rows2[i]] = np.packbits(b2, bitorder="little")
    dq = torch.from_numpy(query).cuda()
    dt = torch.from_numpy(train).cuda()
    outs = {}
    for kern in (abi.MATCH_MATRIX, abi.MATCH_POPCOUNT):
        keys = torch.empty((NQ, 2), dtype=torch.int64, device="cuda")
        out = torch.zeros(NQ * 16, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0,
                      match_kernel=kern) as fe:
            fe.match_shard_keys(dq.data_ptr(), NQ, dt.data_ptr(), NT, 0, keys.data_ptr())
            fe.merge_shard_keys(keys.data_ptr(), 1, NQ, out.data_ptr())
            fe.sync()
            outs[kern] = out.cpu().numpy().view(abi.MATCH2_DTYPE).copy()
            if kern == abi.MATCH_MATRIX:
                # the same job in 8 uneven shards, merged
                keys8 = torch.empty((SH, NQ, 2), dtype=torch.int64, device="cuda")
                out8 = torch.zeros(NQ * 16, dtype=torch.uint8, device="cuda")
                torch.cuda.synchronize()
                bounds = [0, 1, 125_000, 250_001, 375_000, 500_000, 700_000, 999_999, NT]
                for s in range(SH):
                    lo, hi = bounds[s], bounds[s + 1]
                    fe.match_shard_keys(dq.data_ptr(), NQ, dt[lo:hi].data_ptr(), hi - lo, lo,
                                        keys8[s].data_ptr())
                fe.merge_shard_keys(keys8.data_ptr(), SH, NQ, out8.data_ptr())
                fe.sync()
                sharded = out8.cpu().numpy().view(abi.MATCH2_DTYPE)
    m, v = outs[abi.MATCH_MATRIX], outs[abi.MATCH_POPCOUNT]
    for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
        assert np.array_equal(m[name], v[name]), name        # two kernels, one answer
        assert np.array_equal(m[name], sharded[name]), name  # sharding does not change it
    k = np.nonzero(keep)[0]
    assert len(k) > 480
    assert np.array_equal(m["idx0"][k], rows[k]) and np.array_equal(m["d0"][k], flips1[k])
    # the second copy differs from the query in flips1 + flips2 bits (disjoint positions)
    assert np.array_equal(m["idx1"][k], rows2[k]) and np.array_equal(m["d1"][k], (flips1 + flips2)[k])
    assert np.all(m["ratio_pass"][k] == (4 * flips1[k] < 3 * (flips1 + flips2)[k]))
    # the reported distances are the true distances to the reported rows
    for i in list(range(500, 520)) + [1999]:
        for idx, d in ((m["idx0"][i], m["d0"][i]), (m["idx1"][i], m["d1"][i])):
            assert int(np.unpackbits(query[i] ^ train[idx]).sum()) == int(d)
        assert m["d0"][i] <= m["d1"][i] and m["nfound"][i] == 2
        # random 256-bit rows: the best of a million lies far below the mean of 128
        assert 60 < m["d0"][i] < 100


def test_stage_probe_counts_intervals():
    # the measurement hook: one interval per extraction while the probe is set,
    # none after it is switched off, and the frame results are not affected
    import torch
    from slam_amd.frontend import FrontEnd
    frames = synth.frame_stream(320, 240, 3, seed=5)
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=300, max_width=320, max_height=240,
                  max_batch=3, frame_slots=12, map_capacity=0) as fe:
        h0 = fe.extract_resident(d.data_ptr(), 3, 320, 240, 320, 320 * 240, 1)
        ref = fe.frame_download(int(h0[1]))
        fe.probe_stage(2)
        for _ in range(3):
            h = fe.extract_resident(d.data_ptr(), 3, 320, 240, 320, 320 * 240, 1)
        ms, n = fe.probe_read()
        assert n == 3 and 0.0 < ms < 50.0
        got = fe.frame_download(int(h[1]))
        assert np.array_equal(got.keypoints, ref.keypoints) and np.array_equal(got.descriptors, ref.descriptors)
        fe.probe_stage(0)
        fe.extract_resident(d.data_ptr(), 3, 320, 240, 320, 320 * 240, 1)
        assert fe.probe_read()[1] == 0


@pytest.mark.parametrize("w,h", [(8, 8), (9, 33), (40, 40), (63, 70), (64, 64), (65, 65), (66, 200), (200, 66)])
def test_extract_tiny_frames(w, h):
    # frames smaller than a tile, than the border band, than the blur's reach:
    # upper pyramid levels shrink to a few pixels; mostly nothing can be detected
    from slam_amd.frontend import FrontEnd
    img = synth.textured_canvas(w, h, 5 + w)
    for mode in (abi.MODE_ORB, abi.MODE_PROVIDED):
        for edge in (22, 31):
            with FrontEnd(mode=mode, fast_threshold=10, nfeatures=500, max_width=w, max_height=h,
                          edge_threshold=edge) as fe:
                f = fe.extract_keypoints(img)
                kps, desc = O.extract(oracle_params(fe), img)
                assert_frame_equal(f, kps, desc)


@pytest.mark.parametrize("w,h,nf", [(4096, 4096, 20000), (4096, 70, 3000), (70, 4096, 3000)])
def test_extract_maximum_dimensions(w, h, nf):
    # the largest frame the library accepts (SLAMFE_MAX_DIM = 4096) and the two
    # most lopsided ones
    from slam_amd.frontend import FrontEnd
    img = synth.textured_canvas(w, h, 11)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=nf, max_width=w, max_height=h) as fe:
        f = fe.extract_keypoints(img)
        kps, desc = O.extract(oracle_params(fe), img)
        assert len(kps) > 200
        assert_frame_equal(f, kps, desc)


def test_match_maximum_train_rows():
    # the largest shard one call accepts (2^23 - 1 rows): rows planted at chunk and
    # tile boundaries and at both ends come back with their distance, both kernels
    # agree, one row more is refused
    import torch
    from slam_amd.frontend import FrontEnd, SlamfeError
    NT, NQ = (1 << 23) - 1, 300
    train = synth.random_descriptors(NT, 5, 1)
    query = synth.random_descriptors(NQ, 5, 2)
    rows = np.array([0, 1, 31, 32, 2047, 2048, 2049, 4095, NT - 2049, NT - 2048, NT - 33, NT - 32,
                     NT - 2, NT - 1])
    for i, r in enumerate(rows):
        query[i] = train[r]
        query[i, 0] ^= (1 << (i % 8))  # one flipped bit
    dq = torch.from_numpy(query).cuda()
    dt = torch.from_numpy(train).cuda()
    res = {}
    for kern in (abi.MATCH_MATRIX, abi.MATCH_POPCOUNT):
        keys = torch.empty((NQ, 2), dtype=torch.int64, device="cuda")
        out = torch.zeros(NQ * 16, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0,
                      match_kernel=kern) as fe:
            fe.match_shard_keys(dq.data_ptr(), NQ, dt.data_ptr(), NT, 0, keys.data_ptr())
            fe.merge_shard_keys(keys.data_ptr(), 1, NQ, out.data_ptr())
            fe.sync()
            res[kern] = out.cpu().numpy().view(abi.MATCH2_DTYPE).copy()
            with pytest.raises(SlamfeError):
                fe.match_shard_keys(dq.data_ptr(), NQ, dt.data_ptr(), NT + 1, 0, keys.data_ptr())
    m, v = res[abi.MATCH_MATRIX], res[abi.MATCH_POPCOUNT]
    for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
        assert np.array_equal(m[name], v[name]), name
    assert np.array_equal(m["idx0"][:len(rows)], rows) and np.all(m["d0"][:len(rows)] == 1)


def test_search_by_projection_maximum_visible_points():
    # 2^19 visible points (the most one call accepts): same answer as the CPU
    # checker for 200 inliers; one point more is refused
    from slam_amd.frontend import FrontEnd, SlamfeError
    R = 1 << 19
    img = synth.textured_canvas(1241, 376, 50)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=3000, max_width=1241, max_height=376,
                  map_capacity=R + 8) as fe:
        f = fe.extract_keypoints(img)
        uv, pdesc = synth.local_map(f.descriptors, f.pt, R, 1241, 376, seed=7)
        ids = np.arange(R, dtype=np.int32)
        fe.map_put(ids, pdesc)
        inl = synth.rand_below(8, 2, np.arange(200), len(f)).astype(np.int32)
        got = fe.search_by_projection(f, inl, uv, ids)
        want = O.search_by_projection(f.pt, f.descriptors, inl, uv, pdesc, ids)
        assert np.array_equal(got, want) and (got >= 0).sum() > 150
        with pytest.raises(SlamfeError) as e:
            fe.search_by_projection(f, inl, np.concatenate([uv, uv[:1]]), np.arange(R + 1, dtype=np.int32))
        assert e.value.code == abi.SLAMFE_E_CAPACITY


def test_reproject_and_triangulate_a_million_points():
    # sizes far beyond the reference's maps: 2^20 map points through the
    # reprojection (ordered compaction over 4096 workgroups), 10^6 pairs through
    # the triangulation; bit-exact against the CPU checker
    from slam_amd.frontend import FrontEnd
    n = 1 << 20
    with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=n) as fe:
        pose, K, xyz, th = synth.camera_scene(n, 1241, 376, 21)
        fe.map_put_xyz(np.arange(n), xyz)
        uv, idx = fe.reproject(pose, K, th, 1241, 376, n)
        want_uv, want_idx = O.reproject(pose, K, xyz, th, 1241, 376)
        assert len(want_idx) > n // 20
        assert np.array_equal(idx, want_idx)
        assert np.array_equal(uv.view(np.uint32), want_uv.view(np.uint32))
        m = 1_000_000
        P1, P2, p1, p2 = synth.two_view_scene(m, 640, 480, 33)
        got = fe.triangulate(P1, P2, p1, p2)
        want = O.triangulate(P1, P2, p1, p2)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ------------------------------------------- float routines against numpy, not shared source
def test_float_routines_against_numpy_definitions():
    """The keypoint angle (fastAtan2), the descriptor rotation (cos / sin) and the
    triangulation share one source file between the kernels and the CPU checker; here
    the GPU results are held against independent numpy definitions instead: moments
    summed pixel by pixel and numpy's arctan2, the descriptor evaluated with numpy's
    cos / sin, and numpy.linalg.svd in float64."""
    import slow_defs as S
    from slam_amd.frontend import FrontEnd
    from slam_amd import tables
    img = synth.textured_canvas(320, 240, 77)
    with FrontEnd(mode=abi.MODE_ORB, fast_threshold=20, nfeatures=400, nlevels=1, max_width=320,
                  max_height=240, max_batch=1, map_capacity=0) as fe:
        f = fe.extract_keypoints(img)
        assert len(f) > 200
        blurred = S.blur7(img)
        worst = 0.0
        ndesc = 0
        for k, d in zip(f.keypoints, f.descriptors):
            cx, cy = int(k["x"]), int(k["y"])
            assert k["octave"] == 0 and cx == k["x"] and cy == k["y"]
            m10, m01 = S.ic_moments(img, cx, cy)
            want = np.degrees(np.arctan2(float(m01), float(m10))) % 360.0
            diff = abs((float(k["angle"]) - want + 180.0) % 360.0 - 180.0)
            worst = max(worst, diff)
            # cv::fastAtan2 is a polynomial good to 0.3 degrees
            assert diff < 0.3, (cx, cy, k["angle"], want)
            # the descriptor from numpy's cos / sin of the angle the GPU reports
            assert np.array_equal(d, S.brief256(blurred, cx, cy, k["angle"], tables.PATTERN)), (cx, cy)
            ndesc += 1
        assert ndesc == len(f) and worst > 0.0  # (a polynomial, not libm: some error is expected)
        # triangulation: homogeneous least squares by float64 SVD
        P1, P2, p1, p2 = synth.two_view_scene(300, 640, 480, 5)
        got = fe.triangulate(P1, P2, p1, p2).astype(np.float64)
        P1d, P2d = np.asarray(P1, np.float64).reshape(3, 4), np.asarray(P2, np.float64).reshape(3, 4)
        checked = 0
        for (x1, y1), (x2, y2), g in zip(np.asarray(p1, np.float64), np.asarray(p2, np.float64), got):
            A = np.stack([x1 * P1d[2] - P1d[0], y1 * P1d[2] - P1d[1], x2 * P2d[2] - P2d[0], y2 * P2d[2] - P2d[1]])
            sv = np.linalg.svd(A, compute_uv=False)
            v = np.linalg.svd(A)[2][3]
            if abs(v[3]) < 1e-6 or (sv[2] - sv[3]) / sv[0] < 1e-3:
                continue  # ill-conditioned pair: float32 and float64 may differ freely
            want = v[:3] / v[3]
            assert np.abs(g - want).max() <= 2e-3 * max(1.0, np.abs(want).max()), (g, want)
            checked += 1
        assert checked > 200
