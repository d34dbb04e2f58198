"""Deterministic synthetic inputs for tests and benchmarks.

Everything is derived from a counter-based splitmix64 hash with integer
arithmetic only, so the same bytes are produced on every machine and with every
numpy version (numpy's own RNG streams are not used).

The reference reads its frames from a video that is not in its repository
(main.cc:6), so all inputs here are synthetic:

* ``textured_canvas`` / ``textured_frame`` -- three octaves of value noise plus
  random grey rectangles; dense in FAST corners at threshold 20.
* ``frame_pair`` -- a frame and the same scene shifted by (+7, +3) px with +-2
  grey levels of noise (known ground-truth motion).
* ``frame_stream`` -- n frames panning over one canvas.
* ``descriptor_sets`` -- random 256-bit rows, queries derived from train rows by
  bit flips, with exact duplicates to exercise tie-breaks.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def splitmix64(x):
    """splitmix64 finaliser of (x + golden ratio), elementwise on uint64."""
    with np.errstate(over="ignore"):
        z = np.asarray(x, dtype=np.uint64) + _GOLD
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def hash_u64(seed: int, stream: int, idx):
    """Counter-based random uint64: hash of (seed, stream, idx)."""
    with np.errstate(over="ignore"):
        base = splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF))
        base = splitmix64(base ^ np.uint64(stream & 0xFFFFFFFFFFFFFFFF))
        return splitmix64(base ^ (np.asarray(idx, dtype=np.uint64) * _GOLD))


def rand_below(seed: int, stream: int, idx, n: int):
    """Uniform integers in [0, n) (modulo bias is irrelevant here)."""
    return (hash_u64(seed, stream, idx) % np.uint64(n)).astype(np.int64)


def _value_noise(h: int, w: int, cell: int, amp: int, seed: int, stream: int):
    gw = w // cell + 2
    gh = h // cell + 2
    lat = rand_below(seed, stream, np.arange(gw * gh), 2 * amp + 1).reshape(gh, gw) - amp
    ys = np.arange(h)
    xs = np.arange(w)
    gy, fy = ys // cell, ys % cell
    gx, fx = xs // cell, xs % cell
    l00 = lat[gy][:, gx]
    l10 = lat[gy][:, gx + 1]
    l01 = lat[gy + 1][:, gx]
    l11 = lat[gy + 1][:, gx + 1]
    fx = fx[None, :]
    fy = fy[:, None]
    top = l00 * (cell - fx) + l10 * fx
    bot = l01 * (cell - fx) + l11 * fx
    return (top * (cell - fy) + bot * fy) // (cell * cell)


def textured_canvas(w: int, h: int, seed: int) -> np.ndarray:
    """u8 image: value noise (cells 64/16/4 px, amplitudes 64/48/32 around 128)
    plus 400 * (w*h / 307200) random axis-aligned rectangles (side 6..40 px,
    grey level uniform 0..255), painted in index order."""
    img = np.full((h, w), 128, dtype=np.int64)
    for stream, (cell, amp) in enumerate(((64, 64), (16, 48), (4, 32))):
        img += _value_noise(h, w, cell, amp, seed, 10 + stream)
    img = np.clip(img, 0, 255).astype(np.uint8)
    nrect = (400 * w * h) // 307200
    idx = np.arange(nrect)
    rx = rand_below(seed, 20, idx, w)
    ry = rand_below(seed, 21, idx, h)
    rw = 6 + rand_below(seed, 22, idx, 35)
    rh = 6 + rand_below(seed, 23, idx, 35)
    rg = rand_below(seed, 24, idx, 256)
    for i in range(nrect):
        img[ry[i]:ry[i] + rh[i], rx[i]:rx[i] + rw[i]] = rg[i]
    return img


def _noisy(img: np.ndarray, seed: int, stream: int, amp: int = 2) -> np.ndarray:
    n = rand_below(seed, stream, np.arange(img.size), 2 * amp + 1).reshape(img.shape) - amp
    return np.clip(img.astype(np.int64) + n, 0, 255).astype(np.uint8)


def textured_frame(w: int, h: int, seed: int) -> np.ndarray:
    return np.ascontiguousarray(textured_canvas(w, h, seed))


def frame_pair(w: int, h: int, seed: int = 1, noise_seed: int = 2):
    """(A, B): B shows A's scene moved by (+7, +3) px, with +-2 noise."""
    m = 16
    canvas = textured_canvas(w + 2 * m, h + 2 * m, seed)
    a = np.ascontiguousarray(canvas[m:m + h, m:m + w])
    b = canvas[m - 3:m - 3 + h, m - 7:m - 7 + w]
    return a, np.ascontiguousarray(_noisy(b, noise_seed, 30))


def frame_stream(w: int, h: int, n: int, seed: int = 1) -> np.ndarray:
    """n frames [n, h, w] panning over one canvas (3 px right, 2 px down per
    frame, wrapping every 32 frames), each with its own +-2 noise."""
    m = 96
    canvas = textured_canvas(w + m, h + m, seed)
    out = np.empty((n, h, w), dtype=np.uint8)
    for k in range(n):
        ox = (3 * k) % m
        oy = (2 * k) % m
        out[k] = _noisy(canvas[oy:oy + h, ox:ox + w], seed + 1, 100 + k)
    return out


def random_descriptors(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """[n, 32] uint8, uniform random bits."""
    words = hash_u64(seed, 40 + stream, np.arange(n * 4))
    return words.view(np.uint8).reshape(n, 32).copy()


def descriptor_sets(nq: int, nt: int, seed: int = 3, dup_percent: int = 1):
    """(query [nq,32], train [nt,32]).  Train rows are random; dup_percent % of
    them are exact copies of an earlier row.  Query i is train row src[i] with
    0..40 random bit flips."""
    train = random_descriptors(nt, seed, 0)
    ndup = (nt * dup_percent) // 100
    if ndup and nt > 1:
        di = np.arange(ndup)
        dst = 1 + rand_below(seed, 50, di, nt - 1)
        src = rand_below(seed, 51, di, nt) % dst
        for d, s in zip(dst, src):
            train[d] = train[s]
    qi = np.arange(nq)
    src = rand_below(seed, 52, qi, nt)
    query = train[src].copy()
    nflip = rand_below(seed, 53, qi, 41)
    bits = np.unpackbits(query, axis=1, bitorder="little")
    for i in range(nq):
        pos = rand_below(seed, 54 + 0, i * 64 + np.arange(nflip[i]), 256)
        bits[i, pos] ^= 1
    query = np.packbits(bits, axis=1, bitorder="little")
    return np.ascontiguousarray(query), np.ascontiguousarray(train)


def local_map(frame_desc: np.ndarray, frame_xy: np.ndarray, npoints: int, w: int, h: int,
              seed: int = 7):
    """A local map for search-by-projection: npoints projected positions uv
    [npoints,2] float32 and descriptors [npoints,32].  Half of the points are
    copies of frame descriptors with up to 24 flipped bits, placed within a few
    pixels of that keypoint; the rest are random, uniform in the image."""
    n = npoints
    idx = np.arange(n)
    desc = random_descriptors(n, seed, 1)
    u = rand_below(seed, 60, idx, w * 16).astype(np.float32) / np.float32(16.0)
    v = rand_below(seed, 61, idx, h * 16).astype(np.float32) / np.float32(16.0)
    nk = len(frame_desc)
    if nk:
        half = idx[idx % 2 == 0]
        src = rand_below(seed, 62, half, nk)
        d = frame_desc[src].copy()
        bits = np.unpackbits(d, axis=1, bitorder="little")
        nflip = rand_below(seed, 63, half, 25)
        for r in range(len(half)):
            pos = rand_below(seed, 64, half[r] * 64 + np.arange(nflip[r]), 256)
            bits[r, pos] ^= 1
        desc[half] = np.packbits(bits, axis=1, bitorder="little")
        du = (rand_below(seed, 65, half, 161) - 80).astype(np.float32) / np.float32(8.0)
        dv = (rand_below(seed, 66, half, 161) - 80).astype(np.float32) / np.float32(8.0)
        u[half] = frame_xy[src, 0] + du
        v[half] = frame_xy[src, 1] + dv
    uv = np.stack([u, v], axis=1).astype(np.float32)
    return np.ascontiguousarray(uv), np.ascontiguousarray(desc)


def camera_scene(npoints: int, w: int, h: int, seed: int = 7):
    """A keyframe pose, the intrinsics of Init.cc:90-99 (f = 984, principal point
    at the image centre) and npoints map points: most of them in front of the
    camera inside a frustum somewhat wider than the image, the rest behind it,
    far away or off to the side, so that every gate of Init.cc:205,214-215 fires.
    Returns (pose 4x4 f32, K 3x3 f32, xyz [n,3] f32, th)."""
    idx = np.arange(npoints, dtype=np.uint64)

    def uni(stream):  # uniform in [0, 1)
        return (hash_u64(seed, stream, idx) >> np.uint64(40)).astype(np.float64) / float(1 << 24)

    a = np.array([(int(hash_u64(seed, 90 + i, np.uint64(0))) >> 40) / float(1 << 24) - 0.5 for i in range(6)])
    rx, ry, rz = a[:3] * 0.2
    cx_, sx_, cy_, sy_, cz_, sz_ = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx_, -sx_], [0, sx_, cx_]])
    Ry = np.array([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
    Rz = np.array([[cz_, -sz_, 0], [sz_, cz_, 0], [0, 0, 1]])
    pose = np.eye(4)
    pose[:3, :3] = Rz @ Ry @ Rx
    pose[:3, 3] = a[3:] * np.array([0.6, 0.3, 2.0]) + np.array([0, 0, 1.5])
    pose = pose.astype(np.float32)
    K = np.eye(3, dtype=np.float32)
    K[0, 0] = K[1, 1] = 984.0
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    depth = 0.5 + 30.0 * uni(1) ** 2
    u = (uni(2) * 1.3 - 0.15) * w
    v = (uni(3) * 1.3 - 0.15) * h
    cam = np.stack([(u - w / 2.0) / 984.0 * depth, (v - h / 2.0) / 984.0 * depth, depth, np.ones(npoints)], 1)
    kind = rand_below(seed, 4, idx, 20)
    cam[kind == 0, 2] *= -1          # behind the camera
    cam[kind == 1, 2] += 40          # beyond the depth gate
    world = (pose.astype(np.float64) @ cam.T).T[:, :3]
    th = float(pose[2, 3])
    return pose, K, world.astype(np.float32), th


def scene_from_uv(uv: np.ndarray, w: int, h: int, seed: int = 15):
    """Map points that the reprojection of Init.cc:203-218 puts at the given pixel
    positions (to within float rounding): identity keyframe pose, the intrinsics of
    Init.cc:90-99, depths between 1 and 19 (inside the depth gate around th = 0).
    Returns (pose 4x4 f32, K 3x3 f32, xyz [n,3] f32, th)."""
    uv = np.asarray(uv, np.float64).reshape(-1, 2)
    idx = np.arange(len(uv), dtype=np.uint64)
    z = 1.0 + 18.0 * (hash_u64(seed, 1, idx) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    K = np.eye(3, dtype=np.float32)
    K[0, 0] = K[1, 1] = 984.0
    K[0, 2], K[1, 2] = w / 2.0, h / 2.0
    xyz = np.stack([(uv[:, 0] - w / 2.0) / 984.0 * z, (uv[:, 1] - h / 2.0) / 984.0 * z, z], 1)
    return np.eye(4, dtype=np.float32), K, xyz.astype(np.float32), 0.0


def two_view_scene(npoints: int, w: int, h: int, seed: int = 9):
    """Two camera projection matrices as Init.cc:168-195 builds them (P1 = K[I|0],
    P2 = K[R|t], f = 984) and the pixel positions of npoints scene points in both
    views, with up to half a pixel of noise; a few pairs are degenerate (the same
    pixel in both views).  Returns (P1 3x4 f32, P2 3x4 f32, pts1 [n,2] f32, pts2)."""
    idx = np.arange(npoints, dtype=np.uint64)

    def uni(stream):
        return (hash_u64(seed, stream, idx) >> np.uint64(40)).astype(np.float64) / float(1 << 24)

    K = np.array([[984.0, 0, w / 2.0], [0, 984.0, h / 2.0], [0, 0, 1]])
    ang = 0.05
    R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([[-0.1], [0.01], [0.02]])
    P1 = (K @ np.hstack([np.eye(3), np.zeros((3, 1))])).astype(np.float32)
    P2 = (K @ np.hstack([R, t])).astype(np.float32)
    z = 1.0 + 30.0 * uni(1)
    X = np.stack([(uni(2) - 0.5) * z * 0.6, (uni(3) - 0.5) * z * 0.45, z, np.ones(npoints)], 0)
    a = P1.astype(np.float64) @ X
    b = P2.astype(np.float64) @ X
    pts1 = (a[:2] / a[2]).T + (np.stack([uni(4), uni(5)], 1) - 0.5)
    pts2 = (b[:2] / b[2]).T + (np.stack([uni(6), uni(7)], 1) - 0.5)
    pts2[::37] = pts1[::37]
    return P1, P2, pts1.astype(np.float32), pts2.astype(np.float32)
