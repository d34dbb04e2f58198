"""Independent slow definitions (numpy / pure Python) of each stage.

These restate the *definitions* -- not the optimised algorithms -- so that the C
checker in oracle/ can be cross-checked by something that shares no code with
it.  Used on small inputs only.
"""
from __future__ import annotations

import numpy as np

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
GAUSS = np.array([18, 34, 49, 55, 49, 34, 18], dtype=np.int64)


def ring_stack(img):
    """[16, h-6, w-6] ring pixel values for every interior pixel."""
    h, w = img.shape
    im = img.astype(np.int64)
    return np.stack([im[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])


def is_corner_at(img, t):
    """Segment test at threshold t for all interior pixels: 9 contiguous ring
    pixels all > v+t or all < v-t."""
    h, w = img.shape
    v = img.astype(np.int64)[3:h - 3, 3:w - 3]
    r = ring_stack(img)
    out = np.zeros(v.shape, bool)
    for mask in (r > v + t, r < v - t):
        m2 = np.concatenate([mask, mask[:8]])
        for s in range(16):
            out |= m2[s:s + 9].all(axis=0)
    return out


def fast_score_map(img, thr):
    """score = the largest threshold at which the pixel still passes the
    segment test (0 where it fails at thr)."""
    h, w = img.shape
    score = np.zeros((h, w), np.int64)
    if h < 7 or w < 7:
        return score, np.zeros((h, w), bool)
    alive = is_corner_at(img, thr)
    inner = np.where(alive, thr, 0)
    t = thr + 1
    while alive.any() and t <= 255:
        alive &= is_corner_at(img, t)
        inner[alive] = t
        t += 1
    score[3:h - 3, 3:w - 3] = inner
    return score, np.pad(is_corner_at(img, thr), 3)


def fast_nms(img, thr):
    """Row-major list of (x, y, score) after strict 3x3 non-max suppression."""
    score, corner = fast_score_map(img, thr)
    h, w = img.shape
    p = np.pad(score, 1)
    keep = corner.copy()
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx == 0 and dy == 0:
                continue
            keep &= score > p[1 + dy:1 + dy + h, 1 + dx:1 + dx + w]
    ys, xs = np.nonzero(keep)
    return np.stack([xs, ys, score[ys, xs]], axis=1)


def resize_linear_exact(src, dw, dh):
    sh, sw = src.shape

    def taps(ssize, dsize):
        ofs, c1 = [], []
        scale = 1.0 / (float(dsize) / float(ssize))
        for d in range(dsize):
            f = scale * (d + 0.5) - 0.5
            i0 = int(np.floor(f))
            if i0 >= 0 and ssize > 1:
                if i0 < ssize - 1:
                    ofs.append(i0)
                    c1.append(int(np.rint((f - i0) * 256.0)))
                else:
                    ofs.append(ssize - 1)
                    c1.append(0)
            else:
                ofs.append(0)
                c1.append(0)
        return np.array(ofs), np.array(c1, dtype=np.int64)

    xo, xc = taps(sw, dw)
    yo, yc = taps(sh, dh)
    s = src.astype(np.int64)
    xo1 = np.minimum(xo + 1, sw - 1)
    yo1 = np.minimum(yo + 1, sh - 1)
    hrow = s[:, xo] * (256 - xc) + s[:, xo1] * xc
    out = (hrow[yo] * (256 - yc)[:, None] + hrow[yo1] * yc[:, None] + (1 << 15)) >> 16
    return np.clip(out, 0, 255).astype(np.uint8)


def blur7(img):
    p = np.pad(img.astype(np.int64), 3, mode="reflect")
    h, w = img.shape
    rows = sum(GAUSS[k] * p[:, k:k + w] for k in range(7))
    out = sum(GAUSS[k] * rows[k:k + h] for k in range(7))
    return np.clip((out + (1 << 15)) >> 16, 0, 255).astype(np.uint8)


def ic_moments(img, cx, cy):
    m10 = m01 = 0
    for v in range(-15, 16):
        d = UMAX[abs(v)]
        for u in range(-d, d + 1):
            val = int(img[cy + v, cx + u])
            m10 += u * val
            m01 += v * val
    return m10, m01


def hamming_matrix(q, t):
    qb = np.unpackbits(q, axis=1).astype(np.int32)
    tb = np.unpackbits(t, axis=1).astype(np.int32)
    return qb @ (1 - tb).T + (1 - qb) @ tb.T


def knn2(q, t, ratio=0.75):
    """(idx0, idx1, d0, d1, pass) by stable argsort: order (distance, index)."""
    d = hamming_matrix(q, t)
    order = np.argsort(d, axis=1, kind="stable")
    out = []
    for i in range(len(q)):
        idx = order[i, :2]
        ds = d[i, idx]
        i0 = int(idx[0]) if len(idx) > 0 else -1
        i1 = int(idx[1]) if len(idx) > 1 else -1
        d0 = int(ds[0]) if len(idx) > 0 else 0
        d1 = int(ds[1]) if len(idx) > 1 else 0
        ok = len(idx) == 2 and np.float32(d0) < np.float32(ratio) * np.float32(d1)
        out.append((i0, i1, d0, d1, int(ok)))
    return out


def brief256(blurred, cx, cy, angle_deg, pattern):
    """pattern: [256, 4] int.  float32 arithmetic with separate roundings."""
    theta = np.float32(angle_deg) * np.float32(np.pi / 180.0)
    a = np.float32(np.cos(np.float64(theta)))
    b = np.float32(np.sin(np.float64(theta)))
    pat = pattern.astype(np.float32)

    def sample(px, py):
        x = (px * a).astype(np.float32) - (py * b).astype(np.float32)
        y = (px * b).astype(np.float32) + (py * a).astype(np.float32)
        ix = np.rint(x.astype(np.float32)).astype(np.int64)
        iy = np.rint(y.astype(np.float32)).astype(np.int64)
        return blurred[cy + iy, cx + ix].astype(np.int64)

    t0 = sample(pat[:, 0], pat[:, 1])
    t1 = sample(pat[:, 2], pat[:, 3])
    bits = (t0 < t1).astype(np.uint8)
    return np.packbits(bits, bitorder="little")


def search_by_projection(kp_xy, frame_desc, kp_idx, uv, point_desc, point_ids, radius=32.0,
                         max_hamming=128):
    """Literal restatement of the nested loop with list erase."""
    rpoints = [tuple(p) for p in uv]
    ridx = list(range(len(uv)))
    out = []
    for idx in kp_idx:
        best = None
        mind = float("inf")
        for j, k in enumerate(ridx):
            dist = int(np.unpackbits(point_desc[k] ^ frame_desc[idx]).sum())
            dx = np.float32(rpoints[j][0]) - np.float32(kp_xy[idx][0])
            dy = np.float32(rpoints[j][1]) - np.float32(kp_xy[idx][1])
            e = np.float32(np.sqrt(np.float64(dx) ** 2 + np.float64(dy) ** 2))
            if dist <= max_hamming and e <= radius:
                if dist < mind:
                    mind = dist
                    best = j
        if best is None:
            out.append(-1)
        else:
            out.append(int(point_ids[ridx[best]]))
            del rpoints[best]
            del ridx[best]
    return np.array(out, dtype=np.int32)


def resize_linear_8u(src, fx, fy):
    """cv::resize(src, dst, Size(), fx, fy), INTER_LINEAR on 8-bit pixels, written
    from the definition: sample position (d + 0.5) / f - 0.5 in float32, weights
    rounded to 1/2048, horizontal pass kept at 11 extra bits, vertical pass
    ((b * (row >> 4)) >> 16 summed, + 2) >> 2."""
    src = np.asarray(src)
    if src.ndim == 2:
        src = src[:, :, None]
    sh, sw, cn = src.shape
    dw, dh = int(np.rint(sw * fx)), int(np.rint(sh * fy))

    def taps(n_dst, n_src, f):
        inv = 1.0 / f
        pos = ((np.arange(n_dst, dtype=np.float64) + 0.5) * inv - 0.5).astype(np.float32)
        i0 = np.floor(pos).astype(np.int64)
        fr = (pos - i0.astype(np.float32)).astype(np.float32)
        return i0, fr

    ix, fxr = taps(dw, sw, fx)
    fxr = np.where(ix < 0, np.float32(0), fxr)
    ix = np.maximum(ix, 0)
    over = ix >= sw - 1
    fxr = np.where(over, np.float32(0), fxr)
    ix = np.where(over, sw - 1, ix)
    ix1 = np.minimum(ix + 1, sw - 1)
    a1 = np.rint(fxr * np.float32(2048)).astype(np.int64)
    a0 = np.rint((np.float32(1) - fxr) * np.float32(2048)).astype(np.int64)
    iy, fyr = taps(dh, sh, fy)
    b1 = np.rint(fyr * np.float32(2048)).astype(np.int64)
    b0 = np.rint((np.float32(1) - fyr) * np.float32(2048)).astype(np.int64)
    y0 = np.clip(iy, 0, sh - 1)
    y1 = np.clip(iy + 1, 0, sh - 1)
    s = src.astype(np.int64)
    rows = s[:, ix, :] * a0[None, :, None] + s[:, ix1, :] * a1[None, :, None]  # [sh, dw, cn]
    r0, r1 = rows[y0], rows[y1]
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def invert_lu_f32(m):
    """Inverse of a square float32 matrix by Gaussian elimination with partial
    pivoting on [A | I], every operation rounded to float32, the pivot's
    reciprocal stored and used in the back-substitution (cv::invert, DECOMP_LU)."""
    f = np.float32
    n = len(m)
    A = [[f(v) for v in row] for row in np.asarray(m, np.float32)]
    B = [[f(1) if i == j else f(0) for j in range(n)] for i in range(n)]
    for i in range(n):
        k = max(range(i, n), key=lambda j: (abs(A[j][i]), -j))
        if abs(A[k][i]) < f(np.finfo(np.float32).eps) * f(10):
            return np.zeros((n, n), np.float32), False
        A[i], A[k] = A[k], A[i]
        B[i], B[k] = B[k], B[i]
        d = f(-1) / A[i][i]
        for j in range(i + 1, n):
            alpha = A[j][i] * d
            for c in range(i + 1, n):
                A[j][c] = A[j][c] + alpha * A[i][c]
            for c in range(n):
                B[j][c] = B[j][c] + alpha * B[i][c]
        A[i][i] = -d
    for i in range(n - 1, -1, -1):
        for j in range(n):
            s = B[i][j]
            for c in range(i + 1, n):
                s = s - A[i][c] * B[c][j]
            B[i][j] = s * A[i][i]
    return np.array(B, np.float32), True


def matmul_f32(a, b):
    """Small-matrix product with float32 products summed left to right in float32."""
    f = np.float32
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    out = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for i in range(a.shape[0]):
        for j in range(b.shape[1]):
            acc = f(a[i, 0]) * f(b[0, j])
            for k in range(1, a.shape[1]):
                acc = f(acc + f(a[i, k]) * f(b[k, j]))
            out[i, j] = acc
    return out


def reproject(pose, K, xyz, th, W, H):
    """The loop of Init.cc:203-218 written out literally."""
    f = np.float32
    kf, _ = invert_lu_f32(pose)
    K4 = np.eye(4, dtype=np.float32)
    K4[:3, :3] = K
    M = matmul_f32(K4, kf)
    uv, idx = [], []
    for i, (x, y, z) in enumerate(np.asarray(xyz, np.float32).reshape(-1, 3)):
        if z < f(th) or abs(float(f(z - f(th)))) > 20:
            continue
        with np.errstate(all="ignore"):
            pt = matmul_f32(M, np.array([[x], [y], [z], [1]], np.float32))[:, 0]
            pt = pt * f(1.0 / float(pt[3]))
            pt = pt * f(1.0 / float(pt[2]))
        if pt[0] <= 0 or pt[0] >= f(W) or pt[1] <= 0 or pt[1] >= f(H):
            continue
        uv.append((pt[0], pt[1]))
        idx.append(i)
    return np.array(uv, np.float32).reshape(-1, 2), np.array(idx, np.int32)


def svd4_last_row(A):
    """One-sided Jacobi on the columns of a 4x4 float32 matrix (norms and rotation
    parameters in float64, entries in float32, <= 30 sweeps, threshold
    2*FLT_EPSILON); returns the row of Vt that belongs to the smallest singular
    value after a descending selection sort (cv::SVD::compute, Init.cc:14-15)."""
    f32, f64 = np.float32, np.float64
    At = [[f32(A[k][i]) for k in range(4)] for i in range(4)]
    Vt = [[f32(1) if i == k else f32(0) for k in range(4)] for i in range(4)]
    W = []
    for i in range(4):
        sd = f64(0)
        for k in range(4):
            sd = sd + f64(At[i][k]) * f64(At[i][k])
        W.append(sd)
    eps = f64(np.finfo(np.float32).eps) * 2
    for _ in range(30):
        changed = False
        for i in range(3):
            for j in range(i + 1, 4):
                a, b, p = W[i], W[j], f64(0)
                for k in range(4):
                    p = p + f64(At[i][k]) * f64(At[j][k])
                if abs(p) <= eps * np.sqrt(a * b):
                    continue
                p = p * 2
                beta = a - b
                gamma = np.sqrt(p * p + beta * beta)
                if beta < 0:
                    delta = (gamma - beta) * 0.5
                    s_ = np.sqrt(delta / gamma)
                    c_ = p / (gamma * s_ * 2)
                else:
                    c_ = np.sqrt((gamma + beta) / (gamma * 2))
                    s_ = p / (gamma * c_ * 2)
                a = b = f64(0)
                for M in (At, Vt):
                    for k in range(4):
                        t0 = f32(c_ * f64(M[i][k]) + s_ * f64(M[j][k]))
                        t1 = f32(-s_ * f64(M[i][k]) + c_ * f64(M[j][k]))
                        M[i][k], M[j][k] = t0, t1
                        if M is At:
                            a = a + f64(t0) * f64(t0)
                            b = b + f64(t1) * f64(t1)
                W[i], W[j] = a, b
                changed = True
        if not changed:
            break
    for i in range(4):
        sd = f64(0)
        for k in range(4):
            sd = sd + f64(At[i][k]) * f64(At[i][k])
        W[i] = np.sqrt(sd)
    idx = [0, 1, 2, 3]
    for i in range(3):
        j = i
        for k in range(i + 1, 4):
            if W[idx[j]] < W[idx[k]]:
                j = k
        idx[i], idx[j] = idx[j], idx[i]
    return np.array(Vt[idx[3]], np.float32)


def triangulate(P1, P2, pts1, pts2):
    """Init::triangulate (Init.cc:5-31) for point pairs, written out literally."""
    f32 = np.float32
    P1 = np.asarray(P1, np.float32).reshape(3, 4)
    P2 = np.asarray(P2, np.float32).reshape(3, 4)
    out = []
    with np.errstate(all="ignore"):
        for (x1, y1), (x2, y2) in zip(np.asarray(pts1, np.float32), np.asarray(pts2, np.float32)):
            A = np.stack([f32(x1) * P1[2] - P1[0], f32(y1) * P1[2] - P1[1],
                          f32(x2) * P2[2] - P2[0], f32(y2) * P2[2] - P2[1]]).astype(np.float32)
            v = svd4_last_row(A)
            out.append(v[:3] * f32(1.0 / float(v[3])))
    return np.array(out, np.float32).reshape(-1, 3)
