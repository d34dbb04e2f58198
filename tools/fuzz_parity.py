#!/usr/bin/env python3
"""Longer randomised parity run of the HIP path against the CPU checker (not part
of the test suite; `python tools/fuzz_parity.py [cases] [seed]` on a GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402


def params_of(fe):
    p = abi.Params()
    for name, _ in abi.Params._fields_:
        setattr(p, name, getattr(fe.params, name))
    return p


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    bad = 0
    for case in range(cases):
        w, h = int(rng.integers(64, 1300)), int(rng.integers(64, 760))
        if case % 2:  # widths whose rows already have level 0's layout (frames used in place)
            w = 64 * int(rng.integers(1, 21))
        kw = dict(mode=int(rng.integers(0, 2)), fast_threshold=int(rng.integers(3, 80)),
                  nfeatures=int(rng.integers(10, 6000)), nlevels=int(rng.integers(1, 9)),
                  scale_factor=float(rng.choice([1.05, 1.1, 1.2, 1.25, 1.5, 1.7, 2.0])),
                  edge_threshold=int(rng.integers(22, 48)))
        img = synth.textured_canvas(w, h, 1000 + case)
        with FrontEnd(max_width=w, max_height=h, map_capacity=4096,
                      match_kernel=int(rng.integers(0, 3)), **kw) as fe:
            f = fe.extract_keypoints(img)
            kps, desc = O.extract(params_of(fe), img)
            ok = len(f) == len(kps) and all(np.array_equal(f.keypoints[n], kps[n]) for n in kps.dtype.names) \
                and np.array_equal(f.descriptors, desc)
            # a batch through the host-pointer call and the resident call, in the same context
            nb = int(rng.choice([2, 3, 9])) if w * h <= 400 * 300 else 2
            if case % 4 < 2:
                import torch
                frames = [img] + [synth.textured_canvas(w, h, 5000 + case * 16 + i) for i in range(1, nb)]
                want_b = [O.extract(params_of(fe), fr) for fr in frames]
                with FrontEnd(max_width=w, max_height=h, map_capacity=0, max_batch=nb, **kw) as fb:
                    got_b = fb.extract_batch(frames)
                    d = torch.from_numpy(np.ascontiguousarray(np.stack(frames))).cuda()
                    torch.cuda.synchronize()
                    got_r = [fb.frame_download(int(hd)) for hd in fb.extract_resident(d.data_ptr(), nb, w, h, w, w * h, 1)]
                for g, (kb, db) in list(zip(got_b, want_b)) + list(zip(got_r, want_b)):
                    ok = ok and len(g) == len(kb) and all(np.array_equal(g.keypoints[n], kb[n]) for n in kb.dtype.names) \
                        and np.array_equal(g.descriptors, db)
            nq, nt = int(rng.integers(0, 3000)), int(rng.integers(0, 5000))
            q, t = synth.descriptor_sets(max(nq, 1), max(nt, 1), seed=case, dup_percent=int(rng.integers(0, 20)))
            q, t = q[:nq], t[:nt]
            okm = np.array_equal(fe.knn_match(q, t), O.match_knn2(q, t)) if nq else True
            oks = True
            if len(f) > 20:
                npts = int(rng.integers(1, 4000))
                muv, mdesc = synth.local_map(f.descriptors, f.pt, npts, w, h, seed=case)
                fe.map_put(np.arange(npts), mdesc)
                kp_idx = rng.permutation(len(f))[:int(rng.integers(1, len(f)))].astype(np.int32)
                radius = float(rng.choice([32.0, 8.0, 100.0]))
                got = fe.search_by_projection(f, kp_idx, muv, np.arange(npts, dtype=np.int32), radius=radius)
                want = O.search_by_projection(f.pt, f.descriptors, kp_idx, muv, mdesc,
                                              np.arange(npts, dtype=np.int32), radius=radius)
                oks = np.array_equal(got, want)
            # frame ingest with resize, reprojection, triangulation
            okx = True
            if case % 3 == 0 and w >= 96 and h >= 96:
                bgr = np.ascontiguousarray(np.stack([img, np.roll(img, 1, 1), np.roll(img, 2, 0)], axis=2))
                fx, fy = float(rng.choice([0.99, 0.9, 0.75, 1.0])), float(rng.choice([0.99, 0.8, 1.0]))
                with FrontEnd(max_width=w, max_height=h, map_capacity=8192, **kw) as fe2:
                    f2, resized = fe2.extract_scaled_bgr(bgr, fx, fy)
                    want_r = O.resize_linear_8u(bgr, fx, fy)
                    k2, d2 = O.extract(params_of(fe2), want_r)
                    okx = np.array_equal(resized, want_r) and len(f2) == len(k2) and \
                        all(np.array_equal(f2.keypoints[n], k2[n]) for n in k2.dtype.names) and \
                        np.array_equal(f2.descriptors, d2)
                    npts = int(rng.integers(1, 8000))
                    pose, K, xyz, th = synth.camera_scene(npts, w, h, case)
                    fe2.map_put_xyz(np.arange(npts), xyz)
                    uv, idx = fe2.reproject(pose, K, th, w, h, npts)
                    wuv, widx = O.reproject(pose, K, xyz, th, w, h)
                    okx = okx and np.array_equal(idx, widx) and np.array_equal(uv.view(np.uint32), wuv.view(np.uint32))
                    P1, P2, p1, p2 = synth.two_view_scene(int(rng.integers(1, 3000)), w, h, case)
                    okx = okx and np.array_equal(fe2.triangulate(P1, P2, p1, p2).view(np.uint32),
                                                 O.triangulate(P1, P2, p1, p2).view(np.uint32))
        if not (ok and okm and oks and okx):
            bad += 1
            print("MISMATCH", case, w, h, kw, "extract", ok, "match", okm, (nq, nt), "search", oks,
                  "ingest/reproject/triangulate", okx, flush=True)
        elif case % 10 == 0:
            print("case", case, "ok", len(kps), "keypoints", flush=True)
    print("done:", cases, "cases,", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
