#!/usr/bin/env python3
"""In-kernel timing of k_match_mfma from a diagnostic build (make EXTRA=-DSLAMFE_STAMPS
OUT=../libslamfe_stamps.so; run with SLAMFE_LIB pointing at it): cycles per stage
of the stage loop, share of prologue and epilogue, and the shader clock the chip
held, for the relocalisation shape and the frame-pair shape."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402

dev = torch.device("cuda:0")


def report(fe, what):
    out = np.zeros(64 * 16, np.uint64)
    fe._lib.slamfe_debug_match_stamps.argtypes = [C.c_void_p, C.c_int]
    fe._lib.slamfe_debug_match_stamps(out.ctypes.data, out.size)
    st = out.reshape(64, 16).astype(np.int64)
    st = st[st[:, 0] != 0]
    cyc = st[:, 0:8:2]
    wall = st[:, 1:8:2] * 10  # ns (100 MHz counter)
    ns = st[:, 8]
    loop = cyc[:, 2] - cyc[:, 1]
    tot = cyc[:, 3] - cyc[:, 0]
    clk = tot / np.maximum(wall[:, 3] - wall[:, 0], 1)
    print(f"{what}: {len(st)} workgroups sampled, stages {np.median(ns):.0f}; cycles/stage "
          f"{np.median(loop / np.maximum(ns, 1)):.0f} (64 MFMAs/SIMD = 2048 at 2 waves); prologue "
          f"{np.median(cyc[:, 1] - cyc[:, 0]):.0f} loop {np.median(loop):.0f} epilogue "
          f"{np.median(cyc[:, 3] - cyc[:, 2]):.0f} cycles; clock {np.median(clk):.2f} GHz", flush=True)
    if os.environ.get("VERBOSE"):
        t0 = wall[:, 0].min()
        for i in np.argsort(wall[:, 0]):
            print(f"   start {(wall[i, 0] - t0) / 1e3:8.1f} us  end {(wall[i, 3] - t0) / 1e3:8.1f} us  "
                  f"cycles/stage {loop[i] / max(ns[i], 1):.0f}")


def where(fe, nwg):
    out = np.zeros(4096 * 4, np.uint64)
    fe._lib.slamfe_debug_match_where.argtypes = [C.c_void_p, C.c_int]
    fe._lib.slamfe_debug_match_where(out.ctypes.data, out.size)
    w = out.reshape(4096, 4)[:nwg].astype(np.int64)
    t0 = w[:, 0].min()
    start, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0  # us
    hw, xcc = w[:, 2], w[:, 3] & 0xf
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    print(f"   {nwg} workgroups on {len(np.unique(cu))} distinct CUs; kernel span {end.max():.1f} us")
    first = start < 5.0
    cnt = np.bincount(np.unique(cu, return_inverse=True)[1][first])
    print(f"   started in the first 5 us: {first.sum()}; CUs with 1 / 2 / 3+ of them: "
          f"{(cnt == 1).sum()} / {(cnt == 2).sum()} / {(cnt >= 3).sum()}")
    dur = end - start
    print("   duration histogram (us):", np.histogram(dur, bins=8)[0].tolist(),
          [round(x) for x in np.histogram(dur, bins=8)[1].tolist()])
    per_xcc = [int(((xcc == x) & first).sum()) for x in range(8)]
    print("   first-round workgroups per XCC:", per_xcc, " total per XCC:", [int((xcc == x).sum()) for x in range(8)])


NQ, NT = 2000, 1_000_000
q = torch.from_numpy(synth.random_descriptors(NQ, 3, 5)).to(dev)
t = torch.from_numpy(synth.random_descriptors(NT, 3, 6)).to(dev)
keys = torch.empty((NQ, 2), dtype=torch.int64, device=dev)
with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0,
              match_kernel=abi.MATCH_MATRIX) as fe:
    import time
    for _ in range(100):
        fe.match_shard_keys(q.data_ptr(), NQ, t.data_ptr(), NT, 0, keys.data_ptr())
    fe.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        fe.match_shard_keys(q.data_ptr(), NQ, t.data_ptr(), NT, 0, keys.data_ptr())
    fe.sync()
    print(f"host time per call (match + merge kernels): {(time.perf_counter() - t0) * 1e4:.1f} us")
    report(fe, "2000 x 1M")
    where(fe, 8 * ((NT + 8127) // 8128))
    NT2 = 2024
    for _ in range(50):
        fe.match_shard_keys(q.data_ptr(), NQ, t.data_ptr(), NT2, 0, keys.data_ptr())
    fe.sync()
    report(fe, "2000 x 2024 (one pair, chip mostly idle)")
