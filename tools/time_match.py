#!/usr/bin/env python3
"""Time of the brute-force k=2 match alone (descriptors resident in HBM): the
relocalisation shape (2000 x 1M) and the frame-pair shape (256 pairs of 2000 x 2000
as one call).  Prints T comparisons/s for the matrix-core and the popcount kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402

NQ, NT = 2000, int(os.environ.get("NT", 1_000_000))
dev = torch.device("cuda:0")
q = torch.from_numpy(synth.random_descriptors(NQ, 3, 5)).to(dev)
t = torch.from_numpy(synth.random_descriptors(NT, 3, 6)).to(dev)
keys = torch.empty((NQ, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for name, kern in (("matrix", abi.MATCH_MATRIX), ("popcount", abi.MATCH_POPCOUNT)):
    with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0,
                  match_kernel=kern) as fe:
        for _ in range(int(os.environ.get('WARM', 100))):
            fe.match_shard_keys(q.data_ptr(), NQ, t.data_ptr(), NT, 0, keys.data_ptr())
        fe.sync()
        R = int(os.environ.get('REPS', 100))
        t0 = time.perf_counter()
        for _ in range(R):
            fe.match_shard_keys(q.data_ptr(), NQ, t.data_ptr(), NT, 0, keys.data_ptr())
        fe.sync()
        el = (time.perf_counter() - t0) / R
        print(f"{name:9s} {NQ} x {NT}: {el * 1e3:.3f} ms, {NQ * NT / el / 1e12:.3f} T cmp/s", flush=True)
