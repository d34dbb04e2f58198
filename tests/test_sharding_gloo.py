"""The N > 1 path on CPU: two gloo ranks each match the queries against their
shard (the CPU checker stands in for the per-shard kernel), all-gather the
top-2 keys, merge, and must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from slam_amd import sharding, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nq, nt, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    query, train = synth.descriptor_sets(nq, nt, seed=5, dup_percent=3)
    lo, hi = sharding.shard_bounds(nt, world, rank)
    rows = O.match_knn2(query, train[lo:hi])
    keys = sharding.match_rows_to_keys(rows, lo)
    mine = torch.from_numpy(keys.view(np.int64))
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    allk = torch.stack(gathered).numpy().view(np.uint64)
    merged = sharding.merge_keys(allk)
    if rank == 0:
        q.put(merged)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nq,nt", [(200, 3001), (7, 3)])
def test_two_rank_sharded_match_equals_single(nq, nt):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nq, nt, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    query, train = synth.descriptor_sets(nq, nt, seed=5, dup_percent=3)
    want = O.match_knn2(query, train)
    for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
        assert np.array_equal(merged[name], want[name]), name


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 1000, 1_000_000):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
