"""Host logic of sharded brute-force matching (relocalisation against a map
split over several GPUs).

Each rank matches all queries against its contiguous slice of the train rows
and emits, per query, its two best candidates as 64-bit keys
``distance << 32 | global_row``.  Integer order on keys is the (distance,
index) order of ``BFMatcher::knnMatch`` (Init.cc:131), so taking the two
smallest keys over all shards reproduces the single-device result exactly.
The keys travel with one all-gather (RCCL on GPUs; any torch.distributed
backend works).
"""
from __future__ import annotations

import numpy as np

from . import abi

KEY_NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


def shard_bounds(nrows: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of the train rows owned by `rank`."""
    per = (nrows + world - 1) // world
    lo = min(nrows, rank * per)
    hi = min(nrows, lo + per)
    return lo, hi


def frame_bounds(nframes: int, world: int, rank: int):
    """Contiguous slice [lo, hi) of a batch of `nframes` frames extracted by
    `rank`: frames are independent (Init.cc:124-130 handles one per iteration), so a
    batch splits over GPUs with no exchange; the first nframes % world ranks take
    one frame more.  Concatenating the ranks' results in rank order gives the
    single-device result."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank outside the world")
    base, extra = divmod(max(nframes, 0), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def match_rows_to_keys(rows: np.ndarray, train_base: int) -> np.ndarray:
    """[nq, 2] uint64 keys from match rows whose indices are shard-local."""
    nq = len(rows)
    keys = np.full((nq, 2), KEY_NONE, dtype=np.uint64)
    for col, (i, d) in enumerate((("idx0", "d0"), ("idx1", "d1"))):
        ok = rows[i] >= 0
        keys[ok, col] = (rows[d][ok].astype(np.uint64) << np.uint64(32)) | \
            (rows[i][ok].astype(np.int64) + train_base).astype(np.uint64)
    return keys


def merge_keys(keys_all: np.ndarray, ratio: float = 0.75) -> np.ndarray:
    """keys_all: [nshards, nq, 2] -> match rows (MATCH2_DTYPE) with global indices."""
    ns, nq, _ = keys_all.shape
    flat = np.transpose(keys_all, (1, 0, 2)).reshape(nq, ns * 2)
    best = np.sort(flat, axis=1)[:, :2]
    out = np.zeros(nq, abi.MATCH2_DTYPE)
    for col, (i, d) in enumerate((("idx0", "d0"), ("idx1", "d1"))):
        k = best[:, col]
        ok = k != KEY_NONE
        out[i] = np.where(ok, (k & np.uint64(0xFFFFFFFF)).astype(np.int64), -1).astype(np.int32)
        out[d] = np.where(ok, (k >> np.uint64(32)).astype(np.int64), 0).astype(np.uint16)
    out["nfound"] = (out["idx0"] >= 0).astype(np.uint8) + (out["idx1"] >= 0).astype(np.uint8)
    out["ratio_pass"] = ((out["nfound"] == 2) &
                         (out["d0"].astype(np.float32) < np.float32(ratio) * out["d1"].astype(np.float32)))
    return out
