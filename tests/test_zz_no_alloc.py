"""The entry point that allocates nothing (slamfe_match_shard_keys_static): its
keys must equal those of slamfe_match_shard_keys bit for bit, on two calls in a
row, the second with a smaller input (the fixed scratch array keeps the first
call's contents), for both match kernels; jobs beyond the array are refused.
Runs last (file name) and also passes alone in a fresh process.
"""
import numpy as np
import pytest

from slam_amd import abi, synth

pytestmark = pytest.mark.gpu


def _keys(fe, fn, dq, nq, dt, nt, base):
    import torch
    keys = torch.full((max(nq, 1), 2), -7, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()  # the fill runs on torch's stream, the library on its own
    fn(dq.data_ptr(), nq, dt.data_ptr(), nt, base, keys.data_ptr())
    fe.sync()
    return keys.cpu().numpy()[:nq]


@pytest.mark.parametrize("kernel", [abi.MATCH_AUTO, abi.MATCH_POPCOUNT, abi.MATCH_MATRIX])
def test_static_scratch_keys_equal_growable_buffer_keys(kernel):
    import torch
    from slam_amd.frontend import FrontEnd, SlamfeError
    # first a large job (many chunks: the wave-per-query merge), then a smaller one
    # whose partials cover only the front of the array
    # (the popcount kernel cuts the train set into chunks of 256 rows, not thousands: its
    # large job is smaller so that the partials still fit the 4 MiB array)
    big = 60_000 if kernel == abi.MATCH_POPCOUNT else 300_000
    jobs = [(2000, big, 1_000_000), (300, 5000, 17), (129, 33, 0), (40, 3000, 5)]
    nq_max = max(j[0] for j in jobs)
    nt_max = max(j[1] for j in jobs)
    q, t = synth.descriptor_sets(nq_max, nt_max, seed=31, dup_percent=2)
    dq = torch.from_numpy(q).cuda()
    dt = torch.from_numpy(t).cuda()
    with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0,
                  match_kernel=kernel) as fe:
        for nq, nt, base in jobs:
            got = _keys(fe, fe.match_shard_keys_static, dq, nq, dt, nt, base)
            want = _keys(fe, fe.match_shard_keys, dq, nq, dt, nt, base)
            assert np.array_equal(got, want), (kernel, nq, nt)
            assert (got[:, 0] >> 32).max() <= 256 and (got[:, 0] & 0xffffffff).min() >= base
        # a second context on the same GPU shares the array and gives the same keys
        with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0,
                      match_kernel=kernel) as fe2:
            nq, nt, base = jobs[1]
            assert np.array_equal(_keys(fe2, fe2.match_shard_keys_static, dq, nq, dt, nt, base),
                                  _keys(fe, fe.match_shard_keys, dq, nq, dt, nt, base))
        # 2000 x 1M, the relocalisation job, fits; a job whose partials exceed 4 MiB is
        # refused before anything is launched.  (Every pointer handed over is valid for the
        # sizes named all the same: a test must not depend on a refusal to stay in bounds.)
        nq_big = 300_000 if kernel == abi.MATCH_POPCOUNT else 600_000
        big_q = torch.zeros((nq_big, 32), dtype=torch.uint8, device="cuda")
        big_keys = torch.zeros((nq_big, 2), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        with pytest.raises(SlamfeError) as e:
            fe.match_shard_keys_static(big_q.data_ptr(), nq_big, dt.data_ptr(), nt_max, 0, big_keys.data_ptr())
        assert e.value.code == abi.SLAMFE_E_CAPACITY and "static scratch" in str(e.value)
        del big_q, big_keys
        # empty query set: nothing to do
        fe.match_shard_keys_static(0, 0, dt.data_ptr(), 10, 0, dq.data_ptr())


def test_static_entry_point_allocates_no_device_memory():
    """The property the entry point exists for: device memory in use does not change over
    its calls, on a context whose growable partial-result buffer has never been sized (a
    call routed through that buffer would have to allocate here).  The jobs: the
    relocalisation one, 2000 x 1,000,000, whose partials (3.9 MiB) nearly fill the 4 MiB
    array; a smaller one after it; and 170,000 queries x 10,000 rows (3 chunks: 3.9 MiB).
    The keys are compared with the growable entry point's afterwards."""
    import torch
    from slam_amd.frontend import FrontEnd
    nq, nt = 2000, 1_000_000
    nq_wide, nt_wide = 170_000, 10_000
    q, t = synth.descriptor_sets(nq, nt, seed=53, dup_percent=1)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    wide_q = torch.from_numpy(synth.random_descriptors(nq_wide, 57, 2)).cuda()
    keys = torch.full((nq, 2), -7, dtype=torch.int64, device="cuda")
    keys_small = torch.full((nq, 2), -7, dtype=torch.int64, device="cuda")
    keys_wide = torch.full((nq_wide, 2), -7, dtype=torch.int64, device="cuda")
    keys_wide_want = torch.full((nq_wide, 2), -7, dtype=torch.int64, device="cuda")
    with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0,
                  max_keypoints=64) as fe:
        torch.cuda.synchronize()
        free_before, _ = torch.cuda.mem_get_info()
        fe.match_shard_keys_static(dq.data_ptr(), nq, dt.data_ptr(), nt, 0, keys.data_ptr())
        fe.match_shard_keys_static(dq.data_ptr(), 333, dt.data_ptr(), 70_001, 9, keys_small.data_ptr())
        fe.match_shard_keys_static(wide_q.data_ptr(), nq_wide, dt.data_ptr(), nt_wide, 0, keys_wide.data_ptr())
        fe.sync()
        free_after, _ = torch.cuda.mem_get_info()
        assert free_after == free_before, (free_before, free_after)
        got, got_small = keys.cpu().numpy(), keys_small.cpu().numpy()[:333]
        assert np.array_equal(got, _keys(fe, fe.match_shard_keys, dq, nq, dt, nt, 0))
        assert np.array_equal(got_small, _keys(fe, fe.match_shard_keys, dq, 333, dt, 70_001, 9))
        fe.match_shard_keys(wide_q.data_ptr(), nq_wide, dt.data_ptr(), nt_wide, 0, keys_wide_want.data_ptr())
        fe.sync()
        assert torch.equal(keys_wide, keys_wide_want)
        assert int((keys_wide[:, 0] >> 32).max()) <= 256 and int((keys_wide[:, 1] & 0xffffffff).max()) < nt_wide


def test_static_scratch_against_the_cpu_checker():
    import torch
    import oracle_lib as O
    from slam_amd.frontend import FrontEnd
    nq, nt = 257, 4099
    q, t = synth.descriptor_sets(nq, nt, seed=41, dup_percent=3)
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    out = torch.zeros(nq * 16, dtype=torch.uint8, device="cuda")
    with FrontEnd(mode=abi.MODE_ORB, max_width=64, max_height=64, max_batch=1, map_capacity=0) as fe:
        for n in (nq, 100):  # two calls in a row, the second smaller
            keys = torch.empty((n, 2), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            fe.match_shard_keys_static(dq.data_ptr(), n, dt.data_ptr(), nt, 0, keys.data_ptr())
            fe.merge_shard_keys(keys.data_ptr(), 1, n, out.data_ptr())
            fe.sync()
            got = out.cpu().numpy().view(abi.MATCH2_DTYPE)[:n]
            want = O.match_knn2(q[:n], t)
            for name in ("idx0", "idx1", "d0", "d1", "ratio_pass", "nfound"):
                assert np.array_equal(got[name], want[name]), (n, name)
