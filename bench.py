#!/usr/bin/env python3
"""Benchmark of the feature front-end on MI355X.

Metric (BASELINE.json): frames/s of FAST+ORB extraction + brute-force Hamming
matching on 640x480 frames with 2000 features (8-level x1.2 pyramid, FAST
threshold 20), plus G Hamming comparisons/s.

One "step" = one batch of BATCH synthetic frames, already resident in HBM,
through the whole hot path: pyramid, FAST+NMS, selection, orientation, blur,
rBRIEF, then one k=2 ratio-test match per frame against its predecessor (the
per-frame loop of Init.cc:130-131).  Nothing is skipped or cached between
steps.  With --gpus N every rank runs the same per-GPU workload on its own
frames (frame batches shard with no collective): weak scaling, value = total
frames/s.

Also reported in the same JSON line:
  roofline      the dominant kernel against the HBM roofline (algorithmic bytes
                per launch / HIP-event time on the launch stream)
  cpu_baseline  the scalar CPU port of the same path, timed on this host
  hamming       G Hamming comparisons/s of the match kernel; with N > 1 also
                the sharded relocalisation case (2000 queries x 1M map rows,
                map split over the ranks, RCCL all-gather of per-shard top-2)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from slam_amd import abi, synth  # noqa: E402
from slam_amd.frontend import FrontEnd  # noqa: E402

W, H, NFEAT, THR = 640, 480, 2000, 20
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TOPS = 41.0  # measured: tools/ubench_valu.hip, v_xor + accumulating v_bcnt
MFMA_I8_PEAK_TOPS = 2048 * 1024 * 2.4e9 / 1e12  # dense int8 MFMA peak, 5033 TOP/s

STAGES = {"pyramid": 1, "fast_blur": 2, "select": 3, "orient": 4, "describe": 6, "match": 7}


def level_sizes(w, h, nlevels=8, sf=1.2):
    out = []
    for l in range(nlevels):
        scale = np.float32(np.float64(np.float32(sf)) ** l)
        inv = np.float32(1.0) / scale
        out.append((int(np.rint(np.float32(w) * inv)), int(np.rint(np.float32(h) * inv))))
    return out


def algorithmic_bytes_per_frame(w, h):
    """Compulsory u8 traffic of each streaming stage for one frame."""
    px = [a * b for a, b in level_sizes(w, h)]
    return {
        "pyramid": sum(px[:-1]) + sum(px[1:]),  # each resize reads level l-1, writes level l
        # the fused FAST + blur kernel reads every level once and writes the
        # blurred level once (as separate passes the two stages would move
        # 3 * sum(px): FAST read, blur read, blur write)
        "fast_blur": 2 * sum(px),
    }


def cpu_baseline(frames, budget_s=12.0):
    """The scalar CPU port (test infrastructure in oracle/) on a bounded sample
    of the same workload: extract + match against the previous frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    p = abi.reference_params(fast_threshold=THR, nfeatures=NFEAT)
    prev = None
    n = 0
    t0 = time.perf_counter()
    while True:
        img = frames[n % len(frames)]
        _, desc = O.extract(p, img)
        if prev is not None:
            O.match_knn2(prev, desc)
        else:
            O.match_knn2(desc, desc)
        prev = desc
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 4 * len(frames):
            break
    out = {"value": n / el, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{n} frames of the same 640x480 stream, extract + 1 match each, "
                     f"{el:.1f} s, scalar C -O3, 1 thread"}
    # matching alone: one thread, and query rows split over the host threads this
    # process may use (OpenCV's batchDistance runs under parallel_for_)
    from concurrent.futures import ThreadPoolExecutor
    q = synth.random_descriptors(2000, 3, 5)
    t = synth.random_descriptors(100000, 4, 6)
    t0 = time.perf_counter()
    O.match_knn2(q, t)
    one = time.perf_counter() - t0
    nthr = max(1, min(16, len(os.sched_getaffinity(0))))
    parts = [np.ascontiguousarray(q[ix]) for ix in np.array_split(np.arange(len(q)), nthr)]
    with ThreadPoolExecutor(nthr) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda a: O.match_knn2(a, t), parts))
        par = time.perf_counter() - t0
    cm = len(q) * len(t) / 1e9
    out["hamming_gcmps_1_thread"] = round(cm / one, 3)
    out["hamming_gcmps_threads"] = round(cm / par, 3)
    out["hamming_threads"] = nthr
    return out


def hbm_copy_ceiling(dev, nbytes=1 << 30, reps=10):
    """Device-to-device copy rate (read + write bytes per second): the practical
    ceiling an HBM-streaming kernel can be held against."""
    a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2 * nbytes / (ms * 1e-3) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed load before the warm-up steps so that the GPU clock has settled")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent frame streams per GPU, each with its own context and HIP stream")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-reloc", action="store_true", help="skip the 1M-row matching leg")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-rank code path (process group, barriers, all-gather) even "
                         "with one rank: a rehearsal of the N > 1 run on a one-GPU box")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", device_id=dev)

    B = args.batch
    # each rank gets its own frames (different pan offset / noise per rank)
    frames_np = synth.frame_stream(W, H, B, seed=1 + 17 * rank)
    frames = torch.from_numpy(frames_np).to(dev)
    S = max(1, args.streams)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    fes = [FrontEnd(mode=abi.MODE_ORB, fast_threshold=THR, nfeatures=NFEAT, max_width=W,
                    max_height=H, max_batch=B, frame_slots=2 * B + 2, device_id=local_rank,
                    stream=st.cuda_stream, map_capacity=0) for st in streams]
    fe, stream = fes[0], streams[0]
    cap = fe.max_keypoints
    match_outs = [torch.empty((B, cap, 16), dtype=torch.uint8, device=dev) for _ in range(S)]
    states = [{"last": None} for _ in range(S)]

    def step():
        hd = None
        for f, mo, state in zip(fes, match_outs, states):
            h = f.extract_resident(frames.data_ptr(), B, W, H, W, W * H, 1)
            first = int(h[0])
            # frame i is matched against frame i-1; frame 0 against the last frame
            # of the previous step (on the very first step: against itself)
            # (handles are consecutive, so from the second step on the previous
            # step's last frame is handle first - 1 and all B pairs go in one call)
            if state["last"] is not None and state["last"] == first - 1:
                f.match_resident_batch(first - 1, first, B, mo.data_ptr())
            else:
                f.match_resident_batch(first, first, 1, mo.data_ptr())
                if B > 1:
                    f.match_resident_batch(first, first + 1, B - 1, mo[1:].data_ptr())
            state["last"] = first + B - 1
            hd = hd if hd is not None else h
        return hd

    with torch.cuda.stream(stream):
        # Untimed, before the W warm-up steps: keep the GPU under this load for a
        # fraction of a second.  The chip raises its clock only after tens of
        # milliseconds of work, and a short run (few steps of ~1.3 ms) would
        # otherwise be timed partly at the idle clock.
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
            for _ in range(10):
                step()
            torch.cuda.synchronize(dev)
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hd = step()
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # keypoints per frame, for the record (and to derive comparison counts)
    counts = np.array([len(fe.frame_download(int(h)).keypoints) for h in hd[:4]])
    all_counts = []
    for h in hd:
        f = fe.frame_download(int(h))
        all_counts.append(len(f))
    all_counts = np.array(all_counts, dtype=np.int64)

    # per-stage kernel time inside the running pipeline: HIP events around the
    # stage on the stream the kernels are launched on, over extra (untimed) steps
    # of the same loop -- the same context the rocprofv3 averages come from.
    # (Launched back to back in isolation the dominant kernel runs up to 15 %
    # slower: kept as avg_launch_ms_back_to_back.)
    stage_ms = {}
    with torch.cuda.stream(stream):
        for name, sid in STAGES.items():
            fe.probe_stage(sid)
            for _ in range(8):
                step()
            stage_ms[name], nprobe = fe.probe_read()
            assert nprobe == 8, (name, nprobe)
        fe.probe_stage(0)
    isolated_ms = fe.time_stage(STAGES["fast_blur"], B, 10)
    alg = algorithmic_bytes_per_frame(W, H)
    hbm_stage = max(("pyramid", "fast_blur"), key=lambda k: stage_ms[k])
    dominant = max(stage_ms, key=lambda k: stage_ms[k])
    # the roofline object is given for the slowest HBM-streaming kernel
    # (the pyramid stage is seven launches; FAST + blur is one fused launch)
    roof_bytes = alg[hbm_stage] * B
    achieved = roof_bytes / (stage_ms[hbm_stage] * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # measured on launches of tj["frames_per_launch"] frames; traffic is
            # per frame-proportional, scaled to this run's batch
            traffic = tj.get(hbm_stage)
            if traffic is not None:
                traffic = int(traffic * B / tj.get("frames_per_launch", B))
        except Exception:
            traffic = None
    roofline = {"kernel": "k_" + hbm_stage if hbm_stage != "pyramid" else "k_resize (7 launches)",
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": roof_bytes,
                "avg_launch_ms": round(stage_ms[hbm_stage], 5),
                "avg_launch_ms_back_to_back": round(isolated_ms, 5),
                "measured_copy_ceiling_GBps": round(hbm_copy_ceiling(dev), 1)}

    # Hamming comparisons/s of the match kernel inside the frame pipeline
    # (a step matches B pairs in one call: frame i with frame i-1, frame 0 with
    # the last frame of the previous step)
    cmps = float((all_counts[:-1] * all_counts[1:]).sum() + all_counts[0] * all_counts[-1])
    gc = cmps / (stage_ms["match"] * 1e-3) / 1e9
    # The match kernel runs on the matrix cores: one 256-bit comparison = 256
    # int8 multiply-accumulates = 512 operations of v_mfma_i32_32x32x32_i8, whose
    # dense peak is 2048 op/clk/SIMD x 1024 SIMDs x 2.4 GHz.  (The integer-VALU
    # formulation, 16 lane-ops per comparison, tops out at 41.0 T lane-ops/s =
    # 2.56 T cmp/s: measured by tools/ubench_valu.hip.)
    hamming = {"frame_pairs_gcmps": round(gc, 2),
               "mfma_roofline": {"bound": "mfma", "ops_per_cmp": 512,
                                 "achieved": round(gc * 512 / 1e3, 1), "peak": MFMA_I8_PEAK_TOPS,
                                 "unit": "TOP/s", "frac": round(gc * 512 / 1e3 / MFMA_I8_PEAK_TOPS, 4)},
               "valu_formulation_ceiling_gcmps": round(VALU_PEAK_TOPS * 1e3 / 16, 1)}

    # relocalisation: 2000 queries against a 1M-row map sharded over the ranks
    if not args.no_reloc:
        try:
            NQ, NT = 2000, 1_000_000
            per = (NT + world - 1) // world
            lo, hi = rank * per, min(NT, (rank + 1) * per)
            q_np = synth.random_descriptors(NQ, 3, 5)
            shard = torch.from_numpy(synth.random_descriptors(hi - lo, 3 + 101 * rank, 6)).to(dev)
            dq = torch.from_numpy(q_np).to(dev)
            keys = torch.empty((NQ, 2), dtype=torch.int64, device=dev)
            gathered = torch.empty((world, NQ, 2), dtype=torch.int64, device=dev)
            out = torch.empty((NQ, 16), dtype=torch.uint8, device=dev)

            def reloc():
                fe.match_shard_keys(dq.data_ptr(), NQ, shard.data_ptr(), hi - lo, lo, keys.data_ptr())
                if multi:
                    dist.all_gather_into_tensor(gathered, keys)
                    fe.merge_shard_keys(gathered.data_ptr(), world, NQ, out.data_ptr())
                else:
                    fe.merge_shard_keys(keys.data_ptr(), 1, NQ, out.data_ptr())

            with torch.cuda.stream(stream):
                for _ in range(3):
                    reloc()
                torch.cuda.synchronize(dev)
                if multi:
                    dist.barrier()
                torch.cuda.synchronize(dev)
                r0 = time.perf_counter()
                R = 20
                for _ in range(R):
                    reloc()
                torch.cuda.synchronize(dev)
                if multi:
                    dist.barrier()
                rel = time.perf_counter() - r0
            if multi:
                t = torch.tensor([rel], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                rel = float(t.item())
            hamming["reloc_2000x1M_gcmps"] = round(NQ * NT * R / rel / 1e9, 2)
            if multi:
                # throughput with successive query sets pipelined: the all-gather of
                # set i (on RCCL's stream) overlaps the match kernel of set i+1
                keys2 = [keys, torch.empty_like(keys)]
                gath2 = [gathered, torch.empty_like(gathered)]

                def pipelined(count):
                    works = [None, None]
                    for it in range(count + 2):
                        b = it & 1
                        if works[b] is not None:
                            works[b].wait()
                            fe.merge_shard_keys(gath2[b].data_ptr(), world, NQ, out.data_ptr())
                            works[b] = None
                        if it < count:
                            fe.match_shard_keys(dq.data_ptr(), NQ, shard.data_ptr(), hi - lo, lo,
                                                keys2[b].data_ptr())
                            works[b] = dist.all_gather_into_tensor(gath2[b], keys2[b], async_op=True)

                with torch.cuda.stream(stream):
                    pipelined(4)
                    torch.cuda.synchronize(dev)
                    dist.barrier()
                    torch.cuda.synchronize(dev)
                    p0 = time.perf_counter()
                    pipelined(R)
                    torch.cuda.synchronize(dev)
                    dist.barrier()
                    pl = time.perf_counter() - p0
                t = torch.tensor([pl], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                hamming["reloc_pipelined_gcmps"] = round(NQ * NT * R / float(t.item()) / 1e9, 2)
            if world == 1 and not multi:
                # the same job on the vector-ALU popcount kernel (v_xor + v_bcnt), against
                # the measured integer-VALU ceiling: 16 lane-ops per comparison
                with FrontEnd(mode=abi.MODE_ORB, nfeatures=NFEAT, max_width=W, max_height=H, max_batch=1,
                              device_id=local_rank, stream=stream.cuda_stream, map_capacity=0,
                              match_kernel=abi.MATCH_POPCOUNT) as fp:
                    with torch.cuda.stream(stream):
                        for _ in range(3):
                            fp.match_shard_keys(dq.data_ptr(), NQ, shard.data_ptr(), hi - lo, lo, keys.data_ptr())
                        torch.cuda.synchronize(dev)
                        p0 = time.perf_counter()
                        for _ in range(10):
                            fp.match_shard_keys(dq.data_ptr(), NQ, shard.data_ptr(), hi - lo, lo, keys.data_ptr())
                        torch.cuda.synchronize(dev)
                        pc = NQ * NT * 10 / (time.perf_counter() - p0) / 1e9
                hamming["reloc_popcount_kernel_gcmps"] = round(pc, 2)
                hamming["popcount_kernel_valu_roofline"] = {
                    "bound": "integer VALU", "lane_ops_per_cmp": 16, "achieved_Tops": round(pc * 16 / 1e3, 2),
                    "peak_Tops": VALU_PEAK_TOPS, "frac": round(pc * 16 / 1e3 / VALU_PEAK_TOPS, 4)}
            hamming["reloc_mfma_frac"] = round(NQ * NT * R / rel / 1e12 * 512 / (MFMA_I8_PEAK_TOPS * world), 4)
            hamming["reloc_ms_per_query_set"] = round(rel / R * 1e3, 4)
            hamming["reloc_scaling"] = "strong (map rows split over ranks, RCCL all-gather of top-2 keys)"
        except Exception as e:  # the headline number must survive a failure of this leg
            hamming["reloc_error"] = repr(e)

    total_frames = B * S * args.steps * world
    value = total_frames / elapsed
    result = {
        "metric": "frames/sec FAST+ORB extract+match (640x480, 2k feats)",
        "value": round(value, 2),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "640x480 textured frame stream, FAST thr=20, 8-level x1.2 pyramid, "
                               "2000 ORB features, brute-force Hamming k=2 + ratio test per frame",
                   "frames_per_step": B * S, "frame_streams": S, "mode": "ORB (pyramid + quota + orientation)",
                   "keypoints_per_frame_mean": round(float(all_counts.mean()), 1),
                   "input": "gray u8 frames resident in HBM"},
        "roofline": roofline,
        "stage_ms_per_batch": {k: round(v, 5) for k, v in stage_ms.items()},
        "dominant_stage": dominant,
        "hamming": hamming,
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(frames_np)
        result["gpu_over_cpu"] = round(value / result["cpu_baseline"]["value"], 1)
    for f in fes:
        f.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
