#!/usr/bin/env python3
"""Turn the counter summaries of tools/collect_profiles.sh into profiles/pmc_traffic.json,
the file bench.py reads its HBM-traffic and instruction-count figures from (counters
cannot be collected inside the bench process), and record what they were measured on:
the commit of the tree and the SHA-256 of the library binary that ran.  bench.py
compares that hash with the library it loads and says so on its JSON line, so a
reader of the line alone can tell whether the counters belong to the kernels timed.

usage (here, after the GPU run, from the repository root, with the tree that ran):
    python tools/update_pmc_traffic.py gpurun_out/prof_<tag> [<name prefix under profiles/>]
"""
import csv
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def row_of(path, kernel):
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["kernel"].startswith(kernel):
                return r
    raise SystemExit(f"{path}: no row for {kernel}")


def main():
    src = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else None
    fetch = row_of(os.path.join(src, "pmc_fetch_size_summary.csv"), "k_fast_blur")
    write = row_of(os.path.join(src, "pmc_write_size_summary.csv"), "k_fast_blur")
    sq = row_of(os.path.join(src, "pmc_sq_summary.csv"), "k_fast_blur")
    frames = 256
    try:  # frames per launch of the profiled bench runs (its default batch)
        frames = int(json.load(open(os.path.join(src, "bench.json")))["config"]["frames_per_step"])
    except Exception:
        pass
    lv = [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    alg = sum(w * h for w, h in lv) * frames
    fetch_kb, write_kb = float(fetch["FETCH_SIZE"]), float(write["WRITE_SIZE"])
    read_b, write_b = int(fetch_kb * 1024 * 2), int(write_kb * 1024)
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "slam_amd", "include"],
                                capture_output=True, text=True).stdout.strip())
    out = {
        "_note": "HBM-side bytes and instruction counts per k_fast_blur launch from rocprofv3 --pmc (separate "
                 "passes for FETCH_SIZE, WRITE_SIZE and the SQ counters; tools/collect_profiles.sh: bench.py "
                 "--no-cpu --no-reloc --no-legs --steps 3 --warmup 1 --streams 1, frames_per_launch frames of "
                 "640x480 per launch). Counter unit KB. FETCH_SIZE is doubled: on gfx950 it tallies 128-B requests at 64 B "
                 "(calibrated on a plain copy kernel: reads 78,643,200 B, FETCH_SIZE 38,411.8 KB = 0.500x; "
                 "WRITE_SIZE read exactly).",
        "measured_on": {"commit": head + ("+uncommitted kernel changes" if dirty else ""),
                        "libslamfe_sha256": sha256(os.path.join(ROOT, "slam_amd", "libslamfe.so")),
                        "summaries": f"profiles/{prefix}_pmc_*_summary.csv" if prefix else src},
        "fast_blur": read_b + write_b,
        "fast_blur_detail": {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "read_bytes_corrected": read_b,
                             "write_bytes": write_b, "algorithmic_read": alg, "algorithmic_write": alg},
        "frames_per_launch": frames,
        "fast_blur_wave_insts": int(float(sq["SQ_INSTS_VALU"])),
        "fast_blur_mfma_insts": int(float(sq["SQ_INSTS_MFMA"])),
        "_note_wave_insts": "SQ_INSTS_VALU / SQ_INSTS_MFMA per k_fast_blur launch of frames_per_launch frames",
    }
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    if prefix:
        for name in ("pmc_fetch_size_summary.csv", "pmc_write_size_summary.csv", "pmc_sq_summary.csv",
                     "kernel_stats.csv", "bench.json", "bench_under_rocprof.json"):
            p = os.path.join(src, name)
            if os.path.exists(p):
                shutil.copy(p, os.path.join(ROOT, "profiles", f"{prefix}_{name}"))
    print(json.dumps(out["measured_on"]), out["fast_blur"], out["fast_blur_wave_insts"])


if __name__ == "__main__":
    main()
