#!/usr/bin/env python3
"""Print the last call's launch chain from a rocprofv3 kernel trace CSV: per kernel
its duration and the gap since the previous kernel ended.
usage: trace_chain.py <kernel_trace.csv> <first kernel name prefix> [n kernels]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2]
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(first)]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
i0 = starts[-2] if len(starts) > 1 else starts[-1]
prev_end = None
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.2f} us  +{gap:6.2f} gap  {(e - s) / 1e3:7.2f} us  {r['Kernel_Name'][:50]}")
    prev_end = e
