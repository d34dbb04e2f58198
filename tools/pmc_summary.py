#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, the number of
dispatches and the mean of every counter per dispatch.
usage: pmc_summary.py <counter_collection.csv> [out.csv]"""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    if name.startswith("void "):
        name = name[5:]
    return name[:60]


def main():
    rows = defaultdict(lambda: defaultdict(list))
    with open(sys.argv[1], newline="") as f:
        for r in csv.DictReader(f):
            rows[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in rows.values() for c in k})
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "dispatches"] + counters)
    for k, cs in sorted(rows.items()):
        n = max(len(v) for v in cs.values())
        w.writerow([k, n] + [round(sum(cs[c]) / len(cs[c]), 1) if cs.get(c) else "" for c in counters])


if __name__ == "__main__":
    main()
