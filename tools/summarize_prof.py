#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into a small markdown summary
(kernel-trace stats + per-dispatch PMC averages for the aggregation kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(pattern):
    return sorted(glob.glob(pattern, recursive=True))


def kernel_stats(d):
    rows = []
    for f in find(os.path.join(d, "trace", "**", "*kernel_stats.csv")):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def pmc_table(d):
    """-> {(kernel, counter): [values per dispatch]}"""
    out = defaultdict(list)
    for f in find(os.path.join(d, "**", "*counter_collection.csv")):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "?")
                out[(os.path.basename(os.path.dirname(os.path.dirname(f))) if False else k,
                     r.get("Counter_Name", "?"))].append(float(r.get("Counter_Value", "nan")))
    return out


def main():
    d = sys.argv[1]
    print(f"# rocprofv3 summary: {os.path.basename(d)}\n")
    print("## kernel-trace --stats (bench.py --steps 20 --warmup 5)\n")
    print("| kernel | calls | total ns | avg ns | min ns | max ns | % |")
    print("|---|---|---|---|---|---|---|")
    for r in kernel_stats(d)[:12]:
        print("| {} | {} | {} | {} | {} | {} | {} |".format(
            r.get("Name", "?")[:70], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"),
            r.get("MinNs"), r.get("MaxNs"), r.get("Percentage")))
    print("\n## PMC (per dispatch averages; separate passes, bench.py --steps 3 --warmup 4)\n")
    print("| kernel | counter | dispatches | mean | min | max |")
    print("|---|---|---|---|---|---|")
    t = pmc_table(d)
    for (k, c), v in sorted(t.items()):
        if "sweep_kernel" in k or "stream_kernel" in k:
            v = v[-3:]          # the timed steps: a drop-in graph's first two calls (counting pass, preparation) take another path
        print("| {} | {} | {} | {:.6g} | {:.6g} | {:.6g} |".format(k[:60], c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
