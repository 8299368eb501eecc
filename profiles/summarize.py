#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace / stats / PMC passes) into a small text summary.

    python profiles/summarize.py gpurun_out/prof_kt gpurun_out/prof_fetch gpurun_out/prof_write [gpurun_out/prof_sq] > profiles/rNN_x.txt

FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced streams (MI355X_MICROARCH.md, HBM section) -- the "x2" column
applies that correction; gathers of 32..96-B records are not a wide stream, so the truth lies
between the raw and the corrected figure.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"sph::\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"void rocprim::.*?detail::", "rocprim::", name)
    name = re.sub(r"trampoline_kernel<rocprim::ROCPRIM_\d+_NS::detail::wrapped_", "", name)
    return name[:60]


def read(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    rows = []
    for f in fs:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    return rows


def main():
    dirs = sys.argv[1:]
    for d in dirs:
        stats = read(d, "*kernel_stats.csv")
        if stats:
            print(f"== kernel stats ({d}) ==")
            print(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
            for r in stats:
                print(f"{short(r['Name']):60s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} "
                      f"{float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['Percentage']):6.2f}")
        cc = read(d, "*counter_collection.csv")
        if cc:
            acc = defaultdict(lambda: defaultdict(list))
            for r in cc:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print(f"== counters per launch, mean over launches ({d}) ==")
            for k in sorted(acc):
                parts = []
                for c, v in sorted(acc[k].items()):
                    m = sum(v) / len(v)
                    if c in ("FETCH_SIZE", "WRITE_SIZE"):
                        parts.append(f"{c}={m / 1024:.2f} MB" + (f" (x2: {2 * m / 1024:.2f} MB)" if c == "FETCH_SIZE" else ""))
                    else:
                        parts.append(f"{c}={m:.4g}")
                print(f"{k:44s} n={len(next(iter(acc[k].values()))):3d}  " + "  ".join(parts))


if __name__ == "__main__":
    main()
