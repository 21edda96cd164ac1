#!/usr/bin/env python3
"""Gaps between consecutive step-kernel launches from a rocprofv3 kernel trace (CSV): start-to-start, duration and the idle
time in between, per launch -- where a short timed window (the driver's --steps 20 --warmup 5) loses its 3 us per step.
usage: python tools/exp_window_gaps.py <rocprof_out_dir> [first N launches, default 60]"""
import csv
import glob
import os
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-60:]))
rows.sort()
steps = [(s, e, k) for s, e, k in rows if "k_step" in k]
print(f"{len(steps)} step launches; first {n}: duration us / idle gap before it us / kernels of other kinds in that gap")
for i, (s, e, k) in enumerate(steps[:n]):
    gap = (s - steps[i - 1][1]) / 1e3 if i else 0.0
    others = [kk for ss, ee, kk in rows if i and steps[i - 1][1] <= ss < s and "k_step" not in kk]
    print(f"{i:3d} dur {(e - s) / 1e3:6.1f} gap {gap:7.1f} {','.join(o[-30:] for o in others)[:100]}")
dur = [(e - s) / 1e3 for s, e, _ in steps]
gaps = [(steps[i][0] - steps[i - 1][1]) / 1e3 for i in range(1, len(steps))]
for lo, hi in ((5, 25), (25, 150), (150, len(steps))):
    if hi > lo and len(dur) >= hi:
        g = gaps[lo:hi - 1]
        print(f"launches {lo}..{hi}: mean duration {sum(dur[lo:hi]) / (hi - lo):.2f} us, mean gap {sum(g) / max(len(g), 1):.2f} us, max gap {max(g):.1f} us")
