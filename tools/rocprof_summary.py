#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel trace and/or counter collection) per kernel.

    python tools/rocprof_summary.py <rocprof_out_dir> [--json out.json]

Prints one line per kernel: launches, avg/min/max duration (us), and the mean of every
collected counter per launch. Kernel names are shortened to the part before '('.
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def short(name):
    name = re.sub(r"\s*\[clone.*$", "", name)
    name = name.split("(")[0]
    return name[-90:]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("dir")
    p.add_argument("--json", default=None)
    a = p.parse_args()
    dur = defaultdict(list)
    ctr = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for f in glob.glob(os.path.join(a.dir, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(float)
        for r in csv.DictReader(open(f)):
            key = (r.get("Dispatch_Id"), short(r["Kernel_Name"]), r["Counter_Name"])
            per[key] += float(r["Counter_Value"])
        for (d, k, c), v in per.items():
            ctr[k][c].append(v)
    out = {}
    names = sorted(set(dur) | set(ctr), key=lambda k: -sum(dur.get(k, [0])))
    for k in names:
        e = {}
        if k in dur:
            d = dur[k]
            e.update(launches=len(d), avg_us=sum(d) / len(d), min_us=min(d), max_us=max(d), total_us=sum(d))
        for c, v in ctr.get(k, {}).items():
            e[c] = sum(v) / len(v)
            e[c + "_n"] = len(v)
        out[k] = e
        print(k, json.dumps(e))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
