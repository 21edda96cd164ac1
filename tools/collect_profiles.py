#!/usr/bin/env python3
"""Copy the judged rocprof evidence from gpurun_out/ (scratch) into profiles/<tag>/ (tracked):
trimmed kernel-stats CSV, per-kernel PMC means, the calibration of FETCH_SIZE/WRITE_SIZE on
kernels of known byte counts, and profiles/traffic_latest.json that bench.py reports as
roofline.traffic.

    python tools/collect_profiles.py r01 [--workload configs2]
"""
import argparse
import csv
import glob
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "gpurun_out")
GIB = float(1 << 30)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("tag")
    p.add_argument("--workload", default="configs2")
    p.add_argument("--kernel", default="k_step<true, true>")
    a = p.parse_args()
    out = os.path.join(ROOT, "profiles", a.tag)
    os.makedirs(out, exist_ok=True)
    # 1. rocprofv3 --kernel-trace --stats summary of `python3 bench.py` (names trimmed)
    ks = glob.glob(os.path.join(GO, f"prof_kt_{a.workload}", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        ks.sort(key=os.path.getmtime)  # gpurun merges every session's files into gpurun_out/: take the newest
        rows = list(csv.reader(open(ks[-1])))
        with open(os.path.join(out, f"kernel_stats_{a.workload}.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows:
                r[0] = r[0][:100]
                w.writerow(r)
    # 2. PMC passes (tools/pmc_probe.py): per-kernel means
    pmc = {}
    for name in ("prof_fetch", "prof_write", "prof_tcc", "prof_sq", "prof_mfma"):
        fp = os.path.join(GO, f"{name}_{a.workload}.summary.json")
        if os.path.exists(fp):
            for k, v in json.load(open(fp)).items():
                if "k_step" in k or "k_reset" in k or "k_logit" in k or "copyBuffer" in k or "FillFunctor" in k:
                    pmc.setdefault(k, {}).update({kk: vv for kk, vv in v.items() if not kk.endswith("_n")})
    # calibration from the raw CSVs: the 1 GiB copy / fill launches are the big ones
    calib = {}
    for name, ctr in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
        cc = sorted(glob.glob(os.path.join(GO, f"{name}_{a.workload}", "**", "*counter_collection.csv"),
                              recursive=True), key=os.path.getmtime)
        for f in cc[-1:]:
            per = {}
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    key = (r["Dispatch_Id"], r["Kernel_Name"][:60])
                    per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
            for (d, k), v in per.items():
                if "copyBuffer" in k and v > 1e5:
                    calib.setdefault(f"copy_1GiB_{ctr}_KB", []).append(v)
                if "FillFunctor" in k and ctr == "WRITE_SIZE" and 1.0e6 < v < 1.1e6:
                    calib.setdefault("fill_1GiB_WRITE_SIZE_KB", []).append(v)
    calib = {k: sum(v) / len(v) for k, v in calib.items()}
    kk = [k for k in pmc if a.kernel in k]
    res = {"workload": a.workload, "pmc": pmc, "calibration": calib}
    if kk and "FETCH_SIZE" in pmc[kk[0]] and "WRITE_SIZE" in pmc[kk[0]]:
        e = pmc[kk[0]]
        # MI355X_MICROARCH.md §HBM: counters are in KB (x1024); on gfx950 FETCH_SIZE reports 1/2 of the bytes of
        # 16-B-per-lane reads (confirmed here: a 1 GiB copy reads FETCH_SIZE = 524288 KB), WRITE_SIZE is exact
        rd_corr = GIB / (calib["copy_1GiB_FETCH_SIZE_KB"] * 1024) if "copy_1GiB_FETCH_SIZE_KB" in calib else 2.0
        wr_corr = GIB / (calib["copy_1GiB_WRITE_SIZE_KB"] * 1024) if "copy_1GiB_WRITE_SIZE_KB" in calib else 1.0
        rd = e["FETCH_SIZE"] * 1024 * rd_corr
        wr = e["WRITE_SIZE"] * 1024 * wr_corr
        res["traffic"] = {"read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
                          "read_correction": rd_corr, "write_correction": wr_corr,
                          "avg_us": e.get("avg_us")}
        tl = os.path.join(ROOT, "profiles", "traffic_latest.json")
        cur = json.load(open(tl)) if os.path.exists(tl) else {}
        cur[a.workload] = rd + wr
        json.dump(cur, open(tl, "w"), indent=1)
    json.dump(res, open(os.path.join(out, f"pmc_{a.workload}.json"), "w"), indent=1)
    logs = ["bench.log", "pytest_gpu.log", "smoke.log"] + [os.path.basename(x) for x in
                                                           glob.glob(os.path.join(GO, "bench_*.log"))]
    for f in logs:
        src = os.path.join(GO, f)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(out, f))
    print(json.dumps(res.get("traffic"), indent=1))
    print(json.dumps(calib, indent=1))


if __name__ == "__main__":
    main()
