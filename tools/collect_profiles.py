#!/usr/bin/env python3
"""Copy the judged rocprof evidence from gpurun_out/ (scratch) into profiles/<round>/ (tracked):
trimmed kernel-stats CSV of `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`, per-kernel PMC means of the
tools/pmc_probe.py passes, the in-process calibration of FETCH_SIZE / WRITE_SIZE on kernels of known byte counts, and
profiles/traffic_latest.json, which bench.py reports as roofline.traffic when the kernel sources still hash to the
`src_sha` recorded here.

    python tools/collect_profiles.py r02 configs2 configs2_sorted configs3_pm
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "gpurun_out")
GIB = float(1 << 30)
KEEP = ("k_step", "k_reset", "k_posterior", "k_pm_", "k_rollout", "k_group", "copyBuffer", "FillFunctor")


def newest(pattern):
    f = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def collect(tag, out, rnd):
    # 1. rocprofv3 --kernel-trace --stats summary of bench.py (names trimmed)
    ks = newest(os.path.join(GO, f"prof_kt_{tag}", "**", "*kernel_stats.csv"))
    if ks:
        rows = list(csv.reader(open(ks)))
        with open(os.path.join(out, f"kernel_stats_{tag}.csv"), "w", newline="") as f:
            w = csv.writer(f)
            for r in rows:
                r[0] = r[0][:100]
                w.writerow(r)
    log = os.path.join(GO, f"prof_kt_{tag}.log")
    if os.path.exists(log):
        line = [ln for ln in open(log).read().splitlines() if ln.startswith('{"metric"')]
        if line:
            open(os.path.join(out, f"bench_under_rocprof_{tag}.json"), "w").write(line[-1] + "\n")
    # 2. PMC passes (tools/pmc_probe.py): per-kernel means
    # Every pass records the hash of the kernel sources it ran on (<pass>_<tag>.probe.json, written next to its
    # summary by tools/gpu_session.sh). Passes are merged only when they ran on the SAME sources as the newest one:
    # a summary left in gpurun_out/ by an older build would otherwise smuggle kernels that no longer exist into
    # pmc_<tag>.json.
    probe = {}
    pj = os.path.join(GO, f"pmc_probe_{tag}.json")
    if os.path.exists(pj):
        probe = json.load(open(pj))
    pmc, skipped = {}, []
    for name in ("prof_fetch", "prof_write", "prof_tcc", "prof_sq", "prof_sq2", "prof_mfma"):
        fp = os.path.join(GO, f"{name}_{tag}.summary.json")
        if not os.path.exists(fp):
            continue
        pp = os.path.join(GO, f"{name}_{tag}.probe.json")
        sha = json.load(open(pp)).get("src_sha") if os.path.exists(pp) else None
        if sha != probe.get("src_sha"):
            skipped.append({"pass": name, "src_sha": sha})
            print(f"{tag}: pass {name} ran on kernel sources {sha}, not {probe.get('src_sha')}: left out")
            continue
        for k, v in json.load(open(fp)).items():
            if any(x in k for x in KEEP):
                pmc.setdefault(k, {}).update({kk: vv for kk, vv in v.items() if not kk.endswith("_n")})
    # calibration from the raw CSVs: the 1 GiB copy launches of the probe
    calib = {}
    for name, ctr in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
        f = newest(os.path.join(GO, f"{name}_{tag}", "**", "*counter_collection.csv"))
        if f:
            per = {}
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == ctr:
                    key = (r["Dispatch_Id"], r["Kernel_Name"][:60])
                    per[key] = per.get(key, 0.0) + float(r["Counter_Value"])
            vals = [v for (d, k), v in per.items() if "copyBuffer" in k and v > 1e5]
            if vals:
                calib[f"copy_1GiB_{ctr}_KB"] = sum(vals) / len(vals)
    res = {"tag": tag, "probe": probe, "pmc": pmc, "calibration": calib, "passes_left_out": skipped}
    is_pm = "_pm" in tag
    main = "k_posterior_mean" if is_pm else probe.get("step_kernel", "k_step64")
    kk = [k for k in pmc if main in k and "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]]
    if kk:
        e = pmc[kk[0]]
        # MI355X_MICROARCH.md §HBM: counters are in KB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16-B-per-lane
        # reads (confirmed in-process: a 1 GiB copy reads FETCH_SIZE = 524288 KB), WRITE_SIZE is exact
        rd_corr = GIB / (calib["copy_1GiB_FETCH_SIZE_KB"] * 1024) if "copy_1GiB_FETCH_SIZE_KB" in calib else 2.0
        wr_corr = GIB / (calib["copy_1GiB_WRITE_SIZE_KB"] * 1024) if "copy_1GiB_WRITE_SIZE_KB" in calib else 1.0
        rd, wr = e["FETCH_SIZE"] * 1024 * rd_corr, e["WRITE_SIZE"] * 1024 * wr_corr
        res["traffic"] = {"kernel": kk[0], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                          "bytes_per_launch": rd + wr, "read_correction": rd_corr, "write_correction": wr_corr,
                          "kernel_avg_us": e.get("avg_us")}
        if not is_pm and probe.get("src_sha"):
            tl = os.path.join(ROOT, "profiles", "traffic_latest.json")
            cur = json.load(open(tl)) if os.path.exists(tl) else {}
            cur = {k: v for k, v in cur.items() if isinstance(v, dict)}
            commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                                    text=True).stdout.strip()
            cur[tag] = dict(res["traffic"], src_sha=probe["src_sha"], step_kernel=probe.get("step_kernel"),
                            commit=commit, profile=f"profiles/{rnd}/pmc_{tag}.json")
            json.dump(cur, open(tl, "w"), indent=1)
    json.dump(res, open(os.path.join(out, f"pmc_{tag}.json"), "w"), indent=1)
    print(tag, json.dumps(res.get("traffic"), indent=1))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("round")
    p.add_argument("tags", nargs="*", default=["configs2"])
    a = p.parse_args()
    out = os.path.join(ROOT, "profiles", a.round)
    os.makedirs(out, exist_ok=True)
    for tag in a.tags:
        collect(tag, out, a.round)
    logs = ["toolchain.txt", "pm_i8_variants.log", "bench.log", "pytest_gpu.log", "smoke.log", "exp_step_kernels.log", "ab_step64.log", "ab2.log", "nsweep.log",
            "bench_rollout.log", "bench_configs1.log", "bench_configs3.log", "bench_gloo2.log", "pm_trace.log",
            "pm_variants.log", "pm_tests_matrix.log", "pm_rollout.log", "mfma_overlap_probe.log"]
    # gpurun_out/ is scratch that survives rounds: only logs written after the newest file of the other rounds'
    # directories belong to this one
    older = [os.path.getmtime(f) for d in glob.glob(os.path.join(ROOT, "profiles", "r*")) if os.path.abspath(d) != os.path.abspath(out)
             for f in glob.glob(os.path.join(d, "*"))]
    since = max(older) if older else 0.0
    for f in logs:
        src = os.path.join(GO, f)
        if os.path.exists(src) and os.path.getmtime(src) > since:
            shutil.copy(src, os.path.join(out, f))
        elif os.path.exists(src):
            print(f"{f}: older than the previous round's evidence, not copied")


if __name__ == "__main__":
    main()
