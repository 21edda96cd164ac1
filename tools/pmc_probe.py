#!/usr/bin/env python3
"""Workload for rocprofv3 counter passes: N env steps of a bench workload plus calibration
kernels of KNOWN HBM byte counts (a 1 GiB device copy and a 1 GiB fill), so FETCH_SIZE /
WRITE_SIZE can be calibrated in this process (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads 1/2
on wide coalesced reads on gfx950; other shapes must be calibrated).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_probe.py
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
from weather2alert_amd import HeatAlertVecEnv, synth, tables  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--workload", default="configs2")
p.add_argument("--steps", type=int, default=24)
p.add_argument("--num-envs", type=int, default=None)
p.add_argument("--no-obs", action="store_true")
p.add_argument("--step-kernel", default="auto", choices=["auto", "classic", "wide", "unpacked"])
p.add_argument("--episode-order", default="iid", choices=["iid", "sorted"])
p.add_argument("--reward-mode", default="sampled", choices=["sampled", "posterior_mean"])
p.add_argument("--pm-kernel", default="vector", choices=["vector", "matrix", "matrix_i8"],
               help="posterior-mean reward kernel (all are in the one library, selected at run time)")
a = p.parse_args()
wname, n_default, augment, desc = bench.WORKLOADS[a.workload]
n = a.num_envs or n_default
dev = torch.device("cuda:0")
sd = synth.make_synth(wname, years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=augment, write_obs=not a.no_obs,
                      step_kernel=a.step_kernel, episode_order=a.episode_order, reward_mode=a.reward_mode,
                      pm_kernel=a.pm_kernel)
g = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(8)]
env.reset(seed=0)
env_kernel = env.step_kernel_name
# calibration: 1 GiB copy (reads 1 GiB, writes 1 GiB) and 1 GiB fill, float32
src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
dst = torch.empty_like(src)
torch.cuda.synchronize()
for _ in range(3):
    dst.copy_(src)
    dst.fill_(1.0)
torch.cuda.synchronize()
for i in range(a.steps):
    env.step(pool[i & 7])
torch.cuda.synchronize()
import json  # noqa: E402

from weather2alert_amd import build as wbuild  # noqa: E402

os.makedirs("gpurun_out", exist_ok=True)
tag = (a.workload + ("_noobs" if a.no_obs else "") + ("_sorted" if a.episode_order == "sorted" else "")
       + (("_pm_" + a.pm_kernel) if a.reward_mode == "posterior_mean" else ""))
json.dump({"workload": tag, "src_sha": wbuild.source_sha(), "num_envs": n, "steps": a.steps,
           "step_kernel": env_kernel},
          open(f"gpurun_out/pmc_probe_{tag}.json", "w"))
print("probe done", desc, "steps", a.steps)
env.close()
