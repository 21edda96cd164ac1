#!/usr/bin/env python3
"""reward_mode="posterior_mean" at 1 M envs (BASELINE configs[2] tables), per reward kernel of the library: step() time
(k_pm_prep + reward kernel + k_step64<given>) and the whole-episode rollout() (one launch where the kernel has a
one-launch form -- k_pm_rollout_i8 / k_pm_rollout -- else policy kernel + pre-pass + reward kernel + step kernel per day).
    python tools/exp_posterior.py [--kernels matrix_i8,vector,matrix]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from weather2alert_amd import HeatAlertVecEnv, synth, tables  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--kernels", default="matrix_i8,vector,matrix")
a = p.parse_args()
dev = torch.device("cuda:0")
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
dt = tables.DeviceTables(tables.compile_from_synth(sd), dev)
n = 1 << 20
g = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(8)]
pol = dict(kind="threshold", feature="heat_qi", threshold=0.9, require_budget=True)
for name in a.kernels.split(","):
    env = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=True, reward_mode="posterior_mean",
                          pm_kernel=name)
    env.reset(seed=0)
    for i in range(5):
        env.step(pool[i & 7])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(40):
        env.step(pool[i & 7])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 40
    env.reset(seed=1)
    env.rollout(pol)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = env.rollout(pol)
    torch.cuda.synchronize()
    dt_r = time.perf_counter() - t0
    T = env.ct.T
    print(f"{name:10s} step {us:7.1f} us = {n / us / 1e3:5.2f} G env-steps/s | rollout: {n} envs x {T} days in "
          f"{dt_r * 1e3:6.2f} ms = {n * T / dt_r / 1e9:5.2f} G env-steps/s ({dt_r * 1e6 / T:5.1f} us per day), "
          f"mean return {float(out['return'].double().mean()):.6f}", flush=True)
    env.close()
