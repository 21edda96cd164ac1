#!/usr/bin/env python3
"""Why a 20-step window (the driver's `bench.py --steps 20 --warmup 5`) reads ~41 us/step when the default
10-episode run reads 37.5: device time of consecutive 20-step windows (HIP events around each window, nothing
between the steps) through two episodes, with the share of env-days that issued an alert in each window."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from weather2alert_amd import HeatAlertVecEnv, synth, tables

dev = torch.device("cuda:0")
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
n = 1 << 20
env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=True)
g = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(16)]
env.reset(seed=0)
step = 0
for _ in range(5):
    env.step(pool[step & 15]); step += 1
for w in range(15):
    used0 = env.state()["used"].sum().item()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(20):
        env.step(pool[step & 15]); step += 1
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    used1 = env.state()["used"].sum().item()
    d0 = (step - 20) % ct.T
    print(f"days {d0:3d}..{d0 + 19:3d}: device {e0.elapsed_time(e1) * 50:.2f} us/step, wall {wall / 20:.2f} us/step, "
          f"alerts issued on {max(used1 - used0, 0) / (20 * n) * 100:.1f} % of env-days", flush=True)
