#!/usr/bin/env python3
"""A/B of the two step kernels (k_step64 vs the 4-lanes-per-env k_step) at 1 M envs on the row-gather path:
random / sorted / single episode tuples (how much is the coefficient gather?), with and without the
observation write, under a sparse (Bernoulli 0.1) and an always-alert policy with budget 153 (every env
fetches both coefficient rows every day: the policy-pessimistic case).

    python tools/exp_step_kernels.py [--quick] [--kernels auto,classic] [--steps 306]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from weather2alert_amd import HeatAlertVecEnv, synth, tables

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--kernels", default="auto,classic")
ap.add_argument("--steps", type=int, default=306)
ap.add_argument("--num-envs", type=int, default=1 << 20)
ap.add_argument("--tag", default="")
ap.add_argument("--pool", type=int, default=8, help="number of distinct action tensors cycled through")
args = ap.parse_args()

dev = torch.device("cuda:0")
n = args.num_envs
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
dt = tables.DeviceTables(ct, dev)
rng = np.random.default_rng(0)
county = rng.integers(0, ct.S, n)
base = dict(county_w=ct.fips_to_weather[county].astype(np.int64), year_i=rng.integers(0, ct.Y, n), coef_col=county,
            sample=rng.integers(0, ct.n_samples, n), budget=rng.integers(0, 12, n))
g = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(args.pool)] * (8 // args.pool)
ones = [torch.ones(n, dtype=torch.int32, device=dev)] * 8


def run(tag, ep, kernel, obs, acts):
    env = HeatAlertVecEnv(n, tables=dt, device=dev, autoreset="disabled", write_obs=obs, step_kernel=kernel)
    best = 1e9
    for rep in range(2):  # second repetition: clocks and caches warm
        env.reset(options={"episodes": ep})
        for i in range(10):
            env.step(acts[i & 7])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = min(args.steps, 140)  # stay inside one episode (autoreset is disabled)
        e0.record()
        for i in range(K):
            env.step(acts[i & 7])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / K)
    print(f"{args.tag}{tag:30s} kernel={kernel:8s} obs={obs!s:5s}: {best:7.2f} us/step  {n / best * 1e-3:6.2f} G env-steps/s",
          flush=True)
    env.close()


order = np.lexsort((base["sample"], base["coef_col"]))
srt = {k: v[order] for k, v in base.items()}
same = {k: np.full(n, v[0]) for k, v in base.items()}
heavy = dict(base, budget=np.full(n, 153))
for kernel in args.kernels.split(","):
    for obs in (True, False):
        run("random tuples, p=0.1", base, kernel, obs, pool)
        run("sorted tuples, p=0.1", srt, kernel, obs, pool)
        if not args.quick:
            run("one tuple, p=0.1", same, kernel, obs, pool)
            run("random, always alert b=153", heavy, kernel, obs, ones)
