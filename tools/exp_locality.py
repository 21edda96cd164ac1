#!/usr/bin/env python3
"""Experiment: how much of the step time is the coefficient/table gather? Times k_step at 1M envs
with (a) random episode tuples, (b) the same tuples sorted by (coef_col, sample), (c) one tuple
for every env (best-case locality), with and without the observation write, gather vs table."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from weather2alert_amd import HeatAlertVecEnv, synth, tables

dev = torch.device("cuda:0")
n = 1 << 20
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
dt = tables.DeviceTables(ct, dev)
rng = np.random.default_rng(0)
county = rng.integers(0, ct.S, n)
base = dict(county_w=ct.fips_to_weather[county].astype(np.int64), year_i=rng.integers(0, ct.Y, n), coef_col=county,
            sample=rng.integers(0, ct.n_samples, n), budget=rng.integers(0, 12, n))
g = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(8)]


def run(tag, ep, path, obs):
    env = HeatAlertVecEnv(n, tables=dt, device=dev, autoreset="disabled", reward_path=path, write_obs=obs)
    env.reset(options={"episodes": ep})
    for i in range(10):
        env.step(pool[i & 7])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K = 100
    for i in range(K):
        env.step(pool[i & 7])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / K
    print(f"{tag:34s} path={path:6s} obs={obs!s:5s}: {us:7.2f} us/step  {n / us * 1e-3:6.2f} G env-steps/s", flush=True)
    env.close()


order = np.lexsort((base["sample"], base["coef_col"]))
srt = {k: v[order] for k, v in base.items()}
same = {k: np.full(n, v[0]) for k, v in base.items()}
quick = "--quick" in sys.argv
for path in ("gather", "table"):
    for obs in (True, False):
        run("random tuples", base, path, obs)
        if not quick:
            run("sorted by (coef_col, sample)", srt, path, obs)
        run("one tuple for all envs", same, path, obs)
