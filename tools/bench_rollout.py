#!/usr/bin/env python3
"""Throughput of the on-device policy rollout (one launch = one 153-day episode for every env)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from weather2alert_amd import HeatAlertVecEnv, synth, tables

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
# bench_rollout.py [num_envs [iid|sorted|iid_unordered]]; *_unordered = rollout_order=False (envs visited by index)
# iid = visiting order + matrix-core kernel (k_rollout_mfma); iid_vector = visiting order, k_rollout64 (rollout_mfma=False)
orders = (sys.argv[2],) if len(sys.argv) > 2 else ("iid", "iid_vector", "iid_unordered", "sorted")
for order in orders:
    env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=True, episode_order=order.split("_")[0],
                          rollout_order=not order.endswith("_unordered"), rollout_mfma=not order.endswith("_vector"))
    env.reset(seed=0)
    pol = dict(kind="threshold", feature="heat_qi", threshold=0.9, require_budget=True)
    env.rollout(pol)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        out = env.rollout(pol)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"rollout {order}: {n} envs x {ct.T} days in {dt * 1e3:.2f} ms = {n * ct.T / dt / 1e9:.1f} G env-steps/s "
          f"(mean return {float(out['return'].mean()):.2f}, mean alerts {float(out['alerts'].float().mean()):.2f})")
    env.close()
