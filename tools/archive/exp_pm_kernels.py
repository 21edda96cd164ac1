#!/usr/bin/env python3
"""Reward-kernel time of reward_mode="posterior_mean" per kernel of the library (HIP events around the C-ABI call
w2a_posterior_mean_reward = pre-pass + reward kernel), 1 048 576 envs, BASELINE configs[2] / configs[3] tables.
    python tools/exp_pm_kernels.py [--workload configs2] [--kernels vector,matrix,matrix_i8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from weather2alert_amd import HeatAlertVecEnv, _ffi, synth, tables  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--workload", default="configs2")
p.add_argument("--kernels", default="vector,matrix,matrix_i8")
p.add_argument("--tag", default="")
a = p.parse_args()
wname, n, augment, desc = bench.WORKLOADS[a.workload]
dev = torch.device("cuda:0")
sd = synth.make_synth(wname, years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
dt = tables.DeviceTables(tables.compile_from_synth(sd), dev)
g = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(8)]
for name in a.kernels.split(","):
    env = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=augment, reward_mode="posterior_mean",
                          pm_kernel=name)
    env.reset(seed=0)
    for i in range(30):  # a month into the episode: alerts and budgets in a typical state
        env.step(pool[i & 7])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(20):
        _ffi.check(env._lib.w2a_posterior_mean_reward(env._h, pool[i & 7].data_ptr(), _ffi.ACT_I32, env._rew_ptr,
                                                      env._stream()), "w2a_posterior_mean_reward")
    e1.record()
    torch.cuda.synchronize()
    r = env._reward.clone()
    print(f"{a.tag}{a.workload} {name:10s} pre-pass + reward kernel {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us   "
          f"mean reward {float(r.double().mean()):.9f}", flush=True)
    env.close()
