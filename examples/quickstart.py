#!/usr/bin/env python3
"""Quick start on synthetic data (the real HeatAlertsRL files are on the HF hub): a vector env stepped by a
random policy, an on-device threshold-policy evaluation, and the single-env drop-in.

    python examples/quickstart.py            # needs one ROCm GPU
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from weather2alert_amd import HeatAlertEnv, HeatAlertVecEnv, compile_from_synth
from weather2alert_amd import synth

data = synth.make_synth("linear", n_fips=64, years=[2006, 2007, 2008], n_samples=20, seed=0, extra_confounder_fips=6)
tables = compile_from_synth(data)

# 1. 65 536 envs, random policy, same-step autoreset (device tensors throughout)
envs = HeatAlertVecEnv(65536, tables=tables, similar_climate_counties=True)
obs, info = envs.reset(seed=0)
total = torch.zeros(envs.num_envs, device=obs.device)
for _ in range(2 * 153):
    actions = (torch.rand(envs.num_envs, device=obs.device) < 0.05).to(torch.uint8)
    obs, reward, terminated, truncated, info = envs.step(actions)
    total += reward
print("random policy: mean reward per day", float(total.mean()) / 306, "| episodes finished:",
      int(envs.state()["episode_no"].min()))

# 2. evaluate "alert when yesterday's heat index quantile > q" for a few q, one launch per episode
for q in (0.7, 0.8, 0.9, 0.95):
    out = envs.rollout({"kind": "threshold", "feature": "heat_qi", "threshold": q, "require_budget": True},
                       alert_mask=True)
    s = HeatAlertVecEnv.episode_stats(out)
    print(f"threshold {q}: mean return {s['mean_return']:.2f}, alerts/episode {s['mean_alerts']:.2f}, "
          f"80% of alerts issued by day {s['alert_t_80%']:.0f}, over-budget frequency {s['over_budget_freq']:.4f}")
envs.close()

# 3. the drop-in for weather2alert.env.HeatAlertEnv (same reset/step signatures, NumPy-seed parity)
env = HeatAlertEnv(weights="linear", tables=tables)
obs, info = env.reset(location=data.fips_list[0], seed=0)
done, ret = False, 0.0
while not done:
    obs, r, done, _, info = env.step(env.action_space.sample())
    ret += r
print("single env:", info["episode_index"], "return", round(ret, 3), "remaining budget", info["remaining_budget"])
env.close()
