#!/usr/bin/env python3
"""k_rollout_mfma's time against the number of days one launch runs (1 048 576 envs, threshold policy): what of a launch is
per wave (state / coefficient-digit gathers by env id, the final scattered stores) and what is per day.
Run under `rocprofv3 --kernel-trace` and read the per-launch durations (tools/rocprof_summary.py), or take the HIP-event
figures it prints.   python tools/exp_rollout_nsteps.py [num_envs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from weather2alert_amd import HeatAlertVecEnv, synth, tables  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0")
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=True, autoreset="disabled")
pol = dict(kind="threshold", feature="heat_qi", threshold=0.9, require_budget=True)
for days in (1, 16, 32, 64, 153):
    best = 1e9
    for rep in range(4):
        env.reset(seed=rep)
        env.rollout(pol, n_steps=1)  # order + tile list of this episode are built here; the batch is on day 1
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(pol, n_steps=days)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    print(f"rollout(n_steps={days:3d}) after the order exists: {best:7.1f} us ({env.last_rollout_kernel})", flush=True)
env.close()
