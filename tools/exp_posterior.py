#!/usr/bin/env python3
"""Step time of reward_mode="posterior_mean" at 1 M envs (k_pm_prep + k_posterior_mean + k_step64<given>).
With a library built with -DW2A_PM_EXPERIMENT=1 (no sigmoid epilogue) or =2 (VALU FMAs instead of the MFMAs) it
separates the MFMA time from everything else (results are wrong in those builds)."""
import sys, torch, json
sys.path.insert(0,'.')
from weather2alert_amd import HeatAlertVecEnv, synth, tables
dev=torch.device("cuda:0")
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd); n=1<<20
env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=True, reward_mode="posterior_mean")
env.reset(seed=0)
g = torch.Generator(device=dev).manual_seed(1)
pool=[(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(8)]
for i in range(5): env.step(pool[i&7])
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40): env.step(pool[i&7])
e1.record(); torch.cuda.synchronize()
print("pm us/step", e0.elapsed_time(e1)*1e3/40)
