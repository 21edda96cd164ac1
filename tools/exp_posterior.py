#!/usr/bin/env python3
"""Step time of reward_mode="posterior_mean" at 1 M envs (k_pm_prep + k_posterior_mean_v + k_step64<given>).
Build variants (W2A_CXXFLAGS, see tools/exp_pm_variants.sh / exp_pm_trace.sh): -DW2A_PM_MATRIX=1 the fp64-MFMA form,
-DW2A_PMV_NPAD=<draws staged per pass>, -DPMV_THREADS=<256|512|1024>,
-DW2A_PMV_DEBUG_DRAWS=<n> (caps the baseline draw loop: timing only, results wrong). DESIGN.md section 4."""
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
# whole-episode policy rollout in this mode (policy kernel + reward kernels + step kernel per day, no observations)
import time
env.reset(seed=1)
pol = dict(kind="threshold", feature="heat_qi", threshold=0.9, require_budget=True)
env.rollout(pol)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = env.rollout(pol)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("pm rollout: %d envs x %d days in %.2f ms = %.2f G env-steps/s (%.1f us per day)" % (n, ct.T, dt * 1e3, n * ct.T / dt / 1e9, dt * 1e6 / ct.T))
