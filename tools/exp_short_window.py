#!/usr/bin/env python3
"""Why a 20-step window (the driver's `bench.py --steps 20 --warmup 5`) reads ~41 us/step when the steady-state
kernel takes ~37.5 us: per-step device times inside consecutive 20-step windows of the first episode. Early in an
episode budgets are not yet exhausted, ~10 % of env-days issue an alert and fetch the second coefficient row;
over a whole episode it is ~6 % (policy dependence)."""
import sys, time, torch
sys.path.insert(0,'.')
from weather2alert_amd import HeatAlertVecEnv, synth, tables
dev=torch.device("cuda:0")
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd); n=1<<20
env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=True)
g = torch.Generator(device=dev).manual_seed(1234)
pool=[(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(16)]
env.reset(seed=0)
for i in range(5): env.step(pool[i&15])
for rep in range(4):
    torch.cuda.synchronize()
    evs=[torch.cuda.Event(enable_timing=True) for _ in range(21)]
    t0=time.perf_counter()
    evs[0].record()
    for i in range(20):
        env.step(pool[i&15]); evs[i+1].record()
    th=time.perf_counter()
    torch.cuda.synchronize()
    t1=time.perf_counter()
    per=[evs[i].elapsed_time(evs[i+1])*1e3 for i in range(20)]
    print(f"rep{rep}: wall {1e6*(t1-t0):.0f} us, host enqueue {1e6*(th-t0):.0f} us, device {evs[0].elapsed_time(evs[20])*1e3:.0f} us; per-step us:", " ".join(f"{x:.0f}" for x in per))
