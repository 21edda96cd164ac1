import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from weather2alert_amd import HeatAlertVecEnv, synth, tables
dev = torch.device("cuda:0")
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
n = 1 << 20
env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=True)
g = torch.Generator(device=dev).manual_seed(1234)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(16)]
for rep in range(4):
    env.reset(seed=rep)
    for i in range(5):
        env.step(pool[i])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(22)]
    ts = [time.perf_counter()]
    ev[0].record(); ts.append(time.perf_counter())
    for i in range(20):
        env.step(pool[i & 15]); ev[i + 1].record(); ts.append(time.perf_counter())
    ts.append(time.perf_counter())
    torch.cuda.synchronize(); ts.append(time.perf_counter())
    us = lambda a, b: (b - a) * 1e6
    dev_total = ev[0].elapsed_time(ev[20]) * 1e3
    per = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(20)]
    print(f"rep {rep}: wall {us(ts[0], ts[-1]):.0f} us, device ev0->ev20 {dev_total:.0f} us; host: ev0.record {us(ts[0], ts[1]):.1f}, "
          f"first step+record {us(ts[1], ts[2]):.1f}, steps 2..20 {us(ts[2], ts[21]):.0f} (all launches queued at +{us(ts[0], ts[21]):.0f}), "
          f"sync wait {us(ts[22], ts[23]):.0f}; device per launch: first {per[0]:.1f}, rest mean {sum(per[1:]) / 19:.1f}")
