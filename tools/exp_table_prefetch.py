import os, sys, time
sys.path.insert(0, '.')
import torch
from weather2alert_amd import HeatAlertVecEnv, synth, tables
dev = torch.device("cuda:0")
n = 1 << 20
sd = synth.make_synth("nn_full_medicare_all", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
dt = tables.DeviceTables(ct, dev).build_logit_table()
g = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(8)]
for pre in (False, True, False, True):
    env = HeatAlertVecEnv(n, tables=dt, device=dev, reward_path="table", autoreset="disabled")
    env.reset(seed=0)
    evs = []
    for t in range(150):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.step(pool[t & 7]); e1.record(); evs.append((e0, e1))
        if pre:
            dt.L[t + 1].view(-1)[::8].sum()      # touches every 64 B of tomorrow's slice
            dt.X[t + 1].view(-1)[::16].sum()
    torch.cuda.synchronize()
    us = sum(a.elapsed_time(b) for a, b in evs[10:]) / len(evs[10:]) * 1e3
    print(f"prefetch={pre}: step kernel {us:.2f} us")
    env.close()
