#!/usr/bin/env python3
"""What a short timed window pays at its end: torch.cuda.synchronize() alone against an event-query spin in front of it.
20 steps of 1 048 576 envs between two device synchronisations, wall clock, 40 windows each way (profiles/r06/exp_sync_latency.log)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from weather2alert_amd import HeatAlertVecEnv, synth, tables

dev = torch.device("cuda:0")
n = 1 << 20
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=True)
env.reset(seed=0)
acts = [(torch.rand(n, device=dev) < 0.1).to(torch.int32) for _ in range(8)]
for i in range(40):
    env.step(acts[i & 7])
torch.cuda.synchronize()
K = 20


def window(spin: bool, timing_events: int = 0, collect: bool = False) -> float:
    if collect:
        import gc
        gc.collect()
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    t0 = time.perf_counter()
    if timing_events:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    for i in range(K):
        env.step(acts[i & 7])
        if i == 0 and timing_events >= 3:
            e_mid = torch.cuda.Event(enable_timing=True)
            e_mid.record()
    if timing_events:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
    ev.record()
    if spin:
        while not ev.query():
            pass
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def report(tag, **kw):
    ts = sorted(window(**kw) for _ in range(40))
    print(f"{tag:58s}: median {ts[20] * 1e6 / K:6.2f} us per step, best {ts[0] * 1e6 / K:6.2f}, worst {ts[-1] * 1e6 / K:6.2f}  "
          f"({K}-step windows, wall)", flush=True)


for rep in range(2):
    report("synchronize alone", spin=False)
    report("event-query spin + synchronize", spin=True)
    report("timing events at both ends (e0, e1)", spin=False, timing_events=2)
    report("e0, e1 and one behind the first step", spin=False, timing_events=3)
    report("e0, e1, mid + gc.collect() before the window", spin=False, timing_events=3, collect=True)
    report("e0, e1, mid + gc.collect() + spin", spin=True, timing_events=3, collect=True)
env.close()
