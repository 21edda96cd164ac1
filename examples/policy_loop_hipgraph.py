#!/usr/bin/env python3
"""A policy -> step() loop recorded into a hipGraph (torch.cuda.graph) and replayed: step() neither synchronises nor
allocates and reads the day from device memory, so whole blocks of `policy(obs) -> env.step(actions)` -- episode
boundaries included, with the autoreset inside the step kernel (lockstep=False) -- can be replayed without returning to
Python. On a lock-step batch of >= 131 072 envs the recorded steps keep streaming the 16-B packed state.

    python examples/policy_loop_hipgraph.py            # needs one ROCm GPU
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from weather2alert_amd import HeatAlertVecEnv, compile_from_synth, synth

data = synth.make_synth("linear", n_fips=64, years=[2006, 2007, 2008], n_samples=20, seed=0, extra_confounder_fips=6)
tables = compile_from_synth(data)
n, G = 262144, 51                      # 51 recorded days: three replays = one 153-day episode
envs = HeatAlertVecEnv(n, tables=tables, similar_climate_counties=True, lockstep=False)  # restarts inside the step kernel
obs, _ = envs.reset(seed=0)
dev = obs.device
heat = obs[:, envs.feature_names.index("heat_qi")]     # a view of the env's own observation buffer (rewritten by step())
left = obs[:, envs.feature_names.index("remaining_budget")]
actions = torch.zeros(n, dtype=torch.bool, device=dev)
total = torch.zeros(n, device=dev)


def one_day():
    torch.logical_and(heat > 0.92, left > 0, out=actions)  # the "policy": alert on hot days while budget is left
    _, reward, _, _, _ = envs.step(actions)
    total.add_(reward)


# one eager warm-up day (it also puts the batch on its packed form), then the capture; .replay() reads the env's status
# word behind every replay, so a block that could not run (W2A_ST_STALE_GRAPH) raises instead of going unnoticed
block = envs.record_steps(one_day, G)
print("recorded", G, "days; step kernel:", envs.last_step_kernel, "| packed state:", envs.packed_state)
torch.cuda.synchronize()
t0 = time.perf_counter()
replays = 30
for _ in range(replays):
    block.replay()
block.finish()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = envs.state()
print(f"{replays * G} days of {n} envs in {dt * 1e3:.1f} ms = {n * replays * G / dt / 1e9:.1f} G env-steps/s; episodes finished per env: "
      f"{int(st['episode_no'].min())}; mean reward per day {float(total.mean()) / (replays * G + 1):.3f}; status {envs.check_status()}")
envs.close()
