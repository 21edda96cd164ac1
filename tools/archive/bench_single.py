#!/usr/bin/env python3
"""Latency of the num_envs=1 drop-in HeatAlertEnv (every call synchronises, like the reference's API)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from weather2alert_amd import HeatAlertEnv, tables
ct = tables.CompiledTables.load_npz(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "mini_compiled.npz"))
env = HeatAlertEnv(weights="linear", tables=ct)
env.reset(location="06037", seed=0)
rng = np.random.default_rng(0)
t0 = time.perf_counter(); k = 0
for ep in range(5):
    t1 = time.perf_counter(); env.reset(location="06037", seed=ep); tr = time.perf_counter() - t1
    done = False
    while not done:
        _, _, done, _, _ = env.step(int(rng.random() < 0.1)); k += 1
dt = time.perf_counter() - t0
print(f"HeatAlertEnv drop-in: {dt / k * 1e3:.3f} ms/step ({k / dt:.0f} env-steps/s), reset {tr * 1e3:.2f} ms  [reference: 1.55 ms/step, 3.0 ms/reset]")
