#!/usr/bin/env python3
"""Step time of the three ways a same-step-autoreset batch can run (configs[2] shape, 1 048 576 envs, 2 episodes):
lock step with the host-launched reset (k_step64 on the packed state), lockstep=False (k_step with the in-kernel
autoreset: what a batch falls back to after a masked reset or with ragged episode lengths), and the latter forced onto
the classic kernel in lock step for comparison.   python tools/exp_autoreset_modes.py [num_envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from weather2alert_amd import HeatAlertVecEnv, synth, tables  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda:0")
sd = synth.make_synth("linear", years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
ct = tables.compile_from_synth(sd)
g = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(16)]
for name, kw in (("lock step, host-launched reset (k_step64, packed state)", dict()),
                 ("lock step, classic kernel", dict(step_kernel="classic")),
                 ("lockstep=False: in-kernel autoreset", dict(lockstep=False)),
                 ("corrected semantics (faithful=False), lock step", dict(faithful=False)),
                 ("corrected semantics except the observation fix", dict(fixes={"alert_2wks", "lag", "penalty", "augment", "budget"}))):
    env = HeatAlertVecEnv(n, tables=ct, device=dev, similar_climate_counties=True, **kw)
    env.reset(seed=0)
    for i in range(160):
        env.step(pool[i % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(306):
        env.step(pool[i % 16])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 306
    print(f"{name}: {dt * 1e6:.1f} us/step = {n / dt / 1e9:.2f} G env-steps/s ({env.step_kernel_name})", flush=True)
    env.close()
