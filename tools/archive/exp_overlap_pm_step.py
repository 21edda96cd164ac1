#!/usr/bin/env python3
"""What-if, priced before building anything: could the memory-bound k_step64<REWARD_GIVEN> of a posterior-mean step run
UNDER the vector-issue-bound k_posterior_mean_i8 (two streams)? Two independent 1 M-env batches stand in for the two
halves: batch A runs only its reward kernels (w2a_posterior_mean_reward = k_pm_prep + k_posterior_mean_i8), batch B
only k_step64<given>; timed back to back on one stream and concurrently on two.
    python tools/exp_overlap_pm_step.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from weather2alert_amd import HeatAlertVecEnv, _ffi, synth, tables  # noqa: E402

dev = torch.device("cuda:0")
wname, n, augment, _ = bench.WORKLOADS["configs2"]
sd = synth.make_synth(wname, years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
dt = tables.DeviceTables(tables.compile_from_synth(sd), dev)
g = torch.Generator(device=dev).manual_seed(1)
act = (torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32)
A = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=augment, reward_mode="posterior_mean", autoreset="disabled")
B = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=augment, reward_mode="posterior_mean", autoreset="disabled")
A.reset(seed=0)
B.reset(seed=1)
for _ in range(20):
    A.step(act)
    B.step(act)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
flags = B._step_flags


def reward(stream):
    _ffi.check(A._lib.w2a_posterior_mean_reward(A._h, act.data_ptr(), _ffi.ACT_I32, A._rew_ptr, stream.cuda_stream), "pm")


def step(stream):  # B's reward buffer holds a valid reward from its last full step: k_step64<given> just consumes it
    _ffi.check(B._lib.w2a_step(B._h, act.data_ptr(), _ffi.ACT_I32, B._obs_ptr, B._rew_ptr, B._done_ptr, B._fr_ptr, flags,
                               stream.cuda_stream), "step")


def timed(fn, reps=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


B.state()  # canonical form current
r_only = timed(lambda: reward(sa))
B.reset(seed=1)
s_only = timed(lambda: step(sb), reps=30)
B.reset(seed=1)
seq = timed(lambda: (reward(sa), step(sa)))
B.reset(seed=1)


def both():
    reward(sa)
    step(sb)


conc = timed(both)
print(f"reward kernels alone {r_only:.1f} us, k_step64<given> alone {s_only:.1f} us, one stream {seq:.1f} us, two streams {conc:.1f} us "
      f"(saves {seq - conc:.1f} us of a ~124 us posterior-mean step)")
