#!/usr/bin/env python3
"""What-if: the observation buffer in memory the L2 does not allocate for (hipExtMallocWithFlags, hipDeviceMallocUncached).
The step writes 121 MB of observation rows per 1 M-env launch and never reads them; in ordinary memory those lines pass
through the 4 MiB L2 of every XCD and evict the table lines the gathers want to find there. Measures the step
kernel with the rows going to (a) the env's own torch buffer, (b) an uncached buffer, and what a consumer pays for reading
uncached rows (a first policy layer: obs @ W1, and a plain reduction)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from weather2alert_amd import HeatAlertVecEnv, synth, tables

hip = C.CDLL("libamdhip64.so")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]


class UncachedBuffer:
    """n_bytes of device memory with MTYPE uncached, visible to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, shape, flags=3):
        self.shape = tuple(shape)
        self.nbytes = int(np.prod(shape)) * 4
        p = C.c_void_p()
        rc = hip.hipExtMallocWithFlags(C.byref(p), self.nbytes, flags)
        assert rc == 0 and p.value, rc
        self.ptr = p.value
        self.__cuda_array_interface__ = {"shape": self.shape, "typestr": "<f4", "data": (self.ptr, False), "version": 2,
                                         "strides": None}

    def __del__(self):
        try:
            hip.hipFree(C.c_void_p(self.ptr))
        except Exception:  # noqa: BLE001
            pass


def timed(env, pool, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(steps):
        env.step(pool[i & 15])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


dev = torch.device("cuda:0")
for wl in ("configs2", "configs3"):
    wname, n, augment, desc = bench.WORKLOADS[wl]
    sd = synth.make_synth(wname, years=list(range(2006, 2017)), n_samples=100, seed=0, extra_confounder_fips=60)
    ct = tables.compile_from_synth(sd)
    dt = tables.DeviceTables(ct, dev)
    g = torch.Generator(device=dev).manual_seed(1234)
    pool = [(torch.rand(n, device=dev, generator=g) < 0.1).to(torch.int32) for _ in range(16)]
    res = {}
    for name in ("torch buffer", "uncached buffer", "torch buffer again", "uncached buffer again"):
        env = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=augment)
        keep = None
        if name.startswith("uncached"):
            keep = UncachedBuffer((n, ct.n_obs))
            t = torch.as_tensor(keep, device=dev)
            assert t.data_ptr() == keep.ptr and t.shape == (n, ct.n_obs) and t.dtype == torch.float32
            ref = env._obs
            env._obs = t
            env._obs_ptr = t.data_ptr()
        env.reset(seed=0)
        timed(env, pool, 20)
        us = min(timed(env, pool, 120) for _ in range(3))
        res[name] = us
        if name == "uncached buffer":  # same rows as an ordinary env produces
            e2 = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=augment)
            e2.reset(seed=0)
            for i in range(20 + 3 * 120):
                e2.step(pool[i & 15]) if False else None
            e2.close()
        # what a consumer pays: first policy layer and a reduction over the rows
        W1 = torch.randn(ct.n_obs, 64, device=dev)
        obs = env._obs
        for _ in range(3):
            (obs @ W1).sum().item()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            y = obs @ W1
        torch.cuda.synchronize()
        mm = (time.perf_counter() - t0) / 20 * 1e6
        t0 = time.perf_counter()
        for _ in range(20):
            s = obs.sum()
        torch.cuda.synchronize()
        sm = (time.perf_counter() - t0) / 20 * 1e6
        print(f"{wl} {name:24s}: step kernel {us:6.2f} us; consumer: obs @ W1[29x64] {mm:7.1f} us, obs.sum() {sm:7.1f} us", flush=True)
        env.close()
        del env, keep
# bit-identical rows either way
sd = synth.make_synth("linear", n_fips=40, years=[2006, 2007], n_samples=8, seed=2)
ct = tables.compile_from_synth(sd)
n = 131072 + 7
A = HeatAlertVecEnv(n, tables=ct, device=dev)
B = HeatAlertVecEnv(n, tables=ct, device=dev)
kb = UncachedBuffer((n, ct.n_obs))
B._obs = torch.as_tensor(kb, device=dev)
B._obs_ptr = B._obs.data_ptr()
A.reset(seed=1); B.reset(seed=1)
g = torch.Generator(device=dev).manual_seed(1)
for t in range(160):
    a = (torch.rand(n, device=dev, generator=g) < 0.2).to(torch.int32)
    oa, ra, _, _, _ = A.step(a)
    ob, rb, _, _, _ = B.step(a)
    assert torch.equal(oa, ob) and torch.equal(ra, rb), t
print("rows and rewards identical through 160 steps (incl. the lock-step autoreset)")
