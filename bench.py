#!/usr/bin/env python3
"""Benchmark of the HeatAlertEnv hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload configs2] [--num-envs M]

A "step" is one vector-env step(): every env of the batch advances one day (budget gate,
feature-row gather, two 28-term logits against its posterior draw, reward, next observation),
including the same-step autoreset when an episode ends (every 153 steps, lock step) and, on
N>1 GPUs, the RCCL all-gather of the finished episodes' returns. Inputs (tables, state, a
pool of action tensors) are resident in HBM before the timed region.

Prints ONE JSON line (rank 0). `roofline` prices the step kernel against HBM with SURVEY §8d's
algorithmic bytes (489 B per env-step); `cpu_baseline` times the NumPy oracle on the host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_ENV_STEP = 489  # action 4 + state 20r + row 100 + weights 224 + reward 4 + done 1 + obs 116 + state 20w
ALGO_BYTES_NO_OBS = 373
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)

WORKLOADS = {
    # name: (weights list, num_envs per GPU, similar_climate_counties, reward_path, description)
    "configs1": ("linear", 65536, False, "gather",
                 "configs[1]: 65,536 envs, weights/linear (S=746), random county per env"),
    "configs2": ("linear", 1048576, True, "gather",
                 "configs[2]: 1,048,576 envs, weights/linear (S=746), similar_climate_counties=True"),
    "configs3": ("nn_full_medicare_all", 1048576, False, "table",
                 "configs[3]: 1,048,576 envs, nn_full_medicare_all shape (S=720), logit table from the grouped "
                 "fp64-MFMA GEMM"),
    "configs3_gather": ("nn_full_medicare_all", 1048576, False, "gather",
                        "configs[3] shape on the row-gather kernel (A/B for the table path)"),
    "configs1_table": ("linear", 65536, False, "table", "configs[1] shape on the logit-table path"),
}
ALGO_BYTES_TABLE = 305  # SURVEY §8d logit-table path


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1530, help="timed steps (default: 10 episodes, SURVEY §8d)")
    p.add_argument("--warmup", type=int, default=153, help="untimed steps (default: 1 episode)")
    p.add_argument("--workload", default="configs2", choices=sorted(WORKLOADS))
    p.add_argument("--num-envs", type=int, default=None, help="envs per GPU (overrides the workload's)")
    p.add_argument("--no-obs", action="store_true", help="reward-only step variant")
    p.add_argument("--obs-f16", action="store_true", help="opt-in half-precision observations (58 instead of 116 B/env)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the reported extras (sorted episode order)")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="weak: the workload's env count per GPU (default); strong: that count split over the GPUs")
    p.add_argument("--graph", type=int, default=0,
                   help="capture this many consecutive step() calls into one hipGraph and replay it (launch-bound "
                        "small batches); the timed region still runs exactly --steps steps")
    p.add_argument("--episode-order", default="iid", choices=["iid", "sorted"],
                   help="sorted = opt-in relabelling of envs by table row after each reset (same episode multiset)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo = rehearsal of the multi-rank path with several ranks sharing one GPU")
    return p.parse_args()


def _cpu_worker(args):
    """One host process of the multi-core CPU baseline: builds its own tables (spawned, no GPU), steps the
    NumPy vector oracle on its share of the envs and returns (env_steps, seconds spent stepping)."""
    wname, seed, n, steps, rank = args
    import numpy as np

    from oracle import heatalert_oracle as O
    from weather2alert_amd import synth, tables

    sd = synth.make_synth(wname, years=list(range(2006, 2017)), n_samples=100, seed=seed, extra_confounder_fips=60)
    ct = tables.compile_from_synth(sd)
    V = O.VectorOracle(O.RefData.from_synth(sd), sd.fips_weather, sd.years)
    rng = np.random.default_rng(seed + 1 + rank)
    county = rng.integers(0, ct.S, n)
    cc = rng.integers(0, np.maximum(ct.sim_cnt[county], 1))
    V.reset(ct.fips_to_weather[county].astype(np.int64), rng.integers(0, ct.Y, n), cc,
            rng.integers(0, ct.n_samples, n), rng.integers(0, 12, n))
    acts = (rng.random((steps, n)) < 0.1).astype(np.int64)
    V.step(acts[0])
    t0 = time.perf_counter()
    for t in range(1, steps):
        V.step(acts[t])
    return n * (steps - 1), time.perf_counter() - t0


def cpu_baseline_multicore(wname, seed, procs, timeout=180):
    """`procs` independent host processes (plain subprocesses of this script: no fork of a GPU process, no
    multiprocessing start-method pitfalls), each stepping its own share; aggregate = total / slowest."""
    import subprocess

    n, steps = 16384, 154
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                            json.dumps([wname, seed, n, steps, r])], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           env=env, text=True) for r in range(procs)]
    res = []
    deadline = time.time() + timeout
    for p in ps:
        try:
            out, _ = p.communicate(timeout=max(1.0, deadline - time.time()))
            res.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:  # noqa: BLE001
            p.kill()
    if len(res) != procs:
        raise RuntimeError(f"{procs - len(res)} of {procs} CPU workers failed or timed out")
    total = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)
    return {"value": total / slowest, "cores": procs,
            "sample": f"{procs} processes x {n} envs x {steps - 1} steps, slowest worker {slowest:.1f} s"}


def cpu_baseline(sd, ct, seed=0):
    """NumPy oracle (float64, vectorised over envs) timed on the host: bounded sample of the
    same workload. Also times the scalar per-env restatement of the reference's step()."""
    import numpy as np

    from oracle import heatalert_oracle as O

    rd = O.RefData.from_synth(sd)
    V = O.VectorOracle(rd, sd.fips_weather, sd.years)
    rng = np.random.default_rng(seed)
    n, steps = 65536, 154
    county = rng.integers(0, ct.S, n)
    cc = rng.integers(0, np.maximum(ct.sim_cnt[county], 1))
    V.reset(ct.fips_to_weather[county].astype(np.int64), rng.integers(0, ct.Y, n), cc,
            rng.integers(0, ct.n_samples, n), rng.integers(0, 12, n))
    acts = (rng.random((steps, n)) < 0.1).astype(np.int64)
    V.step(acts[0])
    t0 = time.perf_counter()
    for t in range(1, steps):
        V.step(acts[t])
    dt = time.perf_counter() - t0
    vec = n * (steps - 1) / dt
    env = O.OracleEnv(rd)
    env.reset(location=sd.fips_list[0], seed=0)
    t0 = time.perf_counter()
    k = 0
    for ep in range(3):
        env.reset(location=sd.fips_list[ep], seed=ep)
        done = False
        while not done:
            _, _, done, _, _ = env.step(int(acts[k % steps, k % n]))
            k += 1
    scalar = k / (time.perf_counter() - t0)
    return {
        "value": vec, "unit": "env-steps/s", "cores": 1, "kind": "port",
        "sample": f"NumPy float64 vector oracle, {n} envs x {steps - 1} steps of the same tables ({dt:.1f} s)",
        "scalar_port_env_steps_per_s": scalar,
        "host_cpus": os.cpu_count(),
        "reference_env_steps_per_s_build_container": 646.0,  # BASELINE.md §2 (pandas env, 1 core; it cannot travel)
    }


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        print(json.dumps(_cpu_worker(tuple(json.loads(sys.argv[2])))))
        return
    args = parse()
    import numpy as np
    import torch

    from weather2alert_amd import HeatAlertVecEnv, dist as wdist, synth, tables

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        print(f"bench.py: --gpus {args.gpus} needs one process per GPU (python -m torch.distributed.run --nnodes=1 "
              f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...); running on 1 GPU",
              file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU"
    device = torch.device(f"cuda:{local % torch.cuda.device_count() if args.backend == 'gloo' else local}")
    torch.cuda.set_device(device)
    wdist.init_from_env(args.backend, device)

    wname, n_default, augment, rpath, desc = WORKLOADS[args.workload]
    n = args.num_envs or n_default
    if args.scaling == "strong":
        start, stop = wdist.shard_range(n, rank, world)
        if (stop - start) * world != n:
            raise SystemExit("--scaling strong needs an env count divisible by the number of GPUs")
        n = stop - start
    t_setup = time.perf_counter()
    sd = synth.make_synth(wname, years=list(range(2006, 2017)), n_samples=100, seed=args.seed,
                          extra_confounder_fips=60)
    ct = tables.compile_from_synth(sd)
    dt = tables.DeviceTables(ct, device)
    if rpath == "table":
        dt.build_logit_table(timed=True)
    env = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment, env_gid0=rank * n,
                          write_obs=not args.no_obs, reward_path=rpath, episode_order=args.episode_order,
                          lockstep=False if args.graph else None,
                          obs_dtype=torch.float16 if args.obs_f16 else torch.float32)
    gather = wdist.ReturnGatherer(n, device)
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    pool = [(torch.rand(n, device=device, generator=g) < 0.1).to(torch.int32) for _ in range(16)]
    env.reset(seed=args.seed)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    T = ct.T
    stepno = 0

    def one_step():
        nonlocal stepno
        env.step(pool[stepno & 15])
        stepno += 1
        if stepno % T == 0:  # lock step: every env just finished an episode
            gather.gather(env._final_return)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    graph = None
    if args.graph:
        # hipGraph of G consecutive steps (fixed action buffers); autoreset runs inside the kernel so that
        # the captured work is identical for every replay
        assert args.steps % args.graph == 0 and not env._host_auto, "--graph needs lockstep=False and steps % G == 0"
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            env.step(pool[0])
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            for i in range(args.graph):
                env.step(pool[i & 15])
        torch.cuda.synchronize()
    wdist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if graph is None:
        for _ in range(args.steps):
            one_step()
    else:
        for _ in range(args.steps // args.graph):
            graph.replay()
            before = stepno
            stepno += args.graph
            if stepno // T != before // T:
                gather.gather(env._final_return)
    ev1.record()
    torch.cuda.synchronize()
    wdist.barrier()
    wall = time.perf_counter() - t0
    wall = wdist.max_over_ranks(wall, device)
    dev_ms = ev0.elapsed_time(ev1)
    status = env.check_status()
    mean_ret = float(gather.mean(env._final_return).item())

    if rank == 0:
        total_env_steps = float(n) * world * args.steps
        per_launch_s = dev_ms * 1e-3 / args.steps
        bytes_per = ALGO_BYTES_NO_OBS if args.no_obs else ALGO_BYTES_PER_ENV_STEP
        if rpath == "table":
            bytes_per = ALGO_BYTES_TABLE - (116 if args.no_obs else 0)
        achieved = bytes_per * n / per_launch_s / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(args.workload)
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "env_steps_per_sec", "value": total_env_steps / wall, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64" , "data": "synthetic",
            "config": {"workload": desc, "num_envs_per_gpu": n, "num_envs_total": n * world,
                       "episode_days": T, "n_samples": ct.n_samples, "obs": not args.no_obs,
                       "arithmetic": "f32 tables, fp64 logit accumulation, f32 sigmoid/reward",
                       "seed_mode": "device", "autoreset": "same_step", "reward_path": rpath, "episode_order": args.episode_order, "hipgraph_steps": args.graph,
                       "obs_dtype": "f16" if args.obs_f16 else "f32",
                       "logit_table_build_ms": dt.logit_build_ms,
                       "logit_table_gb": None if dt.L is None else dt.L.numel() * 8 / 1e9,
                       "collective": "all_gather_into_tensor(f32[num_envs]) per episode" if world > 1 else "none"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": f"k_step<autoreset={env._dev_auto},obs={not args.no_obs},table={rpath == 'table'}>"
                                   + ("" if env._dev_auto else " + k_reset once per episode"),
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
                         "measured_traffic_gbs": None if traffic is None else traffic / per_launch_s / 1e9,
                         "note": "achieved uses SURVEY 8d algorithmic bytes (489 B: what the reference's step reads and "
                                 "writes). It can exceed the HBM peak because the day slice of X is served by L2 and W "
                                 "by the Infinity Cache, and because this kernel fetches the 112-B effectiveness row "
                                 "only on alert days (eff enters the reward through eff*actual); measured_traffic_gbs "
                                 "is what actually crossed the fabric",
                         "avg_launch_us": per_launch_s * 1e6,
                         "algorithmic_bytes_per_env_step": bytes_per,
                         "timing": "HIP events on the launch stream around the timed steps / steps"},
            "kernel_env_steps_per_sec_per_gpu": n / per_launch_s,
            "status_bits": status, "mean_final_return": mean_ret, "setup_s": t_setup,
        }
        if world == 1 and args.episode_order == "iid" and not args.graph and not args.no_extras:
            # reported extra (not the headline): the opt-in relabelled episode order, same workload
            env.close()
            e2 = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment, reward_path=rpath,
                                 write_obs=not args.no_obs, episode_order="sorted")
            e2.reset(seed=args.seed)
            for i in range(10):
                e2.step(pool[i & 15])
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            for i in range(T):
                e2.step(pool[i & 15])
            s1.record()
            torch.cuda.synchronize()
            us = s0.elapsed_time(s1) * 1e3 / T
            out["sorted_episode_order"] = {"ms_per_step": us * 1e-3, "value": n / us * 1e6, "unit": "env-steps/s",
                                           "note": "opt-in episode_order='sorted': same episode multiset, env indices "
                                                   "relabelled by table row after each reset (incl. the sort)"}
            e2.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, ct, args.seed)
            procs = min(16, os.cpu_count() or 1)
            try:
                out["cpu_baseline"]["multi_core"] = cpu_baseline_multicore(wname, args.seed, procs)
            except Exception as e:  # noqa: BLE001  (a reported extra, never fatal)
                out["cpu_baseline"]["multi_core"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    env.close()
    wdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
