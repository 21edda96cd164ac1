#!/usr/bin/env python3
"""Benchmark of the HeatAlertEnv hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload configs2] [--num-envs M]

A "step" is one vector-env step(): every env of the batch advances one day (budget gate,
feature-row gather, two 28-term logits against its posterior draw, reward, next observation),
including the same-step autoreset when an episode ends (every 153 steps, lock step) and, on
N>1 GPUs, the RCCL all-gather of the finished episodes' returns. Inputs (tables, state, a
pool of action tensors) are resident in HBM before the timed region.

`--gpus N` with N > 1 launches itself: when no launcher has set WORLD_SIZE, this process -- before it
imports torch or touches a GPU -- starts N children (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* set), waits for them and forwards rank 0's JSON line. Under torchrun (WORLD_SIZE set) it is
simply one of the ranks.

Prints ONE JSON line (rank 0). `roofline` prices the step kernel against HBM with the COMPULSORY bytes
of the variant that ran (what must cross HBM per env-step however good the caches are: actions and
per-env state in, reward / done / observation / state out -- 161 B with observations, 45 B without);
tables are not charged because every env of a lock-step batch shares one 1 MB day slice and the 19 MB
coefficient table lives in L2 / Infinity Cache. `roofline.traffic` is the fabric traffic the PMC counters
saw for the same kernel sources (profiles/traffic_latest.json, refused when the sources differ);
`cpu_baseline` times the NumPy oracle on the host.
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

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)
SURVEY_8D_BYTES = {"obs": 489, "no_obs": 373}  # SURVEY §8d's model (charges table rows to HBM): reported beside

WORKLOADS = {
    # name: (weights list, num_envs per GPU, similar_climate_counties, description)
    "configs1": ("linear", 65536, False,
                 "configs[1]: 65,536 envs, weights/linear (S=746), random county per env"),
    "configs2": ("linear", 1048576, True,
                 "configs[2]: 1,048,576 envs, weights/linear (S=746), similar_climate_counties=True"),
    "configs3": ("nn_full_medicare_all", 1048576, False,
                 "configs[3]: 1,048,576 envs, nn_full_medicare_all shape (S=720), random county per env"),
    "configs4": ("nn_full_medicare_all", 1048576, False,
                 "configs[4]: 1,048,576 envs per GPU, nn_full_medicare_all shape (S=720), RCCL all-gather of the "
                 "episodic returns once per episode"),
    "launcher_stub": (None, 4096, False,
                      "launcher / collective rehearsal without env stepping (no GPU needed; not a measurement)"),
}


def compulsory_bytes(n_obs: int, write_obs: bool, packed: bool = False) -> dict:
    """Bytes per env-step that must cross HBM for the row-gather step kernel. `packed`: the batch is in
    lock step and the kernel streams the 16-B packed mirror of the per-env state (8 + 8 B in, 8 B out) instead of the
    canonical words (12 + 12 in, 12 out)."""
    rd = {"action": 4, "pk_hot": 8, "pk_c": 8} if packed else {"action": 4, "hot3": 12, "stepc": 12}
    wr = {"reward": 4, "done": 1, "pk_hot": 8} if packed else {"reward": 4, "done": 1, "hot3": 12}
    if write_obs:
        wr["obs"] = 4 * n_obs
    return {"read": rd, "write": wr, "total": sum(rd.values()) + sum(wr.values())}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=1530, help="timed steps (default: 10 episodes, SURVEY §8d)")
    p.add_argument("--warmup", type=int, default=153, help="untimed steps (default: 1 episode)")
    p.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                   help="default: configs2 per GPU for every N (the config the >= 10 M target is quoted on: one workload for "
                        "the whole 1/2/4/8-GPU curve); with N > 1 the line also carries BASELINE configs[4] "
                        "(nn_full_medicare_all shape, 1 048 576 envs per GPU) measured in the same job: `configs4_sharded`")
    p.add_argument("--num-envs", type=int, default=None, help="envs per GPU (overrides the workload's)")
    p.add_argument("--no-obs", action="store_true", help="reward-only step variant")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true",
                   help="skip the reported extras (always-alert policy, sorted episode order, posterior-mean reward)")
    p.add_argument("--no-parity", action="store_true",
                   help="skip the oracle replay of a strided sample of the timed batch that follows the timed region")
    p.add_argument("--only-extras", default=None, metavar="NAME[,NAME...]",
                   help="single-GPU run: of the reported extras only these (always_alert, sorted, posterior_mean, rollout, "
                        "configs4, configs1)")
    p.add_argument("--no-live-traffic", action="store_true",
                   help="do not start the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE on tools/pmc_probe.py) that "
                        "measure roofline.traffic in this very run; the figure is then replayed from profiles/traffic_latest.json "
                        "when the kernel sources still hash to the recorded value")
    p.add_argument("--no-calibration", action="store_true",
                   help="skip the in-process copy-rate / access-pattern probe that follows the timed region")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="weak: the workload's env count per GPU (default); strong: that count split over the GPUs")
    p.add_argument("--graph", type=int, default=0,
                   help="capture this many consecutive step() calls into one hipGraph and replay it (launch-bound "
                        "small batches); the timed region still runs exactly --steps steps")
    p.add_argument("--episode-order", default="iid", choices=["iid", "sorted"],
                   help="sorted = opt-in relabelling of envs by table row after each reset (same episode multiset)")
    p.add_argument("--step-kernel", default="auto", choices=["auto", "classic", "wide", "unpacked"],
                   help="unpacked = auto without the packed lock-step mirror of the per-env state (A/B)")
    p.add_argument("--launch-timeout", type=float, default=1500.0,
                   help="--gpus N self-launcher: seconds after which still-running ranks are terminated")
    p.add_argument("--tamper", default=None, choices=["return", "obs"],
                   help="(test hook) after the timed region, falsify one finished episode's return / one observation value "
                        "of the timed batch: the parity leg must then fail and the exit code be 5")
    p.add_argument("--fail-rank", type=int, default=-1,
                   help="(launcher test hook) this rank exits with code 3 right after start-up")
    p.add_argument("--sweep", default=None, metavar="N1,N2,...",
                   help="scaling sweep in one command, e.g. --sweep 1,2,4,8: this process (which never touches a GPU) runs "
                        "`bench.py --gpus N` for every N in turn, each as a fresh group of child processes, and prints ONE "
                        "JSON line per N (value, per-GPU value, the job's own single-GPU reference, efficiency, what the "
                        "collective costs, ranks seen) plus a closing summary line. Every job runs the default workload (configs2 "
                        "per GPU, + configs4_sharded for N > 1) unless --workload says otherwise; --scaling weak (default) or strong")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="gloo = rehearsal of the multi-rank path (ranks share the GPUs there are; with the "
                        "launcher_stub workload it needs no GPU at all)")
    a = p.parse_args(argv)
    if a.sweep is not None:
        try:
            a.sweep = [int(x) for x in a.sweep.split(",") if x.strip()]
        except ValueError:
            p.error("--sweep takes a comma-separated list of GPU counts, e.g. 1,2,4,8")
        if not a.sweep or min(a.sweep) < 1:
            p.error("--sweep needs positive GPU counts")
    if a.workload is None:
        a.workload = "configs2"  # for every N: the driver computes scaling efficiency from the per-N values of ONE command
    return a


# ------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args) -> int:
    """`--gpus N` without a launcher: one child per GPU. This process has not imported torch and never touches a
    GPU; children are fresh interpreters (fork + exec of python), each of which initialises its own device.
    All children are polled: when one exits non-zero (or the deadline passes) the others are terminated and the
    failing rank's stderr tail is printed, so a rank that dies at start-up cannot leave the rest -- and the caller --
    waiting in a rendezvous."""
    import subprocess
    import tempfile

    n = args.gpus
    port = _free_port()
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = tempfile.TemporaryFile(mode="w+") if r == 0 else subprocess.DEVNULL
        err = tempfile.TemporaryFile(mode="w+")
        logs.append((out, err))
        procs.append(subprocess.Popen(cmd, env=env, stdout=out, stderr=err, text=True))
    deadline = time.time() + args.launch_timeout
    rc, failed = 0, None
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, rc = bad[0], codes[bad[0]]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            failed, rc = next(r for r, c in enumerate(codes) if c is None), 124
            print(f"bench.py: --launch-timeout {args.launch_timeout:.0f} s passed, rank {failed} still running",
                  file=sys.stderr, flush=True)
            break
        time.sleep(0.05)
    if failed is not None:  # children are plain child processes of ours: ending them is safe
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 5
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        logs[failed][1].seek(0)
        tail = logs[failed][1].read()[-2000:]
        print(f"bench.py: rank {failed} exited with code {rc}; the other ranks were terminated. Its stderr tail:\n{tail}",
              file=sys.stderr, flush=True)
        return rc if rc else 1
    for r, (_, err) in enumerate(logs):  # forward what the ranks wrote to stderr (warnings), rank by rank
        err.seek(0)
        txt = err.read()
        if txt:
            sys.stderr.write(txt)
    logs[0][0].seek(0)
    line = [ln for ln in logs[0][0].read().splitlines() if ln.startswith("{")]
    if line:
        print(line[-1], flush=True)
        return 0
    print("bench.py: rank 0 printed no JSON line", file=sys.stderr, flush=True)
    return 1


def run_sweep(args) -> int:
    """`--sweep N1,N2,...`: one `bench.py --gpus N` job per entry, one after another, each a fresh child (which launches
    its own ranks for N > 1): nothing of one job -- process group, RCCL communicators, device memory -- survives into the
    next. This process imports no torch and touches no GPU. One condensed JSON line per N as it finishes, then a summary
    line with the efficiencies relative to the sweep's own N = 1 (or smallest-N) line."""
    import subprocess

    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "per_gpu_value",
            "single_gpu_value", "weak_efficiency", "efficiency_vs_rank0_alone", "no_collective_value",
            "collective_overhead_frac", "collective_ms", "rccl_ranks_seen", "status_bits", "mean_final_return")
    lines, rc_all = [], 0
    for n in args.sweep:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), "--workload", args.workload, "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--scaling", args.scaling, "--backend", args.backend, "--seed", str(args.seed),
               "--launch-timeout", str(args.launch_timeout), "--no-cpu-baseline"] + ([] if n > 1 else ["--only-extras", "configs4"])
        if args.num_envs:
            cmd += ["--num-envs", str(args.num_envs)]
        if args.no_obs:
            cmd.append("--no-obs")
        if args.no_parity:
            cmd.append("--no-parity")
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
        t0 = time.time()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.launch_timeout + 120)
            out, err, rc = r.stdout, r.stderr, r.returncode
        except subprocess.TimeoutExpired as e:
            out, err, rc = (e.stdout or ""), (e.stderr or ""), 124
            out, err = (out.decode() if isinstance(out, bytes) else out), (err.decode() if isinstance(err, bytes) else err)
        js = [ln for ln in out.splitlines() if ln.startswith("{")]
        if rc != 0 or not js:
            rc_all = rc or 1
            line = {"sweep_n": n, "error": f"bench.py --gpus {n} exited with code {rc}", "stderr_tail": err[-1500:]}
        else:
            d = json.loads(js[-1])
            c4 = d.get("configs4_sharded") or d.get("configs4_single_gpu") or {}
            line = {"sweep_n": n, **{k: d.get(k) for k in keep}, "workload": (d.get("config") or {}).get("workload"),
                    "configs4_value": c4.get("value"), "configs4_kernel_us": c4.get("kernel_us"),
                    "num_envs_per_gpu": (d.get("config") or {}).get("num_envs_per_gpu"),
                    "parity_ok": (d.get("parity") or {}).get("ok"), "job_wall_s": round(time.time() - t0, 1)}
        lines.append(line)
        print(json.dumps(line), flush=True)
    good = [ln for ln in lines if "error" not in ln]
    summary = {"sweep": args.sweep, "scaling": args.scaling, "workload": args.workload, "failed": [ln["sweep_n"] for ln in lines if "error" in ln]}
    if good:
        base = min(good, key=lambda ln: ln["sweep_n"])
        # weak: per-GPU work fixed -> ideal value grows with N; strong: total work fixed -> ideal value grows with N too
        # (the same envs finish N times faster). Either way: value / (value at the smallest N x N / that N)
        summary["efficiency_vs_smallest_n"] = {
            str(ln["sweep_n"]): ln["value"] / (base["value"] * ln["sweep_n"] / base["sweep_n"]) for ln in good}
        summary["baseline_n"] = base["sweep_n"]
        c4 = [ln for ln in good if ln.get("configs4_value")]
        if c4:  # BASELINE configs[4] (nn_full_medicare_all shape), measured inside every job of the sweep
            b4 = min(c4, key=lambda ln: ln["sweep_n"])
            summary["configs4_efficiency_vs_smallest_n"] = {
                str(ln["sweep_n"]): ln["configs4_value"] / (b4["configs4_value"] * ln["sweep_n"] / b4["sweep_n"]) for ln in c4}
    print(json.dumps(summary), flush=True)
    return rc_all


# ------------------------------------------------------------------------------------------ CPU baseline
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

    # per-worker sample capped so that the leg stays short however many cores the host has (every worker also builds
    # its own tables, ~10 s, outside its timed region)
    n, steps = (65536, 154) if procs <= 32 else (16384, 154)
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
    return {"value": total / slowest, "unit": "env-steps/s", "cores": procs, "kind": "port",
            "sample": f"{procs} single-threaded processes (one per host core) x {n} envs x {steps - 1} steps of the "
                      f"NumPy vector oracle, slowest worker {slowest:.1f} s"}


def cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus() -> dict:
    """How many host CPUs this process may actually run on: the smaller of os.cpu_count(), the scheduler affinity mask
    and the cgroup CPU quota (cpu.max, v2; cfs_quota_us / cfs_period_us, v1). A container that shows 256 CPUs but is
    throttled to a 16-CPU quota can keep 16 single-threaded workers busy, not 256."""
    info = {"os_cpu_count": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity"] = info["os_cpu_count"]
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    info["cgroup_quota_cpus"] = quota
    usable = min(info["os_cpu_count"], info["affinity"])
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    info["usable"] = usable
    return info


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
    episodes, dt = 4, 0.0  # ~13 s of single-core NumPy work
    for ep in range(episodes):
        if ep:
            V.reset(ct.fips_to_weather[county].astype(np.int64), rng.integers(0, ct.Y, n), cc,
                    rng.integers(0, ct.n_samples, n), rng.integers(0, 12, n))
        V.step(acts[0])
        t0 = time.perf_counter()
        for t in range(1, steps):
            V.step(acts[t])
        dt += time.perf_counter() - t0
    vec = episodes * n * (steps - 1) / dt
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
        "sample": f"NumPy float64 vector oracle, {episodes} episodes of {n} envs x {steps - 1} steps of the same tables "
                  f"({dt:.1f} s)",
        "scalar_port_env_steps_per_s": scalar,
        "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
        "reference_env_steps_per_s_build_container": 646.0,  # BASELINE.md §2 (pandas env, 1 core; it cannot travel)
    }


# ------------------------------------------------------------------------------------------ stub workload
def run_stub(args, rank, world):
    """Launcher / collective rehearsal: everything bench.py does around the env (process group, barrier,
    per-episode return all-gather, max-over-ranks timing, one JSON line from rank 0) with a stand-in for the
    env's returns. Runs on CPU with --backend gloo; it measures nothing about the kernels."""
    import torch

    from weather2alert_amd import dist as wdist

    use_gpu = args.backend == "nccl"
    device = torch.device(f"cuda:{int(os.environ.get('LOCAL_RANK', '0'))}") if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    wdist.init_from_env(args.backend, device if use_gpu else None)
    n = args.num_envs or WORKLOADS["launcher_stub"][1]
    if args.scaling == "strong":  # a fixed total split over the ranks
        start, stop = wdist.shard_range(n, rank, world)
        if (stop - start) * world != n:
            raise SystemExit("--scaling strong needs an env count divisible by the number of GPUs")
        n = stop - start
    import torch.distributed as td

    seen = td.get_world_size() if td.is_initialized() else 1
    if seen != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the process group has {seen} rank(s)", file=sys.stderr, flush=True)
        return 4
    gather = wdist.ReturnGatherer(n, device)
    gid0 = rank * n
    returns = -(torch.arange(gid0, gid0 + n, dtype=torch.float32, device=device) % 97)
    T, coll = 153, []

    def loop(with_gather):
        t0 = time.perf_counter()
        for s in range(1, args.steps + 1):
            returns.add_(0.0)  # stands in for a step
            if with_gather and s % T == 0:
                c0 = time.perf_counter()
                gather.gather(returns)
                coll.append(time.perf_counter() - c0)
        return time.perf_counter() - t0

    single = None
    if world > 1:  # rank 0 alone, the others idle at the barrier (the real workload's `single_gpu_value`)
        wdist.barrier()
        if rank == 0:
            single = float(n) * args.steps / max(loop(False), 1e-9)
        wdist.barrier()
    wdist.barrier()
    wall = loop(True)
    wdist.barrier()
    wall = wdist.max_over_ranks(wall, device)
    no_coll = None
    if world > 1:
        wdist.barrier()
        w2 = loop(False)
        wdist.barrier()
        no_coll = wdist.max_over_ranks(w2, device)
    allr = gather.gather(returns)
    expect = -(torch.arange(0, n * world, dtype=torch.float32, device=device) % 97)
    ok = bool(torch.equal(allr, expect))
    if rank == 0:
        total = float(n) * world * args.steps
        print(json.dumps({
            "metric": "env_steps_per_sec", "value": total / wall, "unit": "stub steps/s (no env work)", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / max(args.steps, 1), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "n/a", "data": "stub",
            "config": {"workload": WORKLOADS["launcher_stub"][3], "num_envs_per_gpu": n, "num_envs_total": n * world,
                       "backend": args.backend},
            "per_gpu_value": total / wall / world, "single_gpu_value": single,
            "efficiency_vs_rank0_alone": (total / wall / world / single) if single else None,
            "weak_efficiency": (total / wall / world / single) if (single and args.scaling == "weak") else None,
            "no_collective_value": (total / no_coll) if no_coll else None,
            "collective_overhead_frac": ((wall - no_coll) / wall) if no_coll else None,
            "rccl_ranks_seen": seen, "collective_ms": (sum(coll) / len(coll) * 1e3) if coll else None,
            "gather_ok": ok, "stub": True}), flush=True)
    wdist.barrier()
    if td.is_initialized():
        td.destroy_process_group()
    return 0 if ok else 1


# ------------------------------------------------------------------------------------------ box calibration
def box_calibration(env, dt, ct, packed, torch):
    """What THIS box delivers in THIS process, measured right before the run it is compared with (the GPU boxes of one
    pool differ by 7-10 % on the same command): the rate of a plain float4 copy of 1 GiB, and the time of an arithmetic-
    free program that issues the step kernel's traffic on the env's own tables and its own episode tuples
    (tools/fabric_probe.hip as a library: streams only / gathers only / both). Measurement code, not part of the env."""
    import ctypes as C

    from weather2alert_amd import build as wbuild

    out = {"source": "tools/fabric_probe.hip " + " ".join(wbuild.PROBE_FLAGS) + (" -DPROBE_PACKED" if packed else "") +
                     ", in this process, on this run's tables and episode tuples, right after the timed region"}
    try:
        lib = C.CDLL(wbuild.build_probe_lib(packed=packed))
        lib.w2a_probe_last_error.restype = C.c_char_p
        lib.w2a_probe_copy.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_float), C.c_void_p]
        lib.w2a_probe_step_pattern.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
        stream = torch.cuda.current_stream().cuda_stream
        gbs = (C.c_float * 3)()
        if lib.w2a_probe_copy(1 << 30, 5, gbs, stream) != 0:
            raise RuntimeError(lib.w2a_probe_last_error().decode())
        out["copy_gbs"] = float(gbs[0])
        out["copy_gbs_variants"] = {"float4_kernel_one_element_per_thread": float(gbs[1]), "hipMemcpyAsync_d2d": float(gbs[2])}
        st = env.state()
        xrow = (st["county_w"] * ct.Y + st["year_i"]).to(torch.int32).contiguous()
        wrow = (st["coef_col"] * ct.n_samples + st["sample"]).to(torch.int32).contiguous()
        n256 = env.num_envs // 256 * 256
        if n256 and ct.n_obs == 29:
            us = (C.c_float * 3)()
            if lib.w2a_probe_step_pattern(dt.X.data_ptr(), ct.S_w * ct.Y, ct.T, dt.W.data_ptr(), xrow.data_ptr(), wrow.data_ptr(),
                                          n256, 3, 100, us, stream) != 0:
                raise RuntimeError(lib.w2a_probe_last_error().decode())
            out["probe_us"] = {"streams_only": float(us[0]), "gathers_only": float(us[1]), "streams_and_gathers": float(us[2]),
                               "envs": n256}
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001  (a reported calibration, never fatal)
        out["error"] = repr(e)
    return out


def sharded_workload(wl, args, rank, world, device, torch, HeatAlertVecEnv, synth, tables, wdist):
    """One more workload of this very job, sharded like the headline (N > 1): BASELINE configs[4] = nn_full_medicare_all
    shape, 1 048 576 envs per GPU, global env ids rank * n ..., the return all-gather once per episode -- W warm-up steps,
    then exactly K steps between barrier + device synchronisation on both sides, MAX over ranks. Returns rank 0's dict."""
    wname, n_default, augment, desc = WORKLOADS[wl]
    n = args.num_envs or n_default
    if args.scaling == "strong":
        start, stop = wdist.shard_range(n, rank, world)
        if (stop - start) * world != n:
            return {"skipped": f"{n} envs do not divide by {world} ranks"}
        n = stop - start
    sd = synth.make_synth(wname, years=list(range(2006, 2017)), n_samples=100, seed=args.seed, extra_confounder_fips=60)
    ct = tables.compile_from_synth(sd)
    env = HeatAlertVecEnv(n, tables=tables.DeviceTables(ct, device), device=device, similar_climate_counties=augment,
                          env_gid0=rank * n, write_obs=not args.no_obs)
    gather = wdist.ReturnGatherer(n, device)
    g = torch.Generator(device=device).manual_seed(4321 + rank)
    pool = [(torch.rand(n, device=device, generator=g) < 0.1).to(torch.int32) for _ in range(16)]
    env.reset(seed=args.seed)
    T, stepno = ct.T, 0

    def one():
        nonlocal stepno
        env.step(pool[stepno & 15])
        stepno += 1
        if stepno % T == 0:
            gather.gather(env._final_return, async_op=True)

    for _ in range(args.warmup):
        one()
    gather.wait()
    wdist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        one()
    gather.wait()
    e1.record()
    torch.cuda.synchronize()
    wdist.barrier()
    wall = wdist.max_over_ranks(time.perf_counter() - t0, device)
    status = env.check_status()
    packed = env.packed_state
    mean_ret = float(gather.mean(env._final_return).item())
    env.close()
    total = float(n) * world * args.steps
    cb = compulsory_bytes(ct.n_obs, not args.no_obs, packed)
    return {"workload": desc, "value": total / wall, "unit": "env-steps/s", "per_gpu_value": total / wall / world,
            "num_envs_per_gpu": n, "num_envs_total": n * world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps, "device_ms_per_step_rank0": e0.elapsed_time(e1) / args.steps,
            "packed_lockstep_state": packed, "compulsory_bytes_per_env_step": cb["total"], "status_bits": status,
            "mean_final_return": mean_ret,
            "collective": "all_gather_into_tensor(f32[num_envs_per_gpu]) per episode, overlapped with the next episode's steps",
            "note": "BASELINE configs[4] inside the same job as the headline (which keeps configs[2] per GPU for every N, so "
                    "that one command gives the 1/2/4/8 curve on one workload); its one-GPU figure is `configs4_single_gpu` "
                    "of the N = 1 line"}


# ------------------------------------------------------------------------------------------ fabric traffic, live
def live_traffic(workload: str, n: int, no_obs: bool, step_kernel: str, episode_order: str, variant: str, timeout_s: int = 90):
    """roofline.traffic measured in THIS run: two child processes `rocprofv3 --pmc <counter> --kernel-trace -- python3
    tools/pmc_probe.py ...` (FETCH_SIZE, then WRITE_SIZE: separate passes, MI355X_MICROARCH.md HBM section), started after
    every timing of this process is over and its env is closed -- a counter pass cannot run inside a process that is
    already up. The probe steps the same workload on the same library and, in the same process, runs a 1 GiB copy whose
    byte count is known: FETCH_SIZE is calibrated on it (gfx950 reports half the bytes of 16-B-per-lane reads), WRITE_SIZE
    is exact. Returns (entry like those of profiles/traffic_latest.json, None) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this process is itself running under rocprofv3: no nested counter passes"
    roc = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(roc):
        return None, "rocprofv3 not found"
    base = tempfile.mkdtemp(prefix="w2a_pmc_", dir=os.environ.get("TMPDIR") if os.path.isdir(os.environ.get("TMPDIR", "")) else "/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    res, per_pass_s = {}, {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(base, ctr)
            cmd = [roc, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3",
                   os.path.join(ROOT, "tools", "pmc_probe.py"), "--workload", workload, "--num-envs", str(n), "--steps", "16",
                   "--step-kernel", step_kernel, "--episode-order", episode_order] + (["--no-obs"] if no_obs else [])
            t0 = time.perf_counter()
            # its own process group: a pass that hangs is ended WITH the probe it started, not only the profiler's launcher
            pr = subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                  start_new_session=True)
            try:
                out_s, err_s = pr.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                import signal

                os.killpg(pr.pid, signal.SIGKILL)
                pr.communicate()
                return None, f"rocprofv3 --pmc {ctr} pass did not finish within {timeout_s} s"
            per_pass_s[ctr] = round(time.perf_counter() - t0, 1)
            if pr.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} pass exited {pr.returncode}: {(err_s or out_s)[-300:]}"
            per = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr:
                        key = (row["Dispatch_Id"], row["Kernel_Name"])
                        per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
            step = [v for (_, k), v in per.items() if (variant + "<") in k]
            copy = [v for (_, k), v in per.items() if "copyBuffer" in k and v > 1e5]
            if not step or not copy:
                return None, f"the {ctr} pass saw {len(step)} {variant} launches and {len(copy)} calibration copies"
            res[ctr] = (sum(step) / len(step), sum(copy) / len(copy), len(step))
        gib = float(1 << 30)
        rd_corr, wr_corr = gib / (res["FETCH_SIZE"][1] * 1024), gib / (res["WRITE_SIZE"][1] * 1024)
        rd, wr = res["FETCH_SIZE"][0] * 1024 * rd_corr, res["WRITE_SIZE"][0] * 1024 * wr_corr
        return {"bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                "read_correction": rd_corr, "write_correction": wr_corr, "launches_averaged": res["FETCH_SIZE"][2],
                "pass_seconds": per_pass_s,
                "profile": "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of tools/pmc_probe.py in this run"}, None
    except Exception as e:  # noqa: BLE001  (reported, never loses the headline line)
        return None, repr(e)
    finally:
        shutil.rmtree(base, ignore_errors=True)


# ------------------------------------------------------------------------------------------ parity of the timed batch
def parity_check(env, sd, ct, pool, act_log, torch, extra_steps=8, max_sample=4096):
    """Does the batch that was just TIMED hold what the reference would hold? (cpu-baseline leg: the oracle is the checker,
    outside the timed region.) For a strided sample of envs (env 0 and the last one included):
      * the episode tuples the device RNG drew for the current episode -- and the budget stickiness chain from episode 0 --
        against the NumPy restatement of the draw (oracle/sequence_model.draw_episodes);
      * the last FINISHED episode replayed day by day on the float64 VectorOracle from the recorded action tensors
        (act_log: which of the 16 pool tensors every step used): its return against the env's final_return;
      * the current episode replayed up to today: integer state bit-exact, the observation rows the env holds bit-exact,
        the running return, and the reward of the last timed step;
      * `extra_steps` further steps of env and oracle side by side: every reward, observation row and done flag.
    Returns the `parity` object of the JSON line."""
    import numpy as np

    from oracle import heatalert_oracle as O
    from oracle import sequence_model as SM

    n, T = env.num_envs, ct.T
    idx = np.unique(np.concatenate([np.arange(0, n, max(n // max_sample, 1)), [n - 1]]))
    it = torch.as_tensor(idx, device=env.device)
    out = {"sampled": int(len(idx)), "reward_tol": 1e-5, "extra_checked_steps": extra_steps}
    st = {k: v[it].cpu().numpy() for k, v in env.state().items()}
    V = O.VectorOracle(O.RefData.from_synth(sd), sd.fips_weather, sd.years)
    pool_np = [p[it].cpu().numpy().astype(np.int64) for p in pool]
    ints_ok, notes = True, []

    def same(name, got, want):
        nonlocal ints_ok
        if not np.array_equal(np.asarray(got), np.asarray(want)):
            ints_ok = False
            notes.append(f"{name} differs for {int((np.asarray(got) != np.asarray(want)).sum())} sampled envs")

    e_cur = st["episode_no"]
    lock = len(np.unique(e_cur)) == 1 and len(np.unique(st["t"])) == 1 and not st["finished"].any()
    if not lock or env.seed_mode != "device" or env._reset_cfg is None:
        return {**out, "skipped": "the batch is not in lock step on device-RNG episodes: no common history to replay"}
    e_cur, t_cur = int(e_cur[0]), int(st["t"][0])
    # ---- episode tuples: the restated device RNG, with the sticky-budget chain from episode 0 (env.py:167-170, Q9)
    tuples = {}
    if env.episode_order == "iid":
        def chain(sticky0):
            sticky, got = sticky0.copy(), {}
            for ep in range(e_cur + 1):
                cw, yi, cc, sm, b, sticky = SM.draw_episodes(ct, env._reset_cfg, env.env_gid0 + idx, np.full(len(idx), ep), sticky,
                                                             "augment" in env.fixes)
                if ep >= e_cur - 1:
                    got[ep] = (cw, yi, cc, sm, b)
            return got

        tuples = chain(np.full(len(idx), -1, np.int64))
        out["tuples"] = "restated device RNG, sticky-budget chain from episode 0"
        if not np.array_equal(tuples[e_cur][4], st["budget"]) and env._reset_cfg[4] == 0:
            # a budget sticks to an env for good (env.py:167-170, Q9): on an env that had been reset before this seed (the
            # multi-GPU flow re-seeds after rank 0's solo run) the chain starts from the budgets that stuck back then
            tuples = chain(st["sticky_budget"].astype(np.int64))
            out["tuples"] = "restated device RNG; budgets = the sticky budgets the envs carried into this seed (Q9)"
        for name, want in zip(("county_w", "year_i", "coef_col", "sample", "budget"), tuples[e_cur]):
            same(f"episode tuple ({name})", st[name], want)
    else:  # episode_order='sorted' relabels envs after every reset: the tuples are taken from the state itself
        tuples[e_cur] = tuple(st[k].astype(np.int64) for k in ("county_w", "year_i", "coef_col", "sample", "budget"))
        out["tuples"] = "read back from the env (sorted order relabels env indices)"
    n_logged = len(act_log)
    worst_r, worst_ret, steps_replayed, episodes = 0.0, 0.0, 0, []
    # ---- the last finished episode: its return
    last_r = None
    if e_cur - 1 in tuples and n_logged >= t_cur + T:
        V.reset(*tuples[e_cur - 1])
        ret = np.zeros(len(idx))
        for k in range(T):
            _, r, done, _ = V.step(pool_np[act_log[n_logged - t_cur - T + k]])
            ret += r
            last_r = r
        assert done.all()
        fr = env._final_return[it].cpu().numpy().astype(np.float64)
        err = np.abs(fr - ret)
        worst_ret = float(np.max(err / (2e-6 * np.abs(ret) + 2e-5)))  # in units of the suite's return tolerance
        out["final_return_max_abs_err"] = float(err.max())
        steps_replayed += T
        episodes.append(e_cur - 1)
    # ---- the current episode up to today
    obs_o = V.reset(*tuples[e_cur])
    ret = np.zeros(len(idx))
    for k in range(t_cur):
        obs_o, r, done, _ = V.step(pool_np[act_log[n_logged - t_cur + k]])
        ret += r
        last_r = r
    steps_replayed += t_cur
    episodes.append(e_cur)
    same("t", st["t"], V.t); same("used", st["used"], V.used); same("streak", st["streak"], V.streak)
    same("remaining budget", st["budget"] - st["used"], V.budget - V.used)
    same("last_actual", st["last_actual"], V.last_actual)
    hist14 = np.zeros(len(idx), np.int64)
    for k in range(14):
        hist14 |= V.hist[:, 13 - k].astype(np.int64) << k
    same("14-day history", st["hist14"], hist14)
    obs_ok = None
    if env.write_obs:
        obs_ok = bool(np.array_equal(env._obs[it].cpu().numpy(), obs_o.astype(np.float32)))
    err = np.abs(st["episode_return"].astype(np.float64) - ret)
    out["episode_return_max_abs_err"] = float(err.max())
    worst_ret = max(worst_ret, float(np.max(err / (2e-6 * np.abs(ret) + 2e-5))))
    if last_r is not None:  # the reward buffer still holds the last timed step's rewards
        worst_r = float(np.abs(env._reward[it].cpu().numpy().astype(np.float64) - last_r).max())
    # ---- a few more steps side by side: every reward, row and flag
    done_ok = True
    for k in range(extra_steps):
        if t_cur + k >= T - 1:
            break  # not across the terminal step: the env's own autoreset would follow
        a = pool[k & 15]
        obs, r, done, _, _ = env.step(a)
        obs_o, r_o, done_o, _ = V.step(pool_np[k & 15])
        worst_r = max(worst_r, float(np.abs(r[it].cpu().numpy().astype(np.float64) - r_o).max()))
        done_ok = done_ok and bool(np.array_equal(done[it].cpu().numpy(), done_o))
        if env.write_obs:
            obs_ok = obs_ok and bool(np.array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32)))
        steps_replayed += 1
    if not done_ok:
        ints_ok = False
        notes.append("done flags differ")
    out.update(episodes_replayed=episodes, env_steps_replayed_per_env=steps_replayed, max_abs_reward_err=worst_r,
               return_err_over_tolerance=worst_ret, ints_exact=bool(ints_ok), obs_exact=obs_ok,
               status_bits=env.check_status(),
               ok=bool(ints_ok and worst_r <= 1e-5 and worst_ret <= 1.0 and obs_ok is not False), notes=notes,
               checker="oracle/heatalert_oracle.VectorOracle (float64) + oracle/sequence_model.draw_episodes, after the timed region")
    return out


# ------------------------------------------------------------------------------------------ main
def timed_steps(env, pool, steps, torch):
    """(device ms, wall s) of `steps` back-to-back step() calls (no autoreset boundary inside)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        env.step(pool[i & 15])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), time.perf_counter() - t0


def extras(out, args, torch, HeatAlertVecEnv, synth, tables, dt, ct, device, n, augment, pool, cb, T, wname_main=None):
    """Reported extras of the single-GPU run, each measured in the same process after the headline; a failure in one of
    them is recorded under its key and never loses the headline JSON line."""

    def guarded(key, fn):
        try:
            fn()
        except Exception as e:  # noqa: BLE001
            out[key] = {"error": repr(e)}
            try:  # after a device-side error the synchronisation raises again: it must not leave this handler,
                torch.cuda.synchronize()  # or the headline line the wrapper exists to protect would be lost
            except Exception as e2:  # noqa: BLE001
                out[key]["sync_error"] = repr(e2)

    def always_alert():
        # policy-pessimistic case: every env alerts every day with budget 153, so both coefficient rows are fetched on
        # every env-step (the headline policy fetches the second one on ~6 %)
        e2 = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment,
                             write_obs=not args.no_obs, budget=T, step_kernel=args.step_kernel)
        e2.reset(seed=args.seed)
        ones = [torch.ones(n, dtype=torch.int32, device=device)] * 16
        timed_steps(e2, ones, 10, torch)
        kms, _ = timed_steps(e2, ones, 130, torch)
        us = kms * 1e3 / 130
        out["always_alert_policy"] = {
            "kernel_us": us, "value": n / us * 1e6, "unit": "env-steps/s (kernel)",
            "roofline_frac": cb["total"] * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "note": "budget=153 and action=1 every day: both 128-B coefficient rows are gathered on every "
                    "env-step (worst case for the policy-dependent effectiveness-row skip)"}
        e2.close()

    def sorted_order():
        e3 = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment,
                             write_obs=not args.no_obs, episode_order="sorted", step_kernel=args.step_kernel)
        e3.reset(seed=args.seed)
        timed_steps(e3, pool, 10, torch)
        kms, _ = timed_steps(e3, pool, 130, torch)
        us = kms * 1e3 / 130
        res = {"kernel_us": us, "value": n / us * 1e6, "unit": "env-steps/s (kernel)",
               "roofline_frac": cb["total"] * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
               "note": "opt-in episode_order='sorted': same episode multiset, env indices relabelled by table "
                       "row after each reset"}
        # END TO END: whole episodes by the wall clock -- every step() call, the reset kernel after each terminal step and,
        # in sorted mode, the relabelling that follows it (counting sort by coefficient row, state permutation, first
        # observations) -- for the sorted order and, in the same way on a fresh env, for the default iid order
        def whole_episodes(e, episodes=3):
            for _ in range(T - (e._steps_in_episode % T)):  # to the next episode boundary, untimed
                e.step(pool[0])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(episodes * T):
                e.step(pool[i & 15])
            torch.cuda.synchronize()
            return n * episodes * T / (time.perf_counter() - t0)

        res["value_end_to_end"] = whole_episodes(e3)
        e3.close()
        e3 = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment,
                             write_obs=not args.no_obs, step_kernel=args.step_kernel)
        e3.reset(seed=args.seed)
        res["iid_value_end_to_end"] = whole_episodes(e3)
        res["end_to_end_gain"] = res["value_end_to_end"] / res["iid_value_end_to_end"]
        res["end_to_end_note"] = ("env-steps/s by the wall clock over 3 whole episodes, resets and (sorted) the per-episode "
                                  "relabelling included; iid = the default order measured the same way in the same process")
        out["sorted_episode_order"] = res
        e3.close()

    def posterior_mean():
        # reward_mode="posterior_mean": the reward of every env-step as the mean over all 100 posterior draws (the dense
        # "nn_full_medicare reward GEMM" of BASELINE configs[3]/[4]). Every kernel of the library in the same run:
        # the vector-ALU form and the matrix-core (MFMA) forms, selected at run time; "auto" keeps the faster one.
        flop_base = 2.0 * 28 * ct.n_samples * n  # baseline head: 28 coefficients x draws, multiply-add
        res = {"unit": "env-steps/s (pre-pass + reward kernel + k_step64<given>)", "kernels": {}}
        from weather2alert_amd import _ffi

        for name in _ffi.PM_KERNELS:
            e4 = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment,
                                 write_obs=not args.no_obs, reward_mode="posterior_mean", pm_kernel=name)
            e4.reset(seed=args.seed)
            timed_steps(e4, pool, 5, torch)
            kms, _ = timed_steps(e4, pool, 40, torch)
            us = kms * 1e3 / 40
            # the reward kernel alone (HIP events around the C-ABI call; includes its 17-us pre-pass)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                _ffi.check(e4._lib.w2a_posterior_mean_reward(e4._h, pool[0].data_ptr(), _ffi.ACT_I32, e4._rew_ptr,
                                                             e4._stream()), "w2a_posterior_mean_reward")
            e1.record()
            torch.cuda.synchronize()
            res["kernels"][name] = {"us_per_step": us, "value": n / us * 1e6,
                                    "reward_kernels_us": e0.elapsed_time(e1) * 1e3 / 10,
                                    "tflops_baseline_head": flop_base / (us * 1e-6) / 1e12}
            e4.close()
        ea = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment,
                             write_obs=not args.no_obs, reward_mode="posterior_mean", pm_kernel="auto")
        ea.reset(seed=args.seed)
        res["auto_choice"] = ea.pm_kernel_choice
        res["auto_timing_us"] = ea.pm_kernel_timing_us
        # the evaluation sweep of the legacy eval mode: a whole 153-day episode per launch with this reward
        rpol = dict(kind="threshold", feature="heat_qi", threshold=0.9, require_budget=True)
        ea.rollout(rpol)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            ea.rollout(rpol)
        torch.cuda.synchronize()
        dt_r = (time.perf_counter() - t0) / 2
        res["rollout"] = {"kernel": ea.pm_kernel_choice, "ms_per_episode": dt_r * 1e3, "value": n * ct.T / dt_r,
                          "unit": "env-steps/s", "note": "rollout(): threshold policy, one launch per 153-day episode "
                          "(k_pm_rollout_i8 / k_pm_rollout), grouping by column included"}
        ea.close()
        # matrix-core counters of the same kernels from their own rocprofv3 --pmc passes (profiles/r03/, collected by
        # tools/gpu_session.sh prof:configs2:pm_<kernel>; a PMC pass cannot run inside this process)
        for name in res["kernels"]:
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", "r03", f"pmc_configs2_pm_{name}.json")))
                e = next(v for k, v in pj["pmc"].items() if "k_posterior_mean" in k)
                busy = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
                res["kernels"][name]["rocprof"] = {
                    "kernel_avg_us": e.get("avg_us"), "SQ_INSTS_MFMA": e.get("SQ_INSTS_MFMA"),
                    "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_INSTS_VALU": e.get("SQ_INSTS_VALU"),
                    "mfma_busy_frac": busy / (1024 * e["avg_us"] * 2400.0) if e.get("avg_us") else None,
                    "note": "busy cycles / (1024 SIMDs x kernel time x 2.4 GHz); workload configs2, 1 048 576 envs"}
            except Exception:  # noqa: BLE001
                pass
        best = min(res["kernels"], key=lambda k: res["kernels"][k]["us_per_step"])
        res.update(us_per_step=res["kernels"][best]["us_per_step"], value=res["kernels"][best]["value"], fastest=best,
                   vector_peak_tflops_fp64=78.6,
                   mfma_counters="profiles/r03/pm_counters_*.json (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES per kernel)",
                   note="per step: [envs of a column x 28 slots + bias] x [draws], sigmoid / gate / mean epilogue; the "
                        "effectiveness head only for rows with an open-gate alert; time includes the pre-pass and the "
                        "step kernel that consumes the reward")
        out["posterior_mean_reward"] = res

    def rollout():
        # sampled reward, policy evaluated in the kernel, a whole episode per launch: the matrix-core kernel
        # (k_rollout_mfma: table part of both logits as int8 MFMAs per (county, year) tile) and the vector one (k_rollout64)
        rpol = dict(kind="threshold", feature="heat_qi", threshold=0.9, require_budget=True)
        res = {}
        for name, mfma in (("matrix_i8", True), ("vector", False)):
            e6 = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment, rollout_mfma=mfma)
            e6.reset(seed=args.seed)
            e6.rollout(rpol)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                e6.rollout(rpol)
            torch.cuda.synchronize()
            dt_r = (time.perf_counter() - t0) / 5
            res[name] = {"ms_per_episode": dt_r * 1e3, "value": n * ct.T / dt_r}
            e6.close()
        best = min(res, key=lambda k: res[k]["ms_per_episode"])
        out["on_device_rollout"] = {
            "ms_per_episode": res[best]["ms_per_episode"], "value": res[best]["value"], "unit": "env-steps/s",
            "kernel": best, "kernels": res,
            "note": "threshold policy evaluated in the kernel, 153 days per launch, envs visited in feature-row "
                    "order (w2a_rollout_order + tile list, included in the time), no observations written"}

    def configs4():
        # the single-GPU rate of the multi-GPU default workload (configs[4] = nn_full_medicare_all shape), so that
        # `--gpus N` values have their own N = 1 denominator in this file
        w4, n4, aug4, _ = WORKLOADS["configs4"]
        sd4 = synth.make_synth(w4, years=list(range(2006, 2017)), n_samples=100, seed=args.seed,
                               extra_confounder_fips=60)
        e5 = HeatAlertVecEnv(n4, tables=tables.compile_from_synth(sd4), device=device,
                             similar_climate_counties=aug4, write_obs=not args.no_obs)
        e5.reset(seed=args.seed)
        g4 = torch.Generator(device=device).manual_seed(4321)
        pool4 = pool if n4 == n else [(torch.rand(n4, device=device, generator=g4) < 0.1).to(torch.int32)
                                      for _ in range(16)]
        timed_steps(e5, pool4, 10, torch)
        kms, kwall = timed_steps(e5, pool4, 130, torch)
        out["configs4_single_gpu"] = {
            "kernel_us": kms * 1e3 / 130, "value": n4 * 130 / kwall, "unit": "env-steps/s",
            "roofline_frac": cb["total"] * n4 / (kms * 1e-3 / 130) / 1e9 / HBM_PEAK_GBS,
            "note": "python bench.py --gpus N (N > 1) runs this workload per GPU: compare its values with "
                    "N x this one, not with the configs[2] headline above"}
        e5.close()

    def configs1():
        # BASELINE configs[1] in the driver's own line: 65 536 envs, weights/linear, random county per env. Below 131 072
        # envs w2a_step picks the 4-lanes-per-env kernel (k_step, canonical 161-B state: more, shorter waves hide the two
        # dependent memory hops better); at 5 us per launch the host's launch rate matters, so the same steps are also
        # timed as a hipGraph of G recorded steps (in-kernel autoreset, so that every replay is the same work)
        w1, n1, aug1, desc1 = WORKLOADS["configs1"]
        if wname_main == w1:
            dt1, ct1 = dt, ct
        else:
            sd1 = synth.make_synth(w1, years=list(range(2006, 2017)), n_samples=100, seed=args.seed, extra_confounder_fips=60)
            ct1 = tables.compile_from_synth(sd1)
            dt1 = tables.DeviceTables(ct1, device)
        g1 = torch.Generator(device=device).manual_seed(4321)
        pool1 = [(torch.rand(n1, device=device, generator=g1) < 0.1).to(torch.int32) for _ in range(16)]
        e1 = HeatAlertVecEnv(n1, tables=dt1, device=device, similar_climate_counties=aug1, write_obs=not args.no_obs)
        e1.reset(seed=args.seed)
        timed_steps(e1, pool1, 20, torch)
        kms, kwall = timed_steps(e1, pool1, 120, torch)
        us = kms * 1e3 / 120
        cb1 = compulsory_bytes(ct1.n_obs, not args.no_obs, e1.packed_state)
        res = {"workload": desc1, "step_kernel": e1.last_step_kernel, "kernel_us": us, "value": n1 * 120 / kwall,
               "unit": "env-steps/s (eager step() loop, wall clock)", "kernel_env_steps_per_s": n1 / us * 1e6,
               "compulsory_bytes_per_env_step": cb1["total"],
               "roofline_frac": cb1["total"] * n1 / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
               "note": "kernel_us = HIP events around 120 back-to-back step() calls of one episode / 120: at this size it "
                       "contains the launch gaps the host leaves (the kernel alone: profiles/, rocprofv3 --kernel-trace)"}
        e1.close()
        # the same steps as one hipGraph per G = 51 days (three replays = one 153-day episode)
        G, reps = 51, 30
        e2 = HeatAlertVecEnv(n1, tables=dt1, device=device, similar_climate_counties=aug1, write_obs=not args.no_obs,
                             lockstep=False)
        e2.reset(seed=args.seed)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            e2.step(pool1[0])
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for i in range(G):
                e2.step(pool1[i & 15])
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(reps):
            graph.replay()
        ev1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        gus = ev0.elapsed_time(ev1) * 1e3 / (reps * G)
        cb2 = compulsory_bytes(ct1.n_obs, not args.no_obs, False)
        res["hipgraph"] = {"steps_per_graph": G, "replays": reps, "us_per_step": gus, "value": n1 * G * reps / wall,
                           "unit": "env-steps/s (wall clock)", "step_kernel": e2.last_step_kernel + " (in-kernel autoreset)",
                           "roofline_frac": cb2["total"] * n1 / (gus * 1e-6) / 1e9 / HBM_PEAK_GBS,
                           "status_bits": e2.check_status()}
        e2.close()
        out["configs1_single_gpu"] = res

    only = None if not args.only_extras else set(args.only_extras.split(","))
    want = lambda name: only is None or name in only  # noqa: E731
    if want("always_alert"):
        guarded("always_alert_policy", always_alert)
    if want("sorted"):
        guarded("sorted_episode_order", sorted_order)
    if want("posterior_mean"):
        guarded("posterior_mean_reward", posterior_mean)
    if want("rollout"):
        guarded("on_device_rollout", rollout)
    if args.workload == "configs2" and want("configs4"):
        guarded("configs4_single_gpu", configs4)
    if args.workload != "configs1" and want("configs1"):
        guarded("configs1_single_gpu", configs1)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        print(json.dumps(_cpu_worker(tuple(json.loads(sys.argv[2])))))
        return 0
    args = parse()
    if args.sweep:
        return run_sweep(args)  # never imports torch, never touches a GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)  # before anything imports torch / initialises a GPU
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("W2A_BENCH_TEST_HANG_RANK") == str(rank):  # launcher test hook: a rank that never finishes
        time.sleep(600)
    if rank == args.fail_rank:
        print(f"rank {rank}: --fail-rank test hook, exiting 3", file=sys.stderr, flush=True)
        return 3
    if args.workload == "launcher_stub":
        return run_stub(args, rank, world)

    import numpy as np
    import torch

    from weather2alert_amd import HeatAlertVecEnv, build as wbuild, dist as wdist, synth, tables

    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (use --workload launcher_stub --backend gloo to " \
                                      "rehearse the launcher without one)"
    device = torch.device(f"cuda:{local % torch.cuda.device_count() if args.backend == 'gloo' else local}")
    torch.cuda.set_device(device)
    wdist.init_from_env(args.backend, device)
    seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    if seen != args.gpus:  # a group smaller than asked for would be benchmarked as if it were whole
        print(f"bench.py: --gpus {args.gpus} but the process group has {seen} rank(s)", file=sys.stderr, flush=True)
        return 4
    if local == 0:  # one rank per node (re)builds libw2a.so if its sources are newer; the others wait and load it
        wbuild.build_lib()
    wdist.barrier()

    wname, n_default, augment, desc = WORKLOADS[args.workload]
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
    env = HeatAlertVecEnv(n, tables=dt, device=device, similar_climate_counties=augment, env_gid0=rank * n,
                          write_obs=not args.no_obs, episode_order=args.episode_order,
                          lockstep=False if args.graph else None, step_kernel=args.step_kernel)
    gather = wdist.ReturnGatherer(n, device)
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    pool = [(torch.rand(n, device=device, generator=g) < 0.1).to(torch.int32) for _ in range(16)]
    env.reset(seed=args.seed)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    T = ct.T
    stepno = 0
    coll_ev = []

    use_gather = True  # the extra run of item (a) below switches the collective off

    def gather_returns():
        # world > 1: the collective is enqueued without blocking the launch stream and overlaps the next episode's
        # steps (RCCL runs it on the process group's stream); it is waited for before the next one and at the end
        if use_gather:
            gather.gather(env._final_return, async_op=world > 1)

    seg_events = []  # (start, end) HIP events around the step launches of each episode inside the timed region
    seg_on = False
    packed_seen = []
    act_log = []  # which of the 16 action tensors every step() of this env used, in order (the parity replay reads it)

    def one_step():
        nonlocal stepno
        phase = stepno % T
        if seg_on and env._host_auto and phase == 0:  # first day of an episode: nothing but step kernels until day T-2
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            seg_events.append([ev, None])
        env.step(pool[stepno & 15])
        act_log.append(stepno & 15)
        if seg_on and phase == 1:
            packed_seen.append(env.packed_state)  # host-side bookkeeping only: no device work, no sync
        if seg_on and env._host_auto and phase == T - 2 and seg_events and seg_events[-1][1] is None:
            ev = torch.cuda.Event(enable_timing=True)  # after day T-2: the terminal step's call also launches the reset
            ev.record()
            seg_events[-1][1] = ev
        stepno += 1
        if stepno % T == 0:  # lock step: every env just finished an episode
            gather_returns()

    # Host housekeeping BEFORE the warm-up steps, so that nothing slow stands between them and the timed region: a
    # collector pass over this process's heap takes tens of milliseconds, and a GPU left idle that long starts its next
    # launches slowly -- 20 steps behind a gc.collect() cost 2.7 us per step more than 20 steps behind other steps
    # (profiles/r06/exp_sync_latency.log: 37.5 against 34.8 us, wall; timing events inside the window cost nothing).
    # Until round 6 the collect sat between the warm-up and the region: 3 of the ~4.5 us per step by which the driver's
    # 20-step window read slower than a long run. The collector stays off from here to the end of the timed region.
    import gc

    gc.collect()
    gc.disable()
    # the first HIP event a process records creates the runtime's event pool (measured: 0.15 ms for that one call,
    # 7 us afterwards, profiles/r03/exp_window_latency.log); the timing events below must not pay for it inside the region
    for _ in range(2):
        _w0, _w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _w0.record(); _w1.record(); _w1.synchronize(); _w0.elapsed_time(_w1)
    for _ in range(args.warmup):
        one_step()
    gather.wait()
    torch.cuda.synchronize()
    if world > 1:  # device time of one blocking collective (reported; the timed loop overlaps it)
        for _ in range(3):
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            gather.gather(env._final_return)
            c1.record()
            coll_ev.append((c0, c1))
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
        act_log.append(0)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            for i in range(args.graph):
                env.step(pool[i & 15])
        torch.cuda.synchronize()
    first_ev = []

    def timed_region(segments: bool):
        """EXACTLY args.steps steps between barrier + device synchronisation on both sides; (wall seconds on this rank,
        device milliseconds). segments: also record the per-episode HIP events of the headline's kernel timing."""
        nonlocal seg_on, stepno
        gc.disable()  # no collector pause inside the timed region (re-enabled right after it; the collect ran before the warm-up)
        wdist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_start = time.perf_counter()
        e0.record()
        if graph is None:
            seg_on = segments
            for i in range(args.steps):
                one_step()
                if i == 0 and segments:  # behind the region's FIRST step launch: see `first_to_last` below
                    first_ev.append(torch.cuda.Event(enable_timing=True))
                    first_ev[-1].record()
            seg_on = False
        else:
            for _ in range(args.steps // args.graph):
                graph.replay()
                act_log.extend(i & 15 for i in range(args.graph))
                before = stepno
                stepno += args.graph
                if stepno // T != before // T:
                    gather_returns()
        gather.wait()
        e1.record()
        torch.cuda.synchronize()
        wdist.barrier()
        w = time.perf_counter() - t_start
        gc.enable()
        # device time from the completion of the region's first step launch to the completion of its last one: steps - 1
        # back-to-back launches WITHOUT what precedes the first one. A region that starts on an idle GPU (it must: barrier +
        # device synchronisation on both sides) pays ~50 us between the e0 marker and the first kernel's start -- the host's
        # return to Python, the ctypes call, the doorbell, the wake-up -- which is launch latency, not launch DURATION; in a
        # 20-step window it reads as 3 us per step (profiles/r06/window_gaps_driver_args.txt: the kernel trace of the same
        # run shows 34.6 us per launch and no gaps where e0 -> e1 says 37.4)
        first_to_last = first_ev[-1].elapsed_time(e1) if (segments and first_ev) else None
        return w, e0.elapsed_time(e1), first_to_last

    # ---- multi-GPU: the same workload on ONE GPU of this very job, before the headline (the other ranks wait idle at a
    # barrier, so rank 0 has its GPU, its PCIe link and the host to itself): the denominator of `weak_efficiency`
    single = None
    if world > 1 and graph is None:
        wdist.barrier()
        if rank == 0:
            use_gather = False
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one_step()
            torch.cuda.synchronize()
            single = float(n) * args.steps / (time.perf_counter() - t1)
            use_gather = True
        wdist.barrier()
        # every rank back on the same day of an episode (rank 0 ran ahead): a whole-batch reset and the W warm-up steps
        # again, untimed
        env.reset(seed=args.seed + 1)
        stepno = 0
        act_log.clear()
        for _ in range(args.warmup):
            one_step()
        gather.wait()
        torch.cuda.synchronize()

    wall, dev_ms, first_to_last_ms = timed_region(segments=True)
    wall = wdist.max_over_ranks(wall, device)
    stepno_head = stepno  # where the headline region ended (the extra region below moves on)
    # the form of the per-env state the step kernel streamed in the measured launches (read NOW: later read-backs of
    # the state bring the canonical words up to date)
    packed = all(packed_seen) if packed_seen else env.packed_state
    # ---- multi-GPU: the same K steps once more WITHOUT the return all-gather: what the collective costs end to end
    no_coll = None
    if world > 1:
        use_gather = False
        w2, _, _ = timed_region(segments=False)
        no_coll = wdist.max_over_ranks(w2, device)
        use_gather = True
    status = env.check_status()
    mean_ret = float(gather.mean(env._final_return).item())
    collective_ms = min(a.elapsed_time(b) for a, b in coll_ev) if coll_ev else None

    # step-kernel launch time, live: HIP events on the launch stream around back-to-back launches inside one
    # episode (no reset kernel, no collective in between); rocprofv3 --kernel-trace of this command must agree
    kernel_us = None
    kernel_timing = None
    segs = [(a, b) for a, b in seg_events if b is not None]
    if segs:
        # the step launches of the TIMED REGION itself: per episode the T-1 back-to-back launches from its first day to
        # the day before the terminal one (whose call also launches the reset kernel)
        kernel_us = sum(a.elapsed_time(b) for a, b in segs) * 1e3 / (len(segs) * (T - 1))
        kernel_timing = (f"HIP events on the launch stream around the {T - 1} back-to-back step launches of each of the "
                         f"{len(segs)} whole episodes inside the timed region (reset kernels and collectives excluded)")
    elif graph is None and args.steps < T - 1 and (stepno_head - args.steps) // T == (stepno_head - 1) // T and (stepno_head % T) != 0:
        # a timed region shorter than an episode that did not cross an episode boundary (the driver's --steps 20): its
        # launches are nothing but step kernels, so the HIP events around the region itself are the measurement
        if first_to_last_ms is not None and args.steps > 1:
            kernel_us = first_to_last_ms * 1e3 / (args.steps - 1)
            kernel_timing = (f"HIP events on the launch stream inside the timed region itself (no reset kernel, no collective "
                             f"in it): from behind its first step launch to behind its last = {args.steps - 1} back-to-back "
                             f"launches; the whole region e0 -> e1 incl. the idle-GPU start latency reads "
                             f"{dev_ms * 1e3 / args.steps:.2f} us per step (`region_us_per_step`)")
        else:
            kernel_us = dev_ms * 1e3 / args.steps
            kernel_timing = (f"HIP events on the launch stream around the {args.steps} back-to-back step launches of the timed "
                             "region itself (no reset kernel, no collective inside)")
    elif graph is None:
        k_steps = min(T - 2, 140)
        if env._host_auto and T - env._steps_in_episode <= k_steps:
            for _ in range(T - env._steps_in_episode):
                env.step(pool[0])  # finish this episode: the measurement must not contain a reset kernel
                act_log.append(0)
        kms, _ = timed_steps(env, pool, k_steps, torch)
        act_log.extend(i & 15 for i in range(k_steps))
        kernel_us = kms * 1e3 / k_steps
    # This box's memory side, in this process, on this run's tables and CURRENT episode tuples: AFTER every timing of the
    # step kernel (the probe is ~50 ms of full-bandwidth work, and the chip throttles for some milliseconds after such a
    # burst: in front of the driver's 20-step window it read 42.7 us per launch against 35 us without,
    # profiles/r04/bench_driver_args_*.log -- and in front of the fallback timing above it would do the same)
    calib = None if args.no_calibration else box_calibration(env, dt, ct, packed, torch)
    # parity of the very batch that was timed, against the oracle (rank 0; cpu-baseline leg, outside every timed region)
    parity = None
    if rank == 0 and not args.no_parity:
        if args.tamper == "return":
            env._final_return[n - 1] += 0.01
        elif args.tamper == "obs":
            env._obs[0, 3] += 1.0
        try:
            parity = parity_check(env, sd, ct, pool, act_log, torch)
        except Exception as e:  # noqa: BLE001  (reported, never loses the headline line)
            parity = {"error": repr(e), "ok": False}
    # ---- N > 1: BASELINE configs[4] in the same job (every rank takes part; rank 0 reports)
    c4_sharded = None
    if world > 1 and args.workload == "configs2" and graph is None and not args.no_extras:
        env.close()
        try:
            c4_sharded = sharded_workload("configs4", args, rank, world, device, torch, HeatAlertVecEnv, synth, tables, wdist)
        except Exception as e:  # noqa: BLE001  (reported, never loses the headline line)
            c4_sharded = {"error": repr(e)}
    rc = 0
    if rank == 0:
        total_env_steps = float(n) * world * args.steps
        per_launch_s = (kernel_us * 1e-6) if kernel_us else dev_ms * 1e-3 / args.steps
        cb = compulsory_bytes(ct.n_obs, not args.no_obs, packed)
        achieved = cb["total"] * n / per_launch_s / 1e9
        variant = env.step_kernel_name
        kname = f"{variant}<obs={not args.no_obs}{', packed lock-step state' if packed else ''}>" + (
            " (in-kernel autoreset)" if env._dev_auto else " + k_reset once per episode")
        # fabric traffic from the PMC passes, only if collected on these very kernel sources
        src_sha = wbuild.source_sha()
        traffic = traffic_note = None
        tp = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tp):
            try:
                ent = json.load(open(tp)).get(args.workload + ("_noobs" if args.no_obs else ""))
            except Exception:  # noqa: BLE001
                ent = None
            if isinstance(ent, dict) and ent.get("src_sha") == src_sha and ent.get("step_kernel") == variant:
                traffic = ent
            elif ent is not None:
                traffic_note = "profiles/traffic_latest.json was collected on other kernel sources: not reported"
        traffic_live = False
        if world == 1 and not args.no_live_traffic and not args.graph:
            # measured in this very run (every timing above is over; the env is closed first so that the probe has the GPU's
            # memory to itself): the replayed figure above is only the fallback
            env.close()
            live, why = live_traffic(args.workload, n, args.no_obs, args.step_kernel, args.episode_order, variant)
            if live is not None:
                traffic, traffic_live, traffic_note = dict(live, src_sha=src_sha), True, None
            else:
                traffic_note = f"live counter passes not available ({why})" + ("; replayed from profiles/traffic_latest.json"
                                                                                 if traffic is not None else "")
        # the memory side's measured rate for this access pattern without any env logic (static reference from
        # profiles/, 1 048 576 envs): kernel time / probe time says how close the kernel is to what the chip delivers
        probe = None
        try:
            pc = json.load(open(os.path.join(ROOT, "profiles", "probe_ceiling.json")))["packed" if packed else "unpacked"]
            pc = pc.get(args.workload)
            if pc and n == 1048576 and not args.no_obs:
                probe = {"probe_us": pc["us"], "kernel_over_probe": per_launch_s * 1e6 / pc["us"],
                         "source": "profiles/r03/fabric_probe.log (tools/fabric_probe.hip, random data, "
                                   + ("packed" if packed else "unpacked") + " state streams)"}
        except Exception:  # noqa: BLE001
            probe = None
        out = {
            "metric": "env_steps_per_sec", "value": total_env_steps / wall, "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32 tables, fp64 logit accumulation, f32 sigmoid/reward", "data": "synthetic",
            "config": {"workload": desc, "num_envs_per_gpu": n, "num_envs_total": n * world,
                       "episode_days": T, "n_samples": ct.n_samples, "obs": not args.no_obs,
                       "policy": "Bernoulli(0.1) actions from a device RNG, table budgets",
                       "seed_mode": "device", "autoreset": "same_step", "reward_path": "gather",
                       "step_kernel": variant, "packed_lockstep_state": packed, "episode_order": args.episode_order,
                       "hipgraph_steps": args.graph,
                       "collective": "all_gather_into_tensor(f32[num_envs_per_gpu]) per episode, overlapped with the next "
                                     "episode's steps" if world > 1 else "none",
                       "single_gpu_reference": f"python bench.py --gpus 1 --workload {args.workload}"
                                               + (f" --num-envs {n}" if args.num_envs else "") +
                                               " (the N = 1 default is configs2, a different table shape: compare "
                                               "multi-GPU values with THIS workload on one GPU)" if world > 1 else None},
            "per_gpu_value": total_env_steps / wall / world,
            # multi-GPU runs judge themselves (None on one GPU): the same workload on rank 0's GPU alone while the other
            # ranks idle at a barrier, and the same K steps once more without the return all-gather
            "single_gpu_value": single,
            "efficiency_vs_rank0_alone": (total_env_steps / wall / world / single) if single else None,
            "weak_efficiency": (total_env_steps / wall / world / single) if (single and args.scaling == "weak") else None,
            "no_collective_value": (total_env_steps / no_coll) if no_coll else None,
            "collective_overhead_frac": ((wall - no_coll) / wall) if no_coll else None,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic is None else traffic["bytes_per_launch"],
                         "kernel": kname, "avg_launch_us": per_launch_s * 1e6,
                         "region_us_per_step": dev_ms * 1e3 / args.steps,  # e0 -> e1 over the whole timed region / K
                         "bytes_model": "compulsory", "compulsory_bytes_per_env_step": cb,
                         "units_per_launch": n,
                         "traffic_unit": "fabric bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE; counts "
                                         "Infinity-Cache hits, so it is an upper bound on HBM bytes)",
                         "traffic_ratio": None if traffic is None else traffic["bytes_per_launch"] / (cb["total"] * n),
                         # the rate at which the measured fabric bytes moved during this run's launches: what saturates
                         # (a 1 GiB fill writes at ~6.9 TB/s on this chip, a float4 copy moves 6.29 TB/s)
                         "fabric_gbs": None if traffic is None else traffic["bytes_per_launch"] / per_launch_s / 1e9,
                         "traffic_source": None if traffic is None else
                         {k: traffic.get(k) for k in ("read_bytes_per_launch", "write_bytes_per_launch", "read_correction",
                                                      "launches_averaged", "pass_seconds",
                                                      "kernel_avg_us", "src_sha", "commit", "profile")},
                         "traffic_note": traffic_note, "kernel_src_sha": src_sha,
                         "probe_ceiling": probe,
                         # measured in THIS process on THIS box right after the timed region (box_calibration)
                         "copy_gbs_this_box": None if not calib else calib.get("copy_gbs"),
                         "copy_gbs_variants_this_box": None if not calib else calib.get("copy_gbs_variants"),
                         "probe_us_this_box": None if not calib else calib.get("probe_us"),
                         "kernel_over_probe": (per_launch_s * 1e6 / calib["probe_us"]["streams_and_gathers"])
                         if calib and calib.get("probe_us") else None,
                         "frac_of_copy_rate_this_box": (achieved / calib["copy_gbs"]) if calib and calib.get("copy_gbs") else None,
                         "calibration": None if not calib else {k: v for k, v in calib.items() if k in ("source", "error")},
                         "traffic_live": traffic_live,
                         "traffic_provenance": None if traffic is None else
                         ("LIVE: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_probe.py: the same "
                          "workload on the same library + a 1 GiB calibration copy) started by this run after its timed "
                          "region" if traffic_live else
                          "REPLAYED from profiles/traffic_latest.json (rocprofv3 --pmc passes of the same command on the "
                          "same kernel sources, src_sha checked)"),
                         "frac_of_measured_copy_bw": achieved / 6290.0,
                         "frac_with_unpacked_161B_model": compulsory_bytes(ct.n_obs, not args.no_obs)["total"] * n
                         / per_launch_s / 1e9 / HBM_PEAK_GBS,
                         "survey_8d_model": {"bytes_per_env_step": SURVEY_8D_BYTES["no_obs" if args.no_obs else "obs"],
                                             "gbs": SURVEY_8D_BYTES["no_obs" if args.no_obs else "obs"] * n
                                             / per_launch_s / 1e9,
                                             "note": "SURVEY 8d charges a 100-B feature row and two 112-B coefficient "
                                                     "rows per env-step to HBM; they are cache-resident, so this "
                                                     "figure is not an HBM rate and may exceed the peak"},
                         "timing": kernel_timing or "HIP events on the launch stream around back-to-back step launches "
                                                     "inside one episode, right after the timed region / launches"},
            "kernel_env_steps_per_sec_per_gpu": n / per_launch_s,
            "rccl_ranks_seen": torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1,
            "collective_ms": collective_ms,
            "status_bits": status, "mean_final_return": mean_ret, "setup_s": t_setup,
            "parity": parity,
        }
        # the driver's record keeps the scalars of `config` and `roofline` (of other objects only the key name): what says
        # that the timed batch holds what the reference would hold is mirrored there
        psum = {"parity_ok": None if parity is None else bool(parity.get("ok")),
                "parity_max_abs_reward_err": None if parity is None else parity.get("max_abs_reward_err"),
                "parity_sampled_envs": None if parity is None else parity.get("sampled"),
                "parity_env_steps_replayed_per_env": None if parity is None else parity.get("env_steps_replayed_per_env"),
                "parity_ints_and_obs_exact": None if parity is None else
                bool(parity.get("ints_exact")) and parity.get("obs_exact") is not False,
                "status_bits": status}
        out["roofline"].update(psum)
        out["config"].update(psum)
        if c4_sharded is not None:
            out["configs4_sharded"] = c4_sharded
        if world == 1 and args.episode_order == "iid" and not args.graph and not args.no_extras:
            env.close()
            extras(out, args, torch, HeatAlertVecEnv, synth, tables, dt, ct, device, n, augment, pool, cb, T, wname)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(sd, ct, args.seed)
                # all host cores this process may use, one single-threaded process per core
                cpus = usable_cpus()
                out["cpu_baseline"]["cpus"] = cpus
                try:
                    out["cpu_baseline"]["multi_core"] = cpu_baseline_multicore(wname, args.seed, cpus["usable"])
                except Exception as e:  # noqa: BLE001  (a reported extra, never fatal)
                    out["cpu_baseline"]["multi_core"] = {"error": repr(e)}
            except Exception as e:  # noqa: BLE001  (the headline JSON line must survive a failing baseline leg)
                out["cpu_baseline"] = {"error": repr(e)}
        so = out.get("sorted_episode_order")
        if isinstance(so, dict) and "value_end_to_end" in so:  # (scalars the driver's record keeps)
            out["roofline"].update(sorted_order_kernel_us=so["kernel_us"], sorted_order_value_end_to_end=so["value_end_to_end"],
                                   iid_value_end_to_end=so["iid_value_end_to_end"])
        print(json.dumps(out), flush=True)
        # the exit code says whether the numbers above count: parity of the timed batch held and no kernel flagged anything
        if parity is not None and not parity.get("ok", False) and "skipped" not in parity:
            print(f"bench.py: PARITY FAILED on the timed batch: {parity.get('notes') or parity.get('error')}", file=sys.stderr, flush=True)
            rc = 5
        elif status != 0 or (parity is not None and parity.get("status_bits")):
            print(f"bench.py: a kernel raised status bits ({status}, parity leg {parity and parity.get('status_bits')})",
                  file=sys.stderr, flush=True)
            rc = 6
    env.close()
    rc = int(wdist.max_over_ranks(float(rc), device))  # every rank leaves with rank 0's verdict (the others hold 0)
    wdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
