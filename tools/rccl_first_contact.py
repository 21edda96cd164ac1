#!/usr/bin/env python3
"""First contact with RCCL on more than one rank: shard invariance of the env batch through the real collective.

World size W, one process per GPU (backend nccl = RCCL over xGMI), each rank steps `--num-envs` envs whose global ids are
rank * n ... rank * n + n - 1 (tables replicated, device RNG keyed by global id: SURVEY 8e) for
`--episodes` whole episodes; after each episode the finished-episode returns f32 [n] of every rank are all-gathered with
dist.ReturnGatherer(async_op=True) -- one all_gather_into_tensor per episode, overlapped with the next episode's steps, the
only communication of the path. Rank 0 writes the gathered returns of every episode to --out (torch.save). A
single-process run of the same W * n global ids must give the same file bit for bit: that is what
tests/test_rccl_gpu.py asserts, and the whole of what "sharding" means for this path (env.py:133-262 has no cross-env
access).

    # self-launch (this process never touches a GPU; it starts one fresh child per rank and waits for them):
    python tools/rccl_first_contact.py --launch 2 --out /tmp/two.pt
    # the same under torchrun:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \\
        tools/rccl_first_contact.py --out /tmp/two.pt
    # the single-process reference of the same global ids:
    python tools/rccl_first_contact.py --single 2 --out /tmp/one.pt

--backend gloo rehearses the same code where there is one GPU (the ranks share it; gloo stages the collective through the
host) and --stub where there is none (returns are a pure function of (global id, episode); no env, no GPU): the launcher,
the rendezvous, the gatherer and the file are then covered by tests/test_dist_cpu.py on CPU.
"""
from __future__ import annotations

import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--launch", type=int, default=0, help="start this many ranks as fresh child processes and wait for them")
    p.add_argument("--single", type=int, default=0, metavar="W",
                   help="no process group: ONE process steps all W * num_envs global ids (the reference the sharded run must equal)")
    p.add_argument("--num-envs", type=int, default=65536, help="envs per rank")
    p.add_argument("--episodes", type=int, default=3)
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    p.add_argument("--stub", action="store_true", help="no env, no GPU: returns = f(global id, episode)")
    p.add_argument("--n-days", type=int, default=24, help="episode length of the synthetic tables")
    p.add_argument("--out", required=True)
    p.add_argument("--timeout", type=float, default=600.0)
    return p.parse_args(argv)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args) -> int:
    """One fresh interpreter per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), polled: a rank that dies takes the
    others down instead of leaving them in a rendezvous."""
    port = _free_port()
    base = [sys.executable, os.path.abspath(__file__), "--num-envs", str(args.num_envs), "--episodes", str(args.episodes),
            "--backend", args.backend, "--n-days", str(args.n_days), "--out", args.out] + (["--stub"] if args.stub else [])
    procs = []
    for r in range(args.launch):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.launch), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(base, env=env, stdout=subprocess.PIPE if r else None, stderr=subprocess.PIPE, text=True))
    deadline = time.time() + args.timeout
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad or time.time() > deadline:
            rc = bad[0].returncode if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.1)
    for r, p in enumerate(procs):
        try:
            _, err = p.communicate(timeout=20)
        except subprocess.TimeoutExpired:
            p.kill()
            _, err = p.communicate()
        if p.returncode != 0:
            rc = rc or p.returncode
            print(f"rank {r} exited {p.returncode}: {(err or '')[-1500:]}", file=sys.stderr, flush=True)
    return rc


def stub_returns(gid0: int, n: int, episode: int):
    import torch

    g = torch.arange(gid0, gid0 + n, dtype=torch.float64)
    return (-(g * 0.25 + 1.0) - 1000.0 * episode).float()


def run(args) -> int:
    import torch

    from weather2alert_amd import dist as wdist

    single = args.single > 0
    rank = 0 if single else int(os.environ.get("RANK", "0"))
    world = 1 if single else int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if single else int(os.environ.get("LOCAL_RANK", "0"))
    n = args.num_envs * (args.single if single else 1)
    if args.stub:
        device = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "needs a ROCm GPU (or --stub)"
        device = torch.device(f"cuda:{local if args.backend == 'nccl' else local % torch.cuda.device_count()}")
        torch.cuda.set_device(device)
    if not single:
        wdist.init_from_env(args.backend, None if args.stub else device)
        seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        if seen != world:
            print(f"process group has {seen} ranks, expected {world}", file=sys.stderr)
            return 4
    gather = wdist.ReturnGatherer(n, device)
    per_episode = []
    status = 0
    kernel = None
    if args.stub:
        for ep in range(args.episodes):
            gather.gather(stub_returns(rank * n, n, ep), async_op=world > 1)
            per_episode.append(gather.wait().clone())
    else:
        from weather2alert_amd import HeatAlertVecEnv, synth, tables

        sd = synth.make_synth("linear", n_fips=40, years=[2006, 2007, 2008], n_samples=10, n_days=args.n_days, seed=0,
                              extra_confounder_fips=4)
        ct = tables.compile_from_synth(sd)
        # the 64-envs-per-wave kernel on every shard size, so that the sharded and the single-process run execute the
        # same arithmetic in the same order (the 4-lanes-per-env kernel w2a_step would pick below 131 072 envs differs
        # in the order of its fp64 additions)
        env = HeatAlertVecEnv(n, tables=ct, device=device, similar_climate_counties=True, env_gid0=rank * n, step_kernel="wide")
        g = torch.Generator(device="cpu").manual_seed(1234)
        # actions are a function of the GLOBAL env id: every rank draws the whole job's pool and keeps its slice
        total = n * world
        pool = [(torch.rand(total, generator=g) < 0.2).to(torch.int32)[rank * n:(rank + 1) * n].to(device) for _ in range(8)]
        env.reset(seed=7)
        T = ct.T
        pending = False
        for ep in range(args.episodes):
            for t in range(T):
                env.step(pool[(ep * T + t) & 7])
                if pending:  # the previous episode's collective was in flight while this step was enqueued: collect it now
                    per_episode.append(gather.wait().clone())
                    pending = False
            # the terminal step has run (and, in lock step, the reset after it): the finished episode's returns, snapshotted
            # by the gatherer and all-gathered without blocking the launch stream
            gather.gather(env._final_return, async_op=world > 1)
            pending = True
        per_episode.append(gather.wait().clone())
        kernel = env.last_step_kernel
        status = env.check_status()
        torch.cuda.synchronize()
        env.close()
    ranks_seen = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    if rank == 0:
        torch.save({"returns": [x.cpu() for x in per_episode], "world": world, "ranks_seen": ranks_seen, "num_envs_total": n * world,
                    "backend": "none" if single else args.backend, "status_bits": status, "step_kernel": kernel}, args.out)
        print(f"rccl_first_contact: {len(per_episode)} gathers of {n * world} returns over {ranks_seen} rank(s) "
              f"[{'single process' if single else args.backend}] -> {args.out}", flush=True)
    wdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return 0 if status == 0 else 6


def main():
    args = parse()
    if args.launch:
        return launch(args)  # never imports torch, never touches a GPU
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
