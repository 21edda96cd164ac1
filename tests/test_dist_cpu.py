"""Multi-process path on CPU: world_size-2 gloo run of the sharding helpers and the episodic
return all-gather that bench.py / users run over RCCL (one process per GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from weather2alert_amd import dist as wdist


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8, 1000, 8388608):
        for world in (1, 2, 3, 8):
            spans = [wdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_local, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = wdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    dev = torch.device("cpu")
    start, stop = wdist.shard_range(n_local * world, rank, world)
    assert (start, stop) == (rank * n_local, (rank + 1) * n_local)
    # each rank's "finished episode returns": a function of the GLOBAL env id
    gid = torch.arange(start, stop, dtype=torch.float32)
    local = -100.0 - gid
    g = wdist.ReturnGatherer(n_local, dev)
    allr = g.gather(local)
    expect = -100.0 - torch.arange(n_local * world, dtype=torch.float32)
    ok = torch.equal(allr, expect)
    # overlapped form: the returns are snapshotted, so the env may overwrite its buffer while the collective runs
    mine = local.clone()
    out = g.gather(mine, async_op=True)
    mine.fill_(123.0)
    ok = ok and torch.equal(g.wait(), expect) and out is g.out and g._work is None
    g.gather(local, async_op=True)
    ok = ok and torch.equal(g.gather(local * 2), expect * 2)  # a new gather first completes the pending one
    m = float(g.mean(local))
    mx = wdist.max_over_ranks(float(rank + 1), dev)
    wdist.barrier()
    q.put((rank, ok, m, mx))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo_return_gather():
    world, n_local = 2, 1000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_local, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    expect_mean = float((-100.0 - torch.arange(n_local * world, dtype=torch.float64)).mean())
    for rank, ok, m, mx in res:
        assert ok
        assert abs(m - expect_mean) < 1e-9
        assert mx == float(world)


def test_single_process_gatherer_is_a_copy():
    g = wdist.ReturnGatherer(5, torch.device("cpu"), world=1)
    x = torch.arange(5, dtype=torch.float32)
    assert torch.equal(g.gather(x), x)
    assert float(g.mean(x)) == 2.0


@pytest.mark.timeout(180)
def test_bench_launches_itself_for_several_ranks():
    """`python bench.py --gpus 2` without a launcher starts one child per rank (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set) before anything touches a GPU and forwards rank 0's JSON line; rehearsed on gloo with the stub
    workload (no env stepping, no GPU): process group of 2, per-episode return all-gather, max-over-ranks timing."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--workload", "launcher_stub", "--steps", "306", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=170)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks_seen"] == 2 and out["gather_ok"] and out["stub"]
    assert out["steps"] == 306 and out["collective_ms"] is not None and out["scaling"] == "weak"
    # a multi-rank line judges itself: the same workload on rank 0 alone, the same steps without the collective
    for k in ("single_gpu_value", "weak_efficiency", "efficiency_vs_rank0_alone", "no_collective_value", "collective_overhead_frac"):
        assert out[k] is not None, k
    assert out["single_gpu_value"] > 0 and out["weak_efficiency"] > 0 and out["collective_overhead_frac"] < 1
    # a mismatching launcher environment is refused, not silently benchmarked on one rank
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "launcher_stub",
                          "--backend", "gloo"], capture_output=True, text=True, env=dict(env, WORLD_SIZE="1", RANK="0"),
                         timeout=60)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def test_bench_launcher_eight_ranks_and_a_rank_that_dies_at_startup():
    """`--gpus 8` rehearsal on gloo (the stub workload: everything around the env -- process group, barrier, the
    per-episode return all-gather, max-over-ranks timing, one JSON line) and failure propagation: a rank that exits at
    start-up must end the whole launch within seconds with a non-zero code, not leave the others in the rendezvous."""
    import json
    import subprocess
    import sys
    import time

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--backend", "gloo", "--workload",
            "launcher_stub", "--steps", "306", "--warmup", "0", "--num-envs", "1024"]
    r = subprocess.run(base, capture_output=True, text=True, env=env, timeout=280)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["rccl_ranks_seen"] == 8 and out["gather_ok"]
    assert out["config"]["num_envs_per_gpu"] == 1024 and out["collective_ms"] is not None
    # strong scaling on the same 8 ranks: a fixed total (8192) split over the ranks, the all-gather reassembles it
    st = subprocess.run(base[:-2] + ["--num-envs", "8192", "--scaling", "strong"], capture_output=True, text=True, env=env,
                        timeout=280)
    assert st.returncode == 0, (st.stdout[-2000:], st.stderr[-2000:])
    so = json.loads([ln for ln in st.stdout.splitlines() if ln.startswith("{")][-1])
    assert so["scaling"] == "strong" and so["config"]["num_envs_per_gpu"] == 1024 and so["config"]["num_envs_total"] == 8192
    assert so["gather_ok"] and so["rccl_ranks_seen"] == 8 and so["weak_efficiency"] is None
    assert so["efficiency_vs_rank0_alone"] is not None
    odd = subprocess.run(base[:-2] + ["--num-envs", "8190", "--scaling", "strong"], capture_output=True, text=True, env=env,
                         timeout=120)
    assert odd.returncode != 0 and "divisible" in (odd.stderr + odd.stdout)
    t0 = time.time()
    bad = subprocess.run(base + ["--fail-rank", "5"], capture_output=True, text=True, env=env, timeout=120)
    assert bad.returncode == 3 and time.time() - t0 < 60, (bad.returncode, time.time() - t0, bad.stderr[-1500:])
    assert "rank 5 exited with code 3" in bad.stderr and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
    # a launch that outlives --launch-timeout is ended too (rank 3 sleeps in place of working)
    slow = subprocess.run(base + ["--fail-rank", "-2", "--launch-timeout", "4"], capture_output=True, text=True,
                          env=dict(env, W2A_BENCH_TEST_HANG_RANK="3"), timeout=120)
    assert slow.returncode == 124 and "launch-timeout" in slow.stderr


@pytest.mark.timeout(240)
def test_bench_sweep_runs_every_gpu_count_as_its_own_job():
    """`bench.py --sweep 1,2,4 [--scaling ...]`: the parent never touches a GPU and runs one `--gpus N` job per entry as a
    fresh child group, one JSON line per N + a summary line; a failing N is reported in its line and in the exit code
    without losing the others. Rehearsed on gloo with the stub workload (the one command for the day
    an 8-GPU node exists)."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    base = [sys.executable, os.path.join(root, "bench.py"), "--backend", "gloo", "--workload", "launcher_stub", "--steps", "306",
            "--warmup", "0", "--num-envs", "1024"]
    r = subprocess.run(base + ["--sweep", "1,2,4"], capture_output=True, text=True, env=env, timeout=200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert [ln.get("sweep_n") for ln in lines[:-1]] == [1, 2, 4] and lines[-1]["sweep"] == [1, 2, 4]
    for ln in lines[:-1]:
        assert ln["n_gpus"] == ln["rccl_ranks_seen"] == ln["sweep_n"] and ln["num_envs_per_gpu"] == 1024 and ln["value"] > 0
        if ln["sweep_n"] > 1:  # a multi-rank job judges itself; the sweep adds the cross-job figure
            assert ln["single_gpu_value"] > 0 and ln["weak_efficiency"] > 0 and ln["collective_overhead_frac"] is not None
    assert set(lines[-1]["efficiency_vs_smallest_n"]) == {"1", "2", "4"} and lines[-1]["efficiency_vs_smallest_n"]["1"] == 1.0
    assert lines[-1]["failed"] == [] and lines[-1]["baseline_n"] == 1
    # strong scaling: a fixed total split over the ranks; a total that does not divide is that job's error, not the sweep's end
    st = subprocess.run(base[:-2] + ["--num-envs", "1026", "--scaling", "strong", "--sweep", "2,4"], capture_output=True, text=True,
                        env=env, timeout=200)
    sl = [json.loads(ln) for ln in st.stdout.splitlines() if ln.startswith("{")]
    assert st.returncode != 0 and sl[0]["num_envs_per_gpu"] == 513 and sl[0]["scaling"] == "strong"
    assert "error" in sl[1] and "divisible" in sl[1]["stderr_tail"] and sl[-1]["failed"] == [4]
    bad = subprocess.run(base + ["--sweep", "1,x"], capture_output=True, text=True, env=env, timeout=60)
    assert bad.returncode != 0 and "--sweep" in bad.stderr


def test_bench_defaults_follow_baseline_configs():
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # one workload for every N: the driver computes scaling efficiency from the per-N values of one command; BASELINE
    # configs[4] rides along in the N > 1 jobs as `configs4_sharded`
    assert b.parse([]).workload == "configs2" and b.parse(["--gpus", "8"]).workload == "configs2"
    assert b.parse(["--sweep", "1,2,4,8"]).workload == "configs2" and b.parse(["--sweep", "1,8"]).sweep == [1, 8]
    assert b.WORKLOADS["configs4"][:3] == ("nn_full_medicare_all", 1048576, False)
    cb = b.compulsory_bytes(29, True)
    assert cb["total"] == 161 and b.compulsory_bytes(29, False)["total"] == 45
    assert b.compulsory_bytes(29, True, packed=True)["total"] == 149  # 8 + 8 + 4 in, 8 + 4 + 1 + 116 out
    cpus = b.usable_cpus()  # the CPU baseline runs one worker per CPU the process may really use
    assert 1 <= cpus["usable"] <= cpus["os_cpu_count"] and cpus["usable"] <= cpus["affinity"]
    assert cpus["cgroup_quota_cpus"] is None or cpus["usable"] <= max(1, int(cpus["cgroup_quota_cpus"] + 0.5))
    assert isinstance(b.cpu_model(), str) and b.cpu_model()


@pytest.mark.timeout(300)
def test_rccl_first_contact_children_on_cpu(tmp_path):
    """tools/rccl_first_contact.py -- the children of the multi-GPU RCCL test (tests/test_rccl_gpu.py) -- with --stub over gloo:
    its launcher (fresh interpreters, polled), the rendezvous on 127.0.0.1, the overlapped return gather per episode and the
    result file, on 2 and on 4 ranks, against its own single-process mode; and a launch whose group cannot form fails instead
    of hanging."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "tools", "rccl_first_contact.py")
    for world in (2, 4):
        two, one = str(tmp_path / f"w{world}.pt"), str(tmp_path / f"s{world}.pt")
        r = subprocess.run([sys.executable, tool, "--launch", str(world), "--backend", "gloo", "--stub", "--num-envs", "3000", "--out", two],
                           capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run([sys.executable, tool, "--single", str(world), "--stub", "--num-envs", "3000", "--out", one],
                           capture_output=True, text=True, timeout=200)
        assert r.returncode == 0, r.stderr[-2000:]
        a, b = torch.load(two), torch.load(one)
        assert a["ranks_seen"] == world and b["ranks_seen"] == 1 and a["num_envs_total"] == b["num_envs_total"] == 3000 * world
        assert len(a["returns"]) == 3 and all(torch.equal(x, y) for x, y in zip(a["returns"], b["returns"]))
    # without a GPU the real children must fail loudly (no CPU fallback), and the launcher must report it and return
    r = subprocess.run([sys.executable, tool, "--launch", "2", "--backend", "gloo", "--num-envs", "64", "--out", str(tmp_path / "x.pt"),
                        "--timeout", "120"], capture_output=True, text=True, timeout=200)
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a ROCm GPU" in r.stderr

