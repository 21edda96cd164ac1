"""First contact with RCCL on more than one rank: tests that run BY THEMSELVES the day the GPU box has two
GPUs, and skip cleanly on the one-GPU pool. The pytest process -- which has initialised its GPU -- only SPAWNS fresh children
(subprocess: one launcher that never touches a GPU, which starts one interpreter per rank); nothing is re-exec'd.

  * tools/rccl_first_contact.py --launch 2          2 ranks x 65 536 envs, env_gid0 = rank * n, three episodes with
    dist.ReturnGatherer(async_op=True) over backend nccl = RCCL; rank 0 saves the gathered returns of every episode
  * tools/rccl_first_contact.py --single 2          ONE process, the same 131 072 global ids
  -> the two files must be equal bit for bit (shard invariance through the real collective) and the group must have had 2 ranks
  * bench.py --gpus 2 --steps 20 --warmup 5         a line with n_gpus 2, rccl_ranks_seen 2 and weak_efficiency

The same children run on ONE GPU with --backend gloo (both ranks share the card, the collective is staged through the host):
that rehearsal is part of the suite on every box; without any GPU (--stub) it is tests/test_dist_cpu.py. The reference has
no counterpart: env.py:133-262 has no cross-env access, which is why the split is this simple.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "rccl_first_contact.py")


def _two_ranks_equal_one_process(tmp_path, backend, n):
    two, one = str(tmp_path / "two.pt"), str(tmp_path / "one.pt")
    r = subprocess.run([sys.executable, TOOL, "--launch", "2", "--backend", backend, "--num-envs", str(n), "--out", two],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    r = subprocess.run([sys.executable, TOOL, "--single", "2", "--num-envs", str(n), "--out", one],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    a, b = torch.load(two), torch.load(one)
    assert a["ranks_seen"] == 2 and a["world"] == 2 and a["backend"] == backend and b["ranks_seen"] == 1
    assert a["num_envs_total"] == b["num_envs_total"] == 2 * n and a["status_bits"] == 0 and b["status_bits"] == 0
    assert a["step_kernel"] == b["step_kernel"] == "k_step64<packed>"
    assert len(a["returns"]) == len(b["returns"]) == 3
    for ep, (x, y) in enumerate(zip(a["returns"], b["returns"])):
        assert x.shape == (2 * n,) and torch.equal(x, y), f"episode {ep}: gathered returns of the sharded run differ from the single-process run"
        assert float(x.min()) < 0.0 and len(torch.unique(x)) > n // 4  # real returns, not a buffer of zeros
    assert not torch.equal(a["returns"][0], a["returns"][1])  # consecutive episodes differ
    return a


@pytest.mark.timeout(1200)
def test_two_gloo_ranks_sharing_one_gpu_equal_one_process(tmp_path):
    """Every box: the children of the RCCL test below with backend gloo, both ranks on cuda:0."""
    _two_ranks_equal_one_process(tmp_path, "gloo", 65536)


@pytest.mark.timeout(1200)
def test_two_rccl_ranks_equal_one_process(tmp_path):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL with more than one rank); the one-GPU pool runs the gloo rehearsal above")
    _two_ranks_equal_one_process(tmp_path, "nccl", 65536)


@pytest.mark.timeout(1200)
def test_bench_on_two_gpus_reports_weak_efficiency():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-extras",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=1100, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["config"]["num_envs_total"] == 2 * 1048576
    assert d["single_gpu_value"] > 1e10 and 0.5 < d["weak_efficiency"] < 1.2, d["weak_efficiency"]
    assert d["roofline"]["parity_ok"] is True and d["status_bits"] == 0
