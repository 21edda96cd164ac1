"""Multi-GPU sharding of the env batch: one process per GPU, no per-step communication.

Envs are independent (nothing in the reference's env.py:133-262 reads another env), so the
batch is split into contiguous global-id ranges, tables are replicated, and the only
exchange is the episodic-return all-gather once per episode. ``backend="nccl"``
is RCCL over xGMI on ROCm; ``gloo`` serves the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """[start, stop) of the contiguous global env ids owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_from_env(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's env; initialises the default group if world > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


class ReturnGatherer:
    """All-gather of the per-env episodic returns f32[n_local] -> f32[world * n_local]
    (equal shard sizes) with one ``all_gather_into_tensor``; on one rank it is a no-op view."""

    def __init__(self, n_local: int, device, world: int | None = None):
        self.world = (dist.get_world_size() if dist.is_initialized() else 1) if world is None else world
        self.n_local = int(n_local)
        self.out = torch.empty(self.world * self.n_local, dtype=torch.float32, device=device)

    def gather(self, local_returns: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            self.out.copy_(local_returns)
        elif dist.get_backend() == "gloo" and local_returns.is_cuda:
            # rehearsal path (several ranks on one GPU / no RCCL): gloo has no CUDA all-gather, stage on the host
            tmp = torch.empty(self.out.shape, dtype=self.out.dtype)
            dist.all_gather_into_tensor(tmp, local_returns.cpu().contiguous())
            self.out.copy_(tmp)
        else:
            dist.all_gather_into_tensor(self.out, local_returns.contiguous())
        return self.out

    def mean(self, local_returns: torch.Tensor) -> torch.Tensor:
        """Scalar mean return over all shards (cheaper than the gather when only this is needed)."""
        s = local_returns.double().sum().reshape(1)
        if self.world > 1:
            if dist.get_backend() == "gloo" and s.is_cuda:
                c = s.cpu()
                dist.all_reduce(c)
                s = c.to(s.device)
            else:
                dist.all_reduce(s)
        return s / (self.world * self.n_local)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if not dist.is_initialized():
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
