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
    (equal shard sizes) with one ``all_gather_into_tensor``; on one rank it is a no-op view.

    ``gather(x, async_op=True)`` overlaps the collective with the next episode's steps: the returns are
    snapshotted into a private buffer (the env overwrites its own on the next terminal step), the collective is
    enqueued without blocking the launch stream (RCCL runs it on the process group's own stream), and ``wait()``
    makes the current stream wait for it before ``out`` is consumed. One collective may be in flight; starting
    another waits for the previous one first."""

    def __init__(self, n_local: int, device, world: int | None = None, force_collective: bool = False):
        self.world = (dist.get_world_size() if dist.is_initialized() else 1) if world is None else world
        self.n_local = int(n_local)
        self.out = torch.empty(self.world * self.n_local, dtype=torch.float32, device=device)
        # force_collective: run the collective even in a one-rank group (exercises RCCL on a single GPU)
        self._collective = self.world > 1 or (force_collective and dist.is_initialized())
        self._src = torch.empty(self.n_local, dtype=torch.float32, device=device) if self._collective else None
        self._work = None
        self._staged = None

    def gather(self, local_returns: torch.Tensor, async_op: bool = False) -> torch.Tensor:
        self.wait()
        if not self._collective:
            self.out.copy_(local_returns)
            return self.out
        if dist.get_backend() == "gloo" and local_returns.is_cuda:
            # rehearsal path (several ranks on one GPU / no RCCL): gloo has no CUDA all-gather, stage on the host
            tmp = torch.empty(self.out.shape, dtype=self.out.dtype)
            self._work = dist.all_gather_into_tensor(tmp, local_returns.cpu().contiguous(), async_op=True)
            self._staged = tmp
        else:
            self._src.copy_(local_returns)
            self._work = dist.all_gather_into_tensor(self.out, self._src, async_op=True)
        if not async_op:
            self.wait()
        return self.out

    def wait(self) -> torch.Tensor:
        """Order the pending collective (if any) before whatever the current stream does next with ``out``."""
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self._staged is not None:
                self.out.copy_(self._staged)
                self._staged = None
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
