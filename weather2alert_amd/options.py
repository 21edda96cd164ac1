"""KernelOptions: the switches of HeatAlertVecEnv that select HOW a step / rollout / reward is computed, never WHAT --
every combination gives the reference's results (kernels agree to the order of the fp64 additions, or ~1e-7 for the
posterior-mean kernels). They exist for A/B measurements and tests; the defaults are the fastest choice on MI355X."""
from __future__ import annotations

from dataclasses import dataclass, fields, replace
from typing import Literal


@dataclass(frozen=True)
class KernelOptions:
    # "auto": batches of >= 131 072 envs run the 64-envs-per-wave kernel (csrc/w2a_step64.hip.h; in-kernel autoreset
    # included), smaller ones the 4-lanes-per-env kernel (the faster choice at each size); "classic" / "wide" force one
    # of them. While the batch is in lock step the 64-envs-per-wave kernel streams a 16-B packed mirror of the per-env
    # state instead of the 24-B canonical words (include/w2a.h, w2a_state_bytes); "unpacked" = "auto" without it.
    step_kernel: Literal["auto", "classic", "wide", "unpacked"] = "auto"
    # False: reward-only steps (no observation rows written)
    write_obs: bool = True
    # "gather": each step gathers the env's coefficient rows and sums the 28 terms. ("table", round 1's precomputed
    # logit table, was removed: slower at every batch size)
    reward_path: Literal["gather", "auto", "table"] = "gather"
    # rollout(): visit the envs in the order of their feature rows (w2a_rollout_order, one counting sort per episode)
    rollout_order: bool = True
    # rollout(): with that order, a batch in lock step and faithful semantics, the 27 action-independent terms of both
    # logits on the int8 matrix cores (csrc/w2a_rollout_mfma.hip.h)
    rollout_mfma: bool = True
    # reward_mode="posterior_mean": which kernel computes the contraction -- "matrix_i8" (int8 matrix cores on exact
    # fixed-point digits; the default: fastest, and a fixed choice keeps rewards bit-reproducible across runs and ranks),
    # "vector" (fp64 FMAs with DPP-broadcast coefficients), "matrix" (fp64 matrix cores), "auto" (times the three on the
    # env's own batch after the first reset and keeps the fastest -- a per-process choice unless the env is given
    # pm_sync_group=, which broadcasts the group's first rank's choice: a collective inside the first reset())
    pm_kernel: Literal["auto", "vector", "matrix", "matrix_i8"] = "matrix_i8"
    # episode_order="sorted": "fused" = w2a_reset_device_rng_sorted (draw keys, one stable radix sort of the coefficient
    # row, k_reset that writes every index the episode of its source env: no record is moved); "relabel" = round 5's
    # sequence w2a_reset_device_rng + w2a_sort_episodes (state permutation, three copies) + w2a_observe. Bit-identical.
    sorted_reset: Literal["fused", "relabel"] = "fused"

    def __post_init__(self):
        if self.step_kernel not in ("auto", "classic", "wide", "unpacked"):
            raise ValueError(f"step_kernel {self.step_kernel!r}")
        if self.reward_path == "table":
            raise ValueError("reward_path='table' (the precomputed logit table of round 1) was removed: it was slower "
                             "than the row-gather kernels at every batch size; use 'gather'")
        if self.reward_path not in ("gather", "auto"):
            raise ValueError(f"reward_path {self.reward_path!r}")
        if self.pm_kernel not in ("auto", "vector", "matrix", "matrix_i8"):
            raise ValueError(f"pm_kernel {self.pm_kernel!r}")
        if self.sorted_reset not in ("fused", "relabel"):
            raise ValueError(f"sorted_reset {self.sorted_reset!r}")

    @classmethod
    def names(cls) -> tuple:
        return tuple(f.name for f in fields(cls))

    def with_overrides(self, **kw) -> "KernelOptions":
        unknown = set(kw) - set(self.names())
        if unknown:
            raise TypeError(f"unexpected keyword argument(s) {sorted(unknown)}")
        return replace(self, **kw) if kw else self
