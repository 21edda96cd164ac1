"""Vectorised, MI355X-resident HeatAlertEnv with the reference's reset()/step() surface.

``HeatAlertVecEnv`` steps ``num_envs`` independent copies of the reference's
``weather2alert.env.HeatAlertEnv`` (``/root/reference/src/weather2alert/env.py``) inside
hand-written gfx950 kernels (``csrc/*.hip.h``, ``csrc/w2a_kernels.hip`` through the C ABI of
``include/w2a.h``). ``HeatAlertEnv`` is the ``num_envs=1`` drop-in with the reference's
exact constructor / ``reset`` / ``step`` signatures and NumPy-seed parity.

All reference quirks listed in SURVEY.md §3.3 (Q1-Q12) are reproduced; there is no CPU
fallback -- constructing an env without a ROCm device or without libw2a.so raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Literal

import numpy as np
import torch

from . import _ffi, stats as _stats
from .info import _LazyInfo
from .options import KernelOptions
from .rng import numpy_parity_episode
from .spaces import Box, Discrete
from .tables import CompiledTables, DeviceTables, compile_from_files

_ACT_CODES = {torch.int32: _ffi.ACT_I32, torch.int64: _ffi.ACT_I64, torch.uint8: _ffi.ACT_U8,
              torch.bool: _ffi.ACT_U8}
_BUDGET_MODES = {"less_than": _ffi.BUDGET_LESS_THAN, "centered": _ffi.BUDGET_CENTERED}
_cur_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)

try:  # a real Gymnasium VectorEnv when the package is importable; the same duck-typed surface otherwise
    from gymnasium.vector import VectorEnv as _VectorEnvBase
except ImportError:  # gymnasium is not a dependency of the reference's numerical path
    _VectorEnvBase = object
try:  # gymnasium >= 1.0 names the autoreset behaviour of a vector env with this enum (metadata["autoreset_mode"])
    from gymnasium.vector import AutoresetMode as _AutoresetMode
except ImportError:
    _AutoresetMode = None


def _autoreset_metadata(mode: str):
    """metadata["autoreset_mode"]: gymnasium's AutoresetMode member when the package has one, else the plain string."""
    if _AutoresetMode is None:
        return mode
    return {"same_step": _AutoresetMode.SAME_STEP, "next_step": _AutoresetMode.NEXT_STEP,
            "disabled": _AutoresetMode.DISABLED}[mode]


class HeatAlertVecEnv(_VectorEnvBase):
    """``num_envs`` heat-alert environments stepped in lock step on one MI355X.

    Constructor keeps the reference's keyword arguments (env.py:20-29) and adds:

    num_envs, device     batch size and ROCm device
    seed_mode            "device": every draw of reset() comes from the counter-based device RNG
                         keyed by (seed, global env id, episode number) -- same distributions as
                         the reference, not the same stream; "numpy_parity": the host replays
                         NumPy's Generator call sequence of env.py:145-177 per env (env i is
                         seeded with seed+i), bit-identical episode tuples to the reference.
    autoreset            "same_step" (finished envs restart inside the same step() call and the
                         returned observation is the new episode's first one; the finished
                         episode's return is in info["final_return"]) or "disabled". same_step is what SB3's
                         DummyVecEnv does around the reference env and Gymnasium's AutoresetMode.SAME_STEP
                         (``metadata["autoreset_mode"]``). No ``final_obs`` is returned: the reference's terminal
                         step hands back the PREVIOUS step's observation unchanged (env.py:257-262, Q6), i.e. a row
                         the caller already holds, and episodes end by termination only (no truncation to
                         bootstrap from). "next_step" is Gymnasium's AutoresetMode.NEXT_STEP (what
                         SyncVectorEnv does by default around the reference env): the terminal step returns done =
                         True with the stale observation; on the NEXT call a finished env ignores its action, starts
                         its next episode and returns that episode's first observation with reward 0 and done =
                         False. Per env the sequence of episodes is the same as with "same_step" (device seed mode
                         only).
    episode_order        "iid" (default): env i keeps its own independent draws, like N reference envs;
                         "sorted": after every (lock-step) reset the envs are relabelled so that env indices
                         follow the coefficient row. The batch holds exactly the same multiset of
                         episodes, only which index holds which episode changes (env identity is not preserved
                         across episodes); neighbouring envs then share table lines and the step kernel's
                         gathers become L2 hits. Needs seed_mode="device" and a uniform episode length.
    lockstep             how same-step autoreset is driven in device seed mode. True: every episode has the same
                         length and the whole batch resets together, so the host counts steps and launches the
                         reset kernel after the terminal step (the step kernel then runs its leaner, full-occupancy
                         variant). False: each env restarts inside the step kernel whenever it finishes (needed
                         after partial resets, and what a loop recorded into a hipGraph needs: no reset can be
                         launched between two recorded steps; a batch that is in lock step anyway keeps the packed
                         16-B state there too). None (default): True when the tables allow it; falls back to
                         False by itself after a masked reset.
                         hipGraphs: step() neither synchronises nor allocates and reads the day from device memory;
                         step once eagerly, then capture (torch.cuda.graph, or record_steps() below, which also reads the
                         status word behind every replay). Needs lockstep=False or autoreset="disabled": a loop whose
                         episode boundaries the host drives (the default in lock step: it counts days and launches the
                         reset) cannot be recorded, and step() raises instead of recording it. The library keeps the form
                         of the state the recorded steps work on current from then on (include/w2a.h, w2a_state_bytes).
    faithful / fixes     faithful=True (default) reproduces every reference quirk -- all parity
                         claims refer to this mode. ``fixes`` opts into individual corrections (faithful=False =
                         all of them): "alert_2wks" (Q1: the agent's 14-day count feeds the reward), "lag" (Q3:
                         alert_lag1 is yesterday's action), "penalty" (Q5: -1 for an alert attempted at budget),
                         "obs" (Q6: step() returns the next day's row), "augment" (Q8: the drawn similar county
                         supplies weather and coefficients), "budget" (Q9: per-episode budgets, no stickiness).
    reward_mode          "sampled" (default, the reference: one posterior draw per episode, env.py:160,209,216) or
                         "posterior_mean": every step's reward is the mean over ALL posterior draws of the env's
                         coefficient column -- the legacy env's eval mode (_deprecated/env.py:332-342) on today's
                         reward form -- one grouped contraction per step (csrc/w2a_posterior.hip.h).
                         Needs lock-step / disabled autoreset and faithful semantics.
    kernel               KernelOptions (weather2alert_amd/options.py): the switches that select HOW a step, a rollout
                         or the posterior-mean reward is computed -- step_kernel, write_obs, rollout_order, rollout_mfma,
                         pm_kernel, reward_path -- never what; each may also be passed by name (step_kernel="wide", ...).
                         Every combination gives the same results up to the order of the fp64 additions (~1e-7 for
                         the posterior-mean kernels); the defaults are the fastest choice on MI355X.
    pm_sync_group        pm_kernel="auto" only: a torch.distributed process group (True = the default group) whose ranks
                         must all use the kernel its first rank measured fastest -- the choice is then BROADCAST inside the
                         first reset() (a collective: every rank of the group must get there). Default None: no
                         communication, each process keeps its own choice; for one kernel on every rank without a
                         collective pass pm_kernel=<name> (the default "matrix_i8" is such a name).
    tables               pre-compiled CompiledTables (skips file loading)
    env_gid0             global id of env 0 (multi-GPU sharding keeps results shard-invariant)
    """

    metadata = {"autoreset_mode": _autoreset_metadata("same_step")}  # the default; instances carry their own mode
    spec = None
    render_mode = None
    closed = False

    def __init__(
        self,
        num_envs: int = 1,
        weights: str = "nn_full_medicare_all",
        years: list | None = None,
        fips_list: list | None = None,  # ignored, like the reference (overwritten at env.py:75)
        similar_climate_counties: bool = False,
        budget: int | None = None,
        data_dir: str | None = None,
        split: str = "65k",
        device: str | torch.device = "cuda:0",
        seed_mode: Literal["device", "numpy_parity"] = "device",
        autoreset: Literal["same_step", "next_step", "disabled"] = "same_step",
        tables: CompiledTables | DeviceTables | None = None,
        env_gid0: int = 0,
        episode_order: Literal["iid", "sorted"] = "iid",
        lockstep: bool | None = None,
        faithful: bool = True,
        fixes: set | list | None = None,
        reward_mode: Literal["sampled", "posterior_mean"] = "sampled",
        kernel: KernelOptions | None = None,
        pm_sync_group=None,
        **kernel_overrides,  # any field of KernelOptions by name (step_kernel=..., pm_kernel=..., write_obs=..., ...)
    ):
        # HOW things are computed (never what): one record, weather2alert_amd/options.py
        self.kernel = (kernel or KernelOptions()).with_overrides(**kernel_overrides)
        write_obs, step_kernel, pm_kernel = self.kernel.write_obs, self.kernel.step_kernel, self.kernel.pm_kernel
        rollout_order, rollout_mfma = self.kernel.rollout_order, self.kernel.rollout_mfma
        self._lib = _ffi.load()
        self.device = torch.device(device)
        if self.device.type != "cuda" or not torch.cuda.is_available():
            raise RuntimeError("HeatAlertVecEnv needs a ROCm GPU (device='cuda:N'); there is no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if num_envs <= 0:
            raise ValueError("num_envs must be positive")
        if seed_mode not in ("device", "numpy_parity"):
            raise ValueError(f"seed_mode {seed_mode!r}")
        if autoreset not in ("same_step", "next_step", "disabled"):
            raise ValueError(f"autoreset {autoreset!r}")
        if autoreset == "next_step" and seed_mode != "device":
            raise ValueError("autoreset='next_step' needs seed_mode='device'")
        self.num_envs = int(num_envs)
        self.similar_climate_counties = bool(similar_climate_counties)
        self.seed_mode = seed_mode
        self.autoreset = autoreset
        self.metadata = {**type(self).metadata, "autoreset_mode": _autoreset_metadata(autoreset)}
        self.env_gid0 = int(env_gid0)
        self.write_obs = bool(write_obs)
        self._ctor_budget = budget
        if isinstance(tables, DeviceTables):
            if tables.device != self.device:
                raise ValueError(f"tables live on {tables.device}, the env on {self.device}")
            self.dtables = tables
        else:
            ct = tables if tables is not None else compile_from_files(data_dir, weights, split, years)
            self.dtables = DeviceTables(ct, self.device)
        reward_path = "gather"
        allf = set(_ffi.FIX_BITS) | {"budget"}
        self.fixes = set(allf) if (not faithful and fixes is None) else set(fixes or ())
        if self.fixes - allf:
            raise ValueError(f"unknown fixes {sorted(self.fixes - allf)}; choose from {sorted(allf)}")
        self.reward_path = reward_path
        self.step_kernel = step_kernel
        if reward_mode not in ("sampled", "posterior_mean"):
            raise ValueError(f"reward_mode {reward_mode!r}")
        if reward_mode == "posterior_mean" and (self.fixes or step_kernel == "classic"):
            raise ValueError("reward_mode='posterior_mean' needs faithful semantics and the 64-envs-per-wave step kernel")
        self.reward_mode = reward_mode
        self.pm_kernel = pm_kernel
        self.pm_kernel_choice = None if pm_kernel == "auto" else pm_kernel  # decided after the first reset
        self.pm_kernel_timing_us: dict = {}
        self.pm_sync_group = pm_sync_group
        self.rollout_order = bool(rollout_order)  # rollout() visits the envs in feature-row order (speed only; A/B)
        # rollout(): the table-sourced part of the logits on the int8 matrix cores (needs the visiting order, a batch in
        # lock step and faithful semantics; speed only; A/B)
        self.rollout_mfma = bool(rollout_mfma)
        self._mfma_ws = None
        self.pm_rollout_kernel = True  # posterior_mean rollouts in one launch when possible (False: per-day launches)
        self._order_stale = True
        if episode_order not in ("iid", "sorted"):
            raise ValueError(f"episode_order {episode_order!r}")
        if episode_order == "sorted" and seed_mode != "device":
            raise ValueError("episode_order='sorted' needs seed_mode='device'")
        self.episode_order = episode_order
        ct = self.ct = self.dtables.ct
        self.fips_list = ct.fips_list
        self.valid_years = ct.years
        self.n_samples = ct.n_samples
        self.feature_names = ct.feature_names
        self._fips_pos = {f: i for i, f in enumerate(ct.fips_list)}
        # Q12: the true observation width (the reference declares len(columns)+2 = 33, env.py:88)
        self.single_observation_space = Box(-np.inf, np.inf, (ct.n_obs,), np.float32)
        self.single_action_space = Discrete(2)  # env.py:95
        self.observation_space = Box(-np.inf, np.inf, (self.num_envs, ct.n_obs), np.float32)
        self.action_space = self.single_action_space

        n, dev = self.num_envs, self.device
        with torch.cuda.device(dev):
            nbytes = self._lib.w2a_state_bytes(n)
            self._state = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)
            self._obs = torch.zeros((n, ct.n_obs), dtype=torch.float32, device=dev)
            self._reward = torch.zeros(n, dtype=torch.float32, device=dev)
            self._done = torch.zeros(n, dtype=torch.uint8, device=dev)
            self._final_return = torch.zeros(n, dtype=torch.float32, device=dev)
            self._truncated = torch.zeros(n, dtype=torch.bool, device=dev)
            h = C.c_void_p()
            _ffi.check(self._lib.w2a_create(C.byref(self.dtables.struct), n, self.env_gid0, self._state.data_ptr(),
                                            nbytes, self._status.data_ptr(), C.byref(h)), "w2a_create")
        self._h = h
        bits = sum(_ffi.FIX_BITS[k] for k in self.fixes if k in _ffi.FIX_BITS)
        if bits:
            _ffi.check(self._lib.w2a_set_semantics(h, bits), "w2a_set_semantics")
        if reward_mode == "posterior_mean" and self.pm_kernel_choice is not None:
            _ffi.check(self._lib.w2a_set_posterior_kernel(h, _ffi.PM_KERNELS[self.pm_kernel_choice]),
                       "w2a_set_posterior_kernel")
        # hot-path constants (step() is called millions of times: no per-call attribute chains)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._obs_ptr = self._obs.data_ptr() if self.write_obs else None
        self._rew_ptr, self._done_ptr = self._reward.data_ptr(), self._done.data_ptr()
        self._fr_ptr = self._final_return.data_ptr()
        self._done_bool = self._done.view(torch.bool)
        self._sort_ws = None
        self._order_ws = None
        self._group_ws = None
        if reward_mode == "posterior_mean":
            with torch.cuda.device(dev):
                self._group_ws = torch.empty(self._lib.w2a_group_workspace_bytes(n, ct.S, ct.n_samples), dtype=torch.uint8,
                                             device=dev)
        nd = np.unique(ct.n_days)
        uniform = len(nd) == 1 and nd[0] > 0
        if episode_order == "sorted":
            if not uniform:
                raise ValueError("episode_order='sorted' needs one episode length for every (county, year)")
            if lockstep is False:
                raise ValueError("episode_order='sorted' implies lockstep")
            with torch.cuda.device(dev):
                self._sort_ws = torch.empty(self._lib.w2a_sort_workspace_bytes(n), dtype=torch.uint8, device=dev)
        if lockstep and not uniform:
            raise ValueError("lockstep=True needs one episode length for every (county, year)")
        self._episode_len = int(nd[0]) if uniform else 0
        self._lockstep = bool(uniform and self.seed_mode == "device" and lockstep is not False)
        self._steps_in_episode = 0
        self._pending_reset = False  # next_step autoreset in lock step: the terminal step has run, the next call restarts
        self._reset_cfg = None
        self._set_step_mode()
        self._w2a_step = self._lib.w2a_step
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._sticky = [budget] * n  # host mirror of self.budget per env (numpy_parity mode, Q9)
        self._needs_reset = True
        self._last_opts: dict = {}
        self._np_random = None
        self._np_random_seed = None

    # ------------------------------------------------------------------ Gymnasium VectorEnv attributes
    @property
    def unwrapped(self):
        return self

    @property
    def np_random(self) -> np.random.Generator:
        """Host Generator seeded by the last explicit reset(seed=...) (Gymnasium convention). Episode draws come
        from the device RNG (seed_mode="device") or from per-env Generators (seed_mode="numpy_parity"), not from
        this one; it serves wrappers and samplers that expect the attribute."""
        if self._np_random is None:
            self._np_random_seed = int(np.random.SeedSequence().entropy % (2**63))
            self._np_random = np.random.default_rng(self._np_random_seed)
        return self._np_random

    @np_random.setter
    def np_random(self, value: np.random.Generator):
        self._np_random, self._np_random_seed = value, -1

    @property
    def np_random_seed(self):
        self.np_random  # noqa: B018  (initialises on first use)
        return self._np_random_seed

    # ------------------------------------------------------------------ plumbing
    def _set_step_mode(self):
        """in-kernel autoreset only when the batch is not in lock step (device seed mode)."""
        auto = self.autoreset in ("same_step", "next_step") and self.seed_mode == "device"
        self._dev_auto = auto and not self._lockstep
        # lock step: the host counts days and launches the reset kernel right after the terminal step (same_step) or at
        # the start of the following call (next_step: _pending_reset)
        self._host_auto = auto and self._lockstep and self.autoreset == "same_step"
        self._host_next = auto and self._lockstep and self.autoreset == "next_step"
        if not self._host_next:
            self._pending_reset = False  # envs left finished by a terminal step restart inside the kernel from here on
        if self.reward_mode == "posterior_mean" and self._dev_auto:
            raise ValueError("reward_mode='posterior_mean' cannot run with the in-kernel autoreset (batches that left "
                             "lock step); use autoreset='disabled' or whole-batch resets")
        self._pm = self.reward_mode == "posterior_mean"
        # episode boundaries driven from this class (a counted reset launch, a pending restart, NumPy draws on the host): such
        # a loop cannot be recorded into a hipGraph -- w2a_step then refuses to record instead of baking a reset into a
        # fixed position of the graph, or none at all (W2A_STEP_NO_CAPTURE; the query costs nothing extra)
        host_driven = self._host_auto or self._host_next or (self.autoreset == "same_step" and not self._dev_auto)
        self._step_flags = ((0 if self.write_obs else _ffi.STEP_NO_OBS) |
                            (_ffi.STEP_NO_CAPTURE if host_driven else 0) |
                            (_ffi.STEP_REWARD_GIVEN if self._pm else 0) |
                            (_ffi.STEP_CLASSIC if self.step_kernel == "classic" else 0) |
                            (_ffi.STEP_WIDE if self.step_kernel == "wide" else 0) |
                            (_ffi.STEP_UNPACKED if self.step_kernel == "unpacked" else 0) |
                            (_ffi.STEP_AUTORESET if self._dev_auto else 0) |
                            (_ffi.STEP_NEXT_STEP if self._dev_auto and self.autoreset == "next_step" else 0))

    @property
    def step_kernel_name(self) -> str:
        """The step kernel w2a_step launches for this env right now (mirrors the dispatch in csrc/w2a_kernels.hip)."""
        wide = self._pm or self.step_kernel == "wide" or (self.step_kernel in ("auto", "unpacked") and
                                                          self.num_envs >= _ffi.S64_MIN_ENVS)
        return "k_step64" if (wide and self.step_kernel != "classic") else "k_step"

    @property
    def last_step_kernel(self) -> str | None:
        """Which kernel the last step() launched, from the library's own record: "k_step" (4 lanes per env), "k_step64"
        (64 envs per wave, canonical state words), "k_step64<packed>" (the lock-step mirror) or None before the first."""
        return {0: "k_step", 1: "k_step64", 2: "k_step64<packed>"}.get(
            self._lib.w2a_query(self._h, _ffi.Q_LAST_STEP_KERNEL))

    @property
    def packed_state(self) -> bool:
        """The library currently holds the step state in its packed lock-step form (the last step ran the packed
        variant of the 64-envs-per-wave kernel)."""
        return bool(self._lib.w2a_query(self._h, _ffi.Q_PACKED_CURRENT)) and not bool(
            self._lib.w2a_query(self._h, _ffi.Q_CANONICAL_CURRENT))

    @property
    def last_rollout_kernel(self) -> str | None:
        """Which kernel the last sampled-reward rollout() launched: "k_rollout_mfma" (int8 matrix cores), "k_rollout64"
        (lane = env, vector ALU), "k_rollout" (4 lanes per env) or None before the first one."""
        return {0: "k_rollout", 1: "k_rollout64", 2: "k_rollout_mfma"}.get(
            self._lib.w2a_query(self._h, _ffi.Q_LAST_ROLLOUT_KERNEL))

    def _stream(self):
        if self._raw_stream is not None:
            return self._raw_stream(self._dev_index)
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self, **kwargs):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            self._lib.w2a_destroy(self._h)
            self._h = None
        self.closed = True

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def check_status(self) -> int:
        """Synchronise and raise if a kernel flagged bad inputs since the last call."""
        out = C.c_int32(0)
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.w2a_read_status(self._h, C.byref(out), self._stream()), "w2a_read_status")
        return self._raise_for_status(out.value)

    @property
    def status_word(self) -> torch.Tensor:
        """The device status word (int32 [1], W2A_ST_* bits, sticky until check_status() clears it): what a loop that
        must not synchronise -- a replayed hipGraph -- copies out or asserts on."""
        return self._status

    @staticmethod
    def _raise_for_status(bits: int) -> int:
        if bits & _ffi.ST_BAD_EPISODE:
            raise KeyError("reset: an episode tuple is out of range or has no data "
                           "(reference: KeyError at env.py:127 / ValueError at env.py:121)")
        if bits & _ffi.ST_BAD_ACTION:
            raise ValueError("step: actions must be 0 or 1 (action_space = Discrete(2))")
        if bits & _ffi.ST_STALE_GRAPH:
            raise RuntimeError("a replayed hipGraph holds step() calls on the packed lock-step form of this env's state, "
                               "which could not be kept current (a masked reset or a restored checkpoint since the "
                               "capture took the batch out of lock step): those replayed steps did nothing. Reset the "
                               "whole batch (or capture again) before replaying (include/w2a.h, w2a_state_bytes)")
        return bits

    def state(self) -> dict[str, torch.Tensor]:
        """Decoded per-env integer state (device tensors)."""
        return self._state_packed()[1]

    def _state_packed(self, fields=None):
        """(one int32 [n_fields, N] buffer, dict of per-field views into it): a single D2H copy fetches all.
        fields: decode only these (w2a_state_view: a NULL array is skipped)."""
        v = _ffi.StateView()
        names = _ffi.STATE_FIELDS if fields is None else [k for k in _ffi.STATE_FIELDS if k in fields]
        buf = torch.empty((len(names), self.num_envs), dtype=torch.int32, device=self.device)
        out = {}
        for i, k in enumerate(names):
            out[k] = buf[i].view(torch.float32) if k == "episode_return" else buf[i]
            setattr(v, k, buf[i].data_ptr())
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.w2a_get_state(self._h, C.byref(v), self._stream()), "w2a_get_state")
        return buf, out

    # ------------------------------------------------------------------ checkpoint / resume
    def state_dict(self) -> dict:
        """Everything needed to resume this batch bit-exactly: the packed per-env state (episode tuples,
        counters, returns, sticky budgets, episode numbers), the last observations and the host-side mirrors.
        (The reference env has no save/restore)"""
        self.state()  # brings the canonical arrays up to date (the library may hold the packed lock-step form)
        return {
            "state": self._state.clone(), "obs": self._obs.clone(), "final_return": self._final_return.clone(),
            "reward": self._reward.clone(), "done": self._done.clone(),
            "host": {"sticky": list(self._sticky), "steps_in_episode": self._steps_in_episode,
                     "lockstep": self._lockstep, "reset_cfg": self._reset_cfg, "last_opts": dict(self._last_opts),
                     "needs_reset": self._needs_reset, "pending_reset": self._pending_reset, "info_location": list(getattr(self, "_info_location", [])),
                     "num_envs": self.num_envs, "env_gid0": self.env_gid0,
                     # format of "state": w2a_state_bytes grew with ABI 17 (the mirror's day words); the canonical part
                     # at its front (header, cold, hot3, stepc) is what a restore needs
                     "abi_version": _ffi.ABI_VERSION, "state_bytes": int(self._state.numel()),
                     # pm_kernel="auto" measures; a resumed run must compute rewards with the same kernel to stay
                     # bit-exact with the run it continues (the kernels agree to ~1e-7, not to the last bit)
                     "pm_kernel_choice": self.pm_kernel_choice},
        }

    def load_state_dict(self, sd: dict) -> None:
        h = sd["host"]
        if h["num_envs"] != self.num_envs or h["env_gid0"] != self.env_gid0:
            raise ValueError("state_dict belongs to a different env batch (num_envs / env_gid0 differ)")
        hdr = self._state[:256].clone()  # slot map written by w2a_create for THIS handle
        src = sd["state"]
        if src.numel() == self._state.numel():
            self._state.copy_(src)
        else:
            # a checkpoint of another build of the library (the lock-step mirror behind the canonical words changed size):
            # the canonical part -- header, cold 16 B, hot3 12 B, stepc 12 B per env, each array on a 256-B boundary -- has
            # had the same layout in every ABI version; w2a_invalidate rebuilds everything derived from it
            a256 = lambda x: (x + 255) & ~255  # noqa: E731
            canon = 256 + a256(16 * self.num_envs) + 2 * a256(12 * self.num_envs)
            if src.numel() < canon or self._state.numel() < canon:
                raise ValueError(f"state_dict['state'] holds {src.numel()} bytes (written by ABI {h.get('abi_version', '<= 17')}); "
                                 f"this build (ABI {_ffi.ABI_VERSION}) needs at least the {canon} canonical bytes of "
                                 f"{self.num_envs} envs: not a checkpoint of this env batch")
            self._state[:canon].copy_(src[:canon])
        self._state[:256].copy_(hdr)
        # the state buffer changed behind the library: it forgets what it derived and scans the restored buffer for its
        # largest budget, sticky ones included (waits for this stream -- the one the copy above ran on -- and no other)
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.w2a_invalidate(self._h, self._stream()), "w2a_invalidate")
        self._obs.copy_(sd["obs"])
        self._final_return.copy_(sd["final_return"])
        self._reward.copy_(sd["reward"])
        self._done.copy_(sd["done"])
        self._sticky = list(h["sticky"])
        self._steps_in_episode, self._lockstep = h["steps_in_episode"], h["lockstep"]
        self._reset_cfg, self._last_opts, self._needs_reset = h["reset_cfg"], dict(h["last_opts"]), h["needs_reset"]
        if h["info_location"]:
            self._info_location = list(h["info_location"])
        self._set_step_mode()
        self._pending_reset = bool(h.get("pending_reset", False)) and self._host_next
        if self._reset_cfg is not None:
            with torch.cuda.device(self.device):
                _ffi.check(self._lib.w2a_set_autoreset(self._h, *self._reset_cfg), "w2a_set_autoreset")
        if self._pm and h.get("pm_kernel_choice") in _ffi.PM_KERNELS:  # continue on the kernel the checkpointed run used
            self.pm_kernel_choice = h["pm_kernel_choice"]
            _ffi.check(self._lib.w2a_set_posterior_kernel(self._h, _ffi.PM_KERNELS[self.pm_kernel_choice]),
                       "w2a_set_posterior_kernel")
        self._regroup()  # posterior_mean: the restored episode tuples need their own column grouping

    # ------------------------------------------------------------------ reset
    def _opt(self, options, key, default):
        v = None if options is None else options.get(key)
        return default if v is None else v

    def reset(self, seed: int | list | None = None, options: dict | None = None):
        """Gymnasium VectorEnv.reset. ``options`` carries the reference's reset kwargs
        (env.py:133-141): location, similar_climate_counties, budget, sample_budget,
        sample_budget_type -- scalars, or per-env sequences in numpy_parity mode -- plus
        "episodes": dict of int arrays (county_w, year_i, coef_col, sample, budget) to inject
        episode tuples directly, and "mask": bool[num_envs] to reset a subset."""
        options = dict(options or {})
        if seed is not None and not isinstance(seed, (list, tuple, np.ndarray)):
            self._np_random, self._np_random_seed = np.random.default_rng(int(seed)), int(seed)
        mask = options.get("mask")
        mask_t = None
        if mask is not None:
            mask_t = torch.as_tensor(np.asarray(mask), dtype=torch.uint8, device=self.device)
        obs_ptr = self._obs.data_ptr() if self.write_obs else None
        self._last_opts = {k: v for k, v in options.items() if k not in ("mask", "episodes")}
        if self._lockstep and (mask is not None or "episodes" in options):
            if self.episode_order == "sorted":
                raise ValueError("episode_order='sorted' resets the whole batch with the device RNG only")
            self._lockstep = False  # partial / injected resets: envs finish at different times from now on
            self._set_step_mode()
        if "episodes" in options:
            self._reset_tuples(options["episodes"], mask_t, obs_ptr)
            if self.seed_mode == "device" and self.autoreset in ("same_step", "next_step"):
                # later episodes of these envs come from the device RNG (in-kernel autoreset)
                self._reset_cfg = self._device_cfg(seed, self._last_opts)
                with torch.cuda.device(self.device):
                    _ffi.check(self._lib.w2a_set_autoreset(self._h, *self._reset_cfg), "w2a_set_autoreset")
        elif self.seed_mode == "numpy_parity":
            self._reset_numpy_parity(seed, options, mask, mask_t, obs_ptr)
        else:
            self._reset_device(seed, options, mask_t, obs_ptr)
        self._needs_reset = False
        return self._obs, self._info()

    def _reset_tuples(self, ep: dict, mask_t, obs_ptr):
        ct, n = self.ct, self.num_envs
        arrs = {}
        for k in ("county_w", "year_i", "coef_col", "sample"):
            a = np.broadcast_to(np.asarray(ep[k], dtype=np.int64), (n,))
            arrs[k] = a
        lim = {"county_w": ct.S_w, "year_i": ct.Y, "coef_col": ct.S, "sample": ct.n_samples}
        sel = np.ones(n, bool) if mask_t is None else mask_t.cpu().numpy().astype(bool)
        for k, a in arrs.items():
            if ((a[sel] < 0) | (a[sel] >= lim[k])).any():
                raise KeyError(f"reset: {k} outside [0, {lim[k]})")
        rows = arrs["county_w"] * ct.Y + arrs["year_i"]
        if (ct.n_days[rows[sel]] <= 0).any():
            raise KeyError("reset: (county, year) has no data (reference: KeyError at env.py:127)")
        if ep.get("budget") is None:
            bud = ct.B0[rows].astype(np.int64)
        else:
            bud = np.broadcast_to(np.asarray(ep["budget"], dtype=np.int64), (n,))
        dev = self.device
        t = {k: torch.from_numpy(np.array(a, dtype=np.int32)).to(dev) for k, a in arrs.items()}
        tb = torch.from_numpy(np.array(bud, dtype=np.int32)).to(dev)
        with torch.cuda.device(dev):
            _ffi.check(self._lib.w2a_reset(self._h, t["county_w"].data_ptr(), t["year_i"].data_ptr(),
                                           t["coef_col"].data_ptr(), t["sample"].data_ptr(), tb.data_ptr(),
                                           None if mask_t is None else mask_t.data_ptr(), obs_ptr, self._stream()),
                       "w2a_reset")
        self._keep = (t, tb, mask_t)  # keep inputs alive until the async launch has consumed them
        self._regroup()

    def _per_env(self, v, i):
        if isinstance(v, (list, tuple, np.ndarray)):
            return v[i]
        return v

    def _reset_numpy_parity(self, seed, options, mask, mask_t, obs_ptr):
        """Host replay of env.py:143-178 per env with NumPy's own Generator."""
        ct, n = self.ct, self.num_envs
        loc_o = options.get("location")
        aug_o = self._opt(options, "similar_climate_counties", self.similar_climate_counties)
        bud_o = options.get("budget")
        sb_o = self._opt(options, "sample_budget", False)
        sbt_o = self._opt(options, "sample_budget_type", "less_than")
        cw, yi, cc, sm, bd = (np.zeros(n, np.int64) for _ in range(5))
        self._info_location = getattr(self, "_info_location", [None] * n)
        sel = np.ones(n, bool) if mask is None else np.asarray(mask, bool)
        for i in range(n):
            if not sel[i]:
                continue
            if seed is None:
                s = np.random.randint(0, 10000)  # env.py:143-144
            elif isinstance(seed, (list, tuple, np.ndarray)):
                s = int(seed[i])
            else:
                s = int(seed) + i
            aug = bool(self._per_env(aug_o, i))
            w, y_i, li, ci, b, self._info_location[i] = numpy_parity_episode(
                ct, s, self._per_env(loc_o, i), aug, self._sticky[i], self._per_env(bud_o, i),
                bool(self._per_env(sb_o, i)), self._per_env(sbt_o, i), "augment" in self.fixes)
            self._sticky[i] = self._ctor_budget if "budget" in self.fixes else b
            cw[i], yi[i], cc[i], sm[i], bd[i] = w, y_i, li, ci, b
        self._reset_tuples(dict(county_w=cw, year_i=yi, coef_col=cc, sample=sm, budget=bd), mask_t, obs_ptr)

    def _device_cfg(self, seed, options):
        loc = options.get("location")
        if loc is None:
            loc_i = -1
        else:
            if loc not in self._fips_pos:
                raise ValueError(f"{loc!r} is not in list")
            loc_i = self._fips_pos[loc]
            if self.ct.fips_to_weather[loc_i] < 0:
                raise KeyError(loc)
        aug = bool(self._opt(options, "similar_climate_counties", self.similar_climate_counties))
        if aug and loc_i >= 0 and self.ct.sim_cnt[loc_i] <= 0:
            raise KeyError(loc)  # county absent from the confounders: confounders.loc[fips] (datautils.py:123)
        bk = self._ctor_budget if self._ctor_budget is not None else options.get("budget")
        mode = _ffi.BUDGET_FIXED
        if self._opt(options, "sample_budget", False):
            typ = self._opt(options, "sample_budget_type", "less_than")
            if typ not in _BUDGET_MODES:
                raise ValueError(f"sample_budget_type {typ!r}")
            mode = _BUDGET_MODES[typ]
        if seed is None:
            seed = int(np.random.randint(0, 2**31 - 1))
        return int(seed) & (2**64 - 1), loc_i, int(aug), -1 if bk is None else int(bk), mode, int("budget" not in self.fixes)

    def _reset_device(self, seed, options, mask_t, obs_ptr):
        ct = self.ct
        if (ct.fips_to_weather < 0).any() or (ct.n_days <= 0).any():
            raise KeyError("seed_mode='device' needs state tables for every fips_list county and year "
                           "(the reference raises KeyError at env.py:127 when it draws a missing pair)")
        cfg = self._device_cfg(seed, options)
        if cfg[2] and (ct.sim_cnt <= 0).any() and cfg[1] < 0:
            raise KeyError("a fips_list county is missing from the confounders table (datautils.py:123)")
        if self.episode_order == "sorted" and mask_t is not None:
            raise ValueError("episode_order='sorted' resets the whole batch; masks are not supported")
        self._reset_cfg = cfg
        # reset() re-seeds (explicit seed, or a fresh one for seed=None like env.py:143-145), so the per-env episode
        # counters restart: equal seeds give equal episodes. Autoresets advance the counters instead.
        self._launch_device_reset(mask_t, obs_ptr, restart=True)
        with torch.cuda.device(self.device):
            _ffi.check(self._lib.w2a_set_autoreset(self._h, *cfg), "w2a_set_autoreset")
        self._keep = (mask_t,)

    def _launch_device_reset(self, mask_t, obs_ptr, restart=False):
        """Device-RNG reset of the batch (restart: episode number 0 for a re-seeding reset(); else the next episode
        number per env, i.e. an autoreset); in sorted mode followed by the relabelling sort and the observation
        pass. Asynchronous, no host sync."""
        lib, st = self._lib, self._stream()
        srt = self.episode_order == "sorted"
        with torch.cuda.device(self.device):
            if srt and self.kernel.sorted_reset == "fused":
                # draw keys -> stable radix sort -> k_reset with "index e receives the episode env src[e] draws": the
                # relabelling without moving a record (1 = the key does not fit 32 bits: the three calls below)
                rc = lib.w2a_reset_device_rng_sorted(self._h, *self._reset_cfg, int(restart), obs_ptr, self._sort_ws.data_ptr(),
                                                     self._sort_ws.numel(), st)
                if rc not in (0, 1):
                    _ffi.check(rc, "w2a_reset_device_rng_sorted")
                srt = rc == 1
                done = rc == 0
            else:
                done = False
            if not done:
                _ffi.check(lib.w2a_reset_device_rng(self._h, *self._reset_cfg, int(restart), None if mask_t is None else
                                                    mask_t.data_ptr(), None if srt else obs_ptr, st),
                           "w2a_reset_device_rng")
            if srt and not done:
                _ffi.check(lib.w2a_sort_episodes(self._h, self._sort_ws.data_ptr(), self._sort_ws.numel(), st),
                           "w2a_sort_episodes")
                if obs_ptr is not None:
                    _ffi.check(lib.w2a_observe(self._h, obs_ptr, st), "w2a_observe")
        self._steps_in_episode = 0
        self._pending_reset = False
        self._regroup()

    def _regroup(self):
        """posterior_mean: env ids sorted by coefficient column for the grouped GEMM; after EVERY reset."""
        self._order_stale = True  # rollout(): the visiting order by feature row belongs to the previous episode
        if self._group_ws is not None:
            if max(int(self.ct.B0.max()), int(self._ctor_budget or 0), int(self._last_opts.get("budget") or 0)) > 65535:
                raise ValueError("reward_mode='posterior_mean' packs remaining_budget into 16 bits: budgets <= 65535")
            with torch.cuda.device(self.device):
                _ffi.check(self._lib.w2a_group_by_column(self._h, self._group_ws.data_ptr(), self._group_ws.numel(),
                                                         self._stream()), "w2a_group_by_column")
            if self.pm_kernel_choice is None:
                self._pick_pm_kernel()

    def _pick_pm_kernel(self, reps: int = 3):
        """pm_kernel="auto": time w2a_posterior_mean_reward with each kernel on this env's own grouped batch (HIP
        events on the launch stream, best of `reps`; the call only writes the reward buffer and its scratch) and keep
        the faster one. Runs once per env, right after the first grouping."""
        act = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
        best = {}
        with torch.cuda.device(self.device):
            for name, code in _ffi.PM_KERNELS.items():
                _ffi.check(self._lib.w2a_set_posterior_kernel(self._h, code), "w2a_set_posterior_kernel")
                times = []
                for _ in range(reps + 1):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    _ffi.check(self._lib.w2a_posterior_mean_reward(self._h, act.data_ptr(), _ffi.ACT_I32, self._rew_ptr,
                                                                   self._stream()), "w2a_posterior_mean_reward")
                    e1.record()
                    e1.synchronize()
                    times.append(e0.elapsed_time(e1) * 1e3)
                best[name] = min(times[1:])
            self.pm_kernel_timing_us = best
            self.pm_kernel_choice = min(best, key=best.get)
            # The kernels differ in the last bits, and the device RNG's shard invariance (env_gid0) would be worth little
            # if the same global env id got other reward bits on another rank: with pm_sync_group= every rank of that
            # group takes the group's first rank's choice. A COLLECTIVE -- so only when the caller asked for it (every
            # rank of the group must then build such an env and reach its first reset()); without it "auto" is a
            # per-process choice (say pm_kernel=<name> for a fixed one).
            if self.pm_sync_group is not None:
                import torch.distributed as td

                names = list(_ffi.PM_KERNELS)
                pick = torch.tensor([names.index(self.pm_kernel_choice)], dtype=torch.int32, device=self.device)
                group = None if self.pm_sync_group is True else self.pm_sync_group
                td.broadcast(pick, src=td.get_global_rank(group, 0) if group is not None else 0, group=group)
                self.pm_kernel_choice = names[int(pick.item())]
            _ffi.check(self._lib.w2a_set_posterior_kernel(self._h, _ffi.PM_KERNELS[self.pm_kernel_choice]),
                       "w2a_set_posterior_kernel")
            self.check_status()  # zero actions on a fresh episode raise no status bit; clear what a mid-episode call may

    # ------------------------------------------------------------------ step
    def step(self, actions):
        """Gymnasium VectorEnv.step: (obs, reward, terminated, truncated, info), all device
        tensors that are reused by the next call (clone them to keep a copy)."""
        if self._needs_reset:
            raise RuntimeError("call reset() before step()")
        if (type(actions) is not torch.Tensor or actions.device != self.device or actions.dtype not in _ACT_CODES
                or actions.numel() != self.num_envs or not actions.is_contiguous()):
            actions = self._coerce_actions(actions)
        if _cur_device() != self._dev_index:  # kernels launch on the current device: it must be this env's
            with torch.cuda.device(self.device):
                return self.step(actions)
        if self._pending_reset:
            # autoreset="next_step" in lock step: the previous call was the terminal step of every env, so this call is
            # their restart -- actions ignored, first observations of the next episodes, reward 0, nobody done
            self._launch_device_reset(None, self._obs_ptr)
            self._reward.zero_()
            self._done.zero_()
            self._keep_act = actions
            return self._obs, self._reward, self._done_bool, self._truncated, _LazyInfo(self)
        if self._pm:  # today's reward of every env as the mean over all posterior draws, from the pre-step state
            rc = self._lib.w2a_posterior_mean_reward(self._h, actions.data_ptr(), _ACT_CODES[actions.dtype],
                                                     self._rew_ptr, self._stream())
            if rc != 0:
                _ffi.check(rc, "w2a_posterior_mean_reward")
        rc = self._w2a_step(self._h, actions.data_ptr(), _ACT_CODES[actions.dtype], self._obs_ptr, self._rew_ptr,
                            self._done_ptr, self._fr_ptr, self._step_flags, self._stream())
        if rc != 0:
            _ffi.check(rc, "w2a_step")
        self._keep_act = actions
        done = self._done_bool
        if self._host_auto:  # lock step: the host counts days and launches the reset after the terminal step
            self._steps_in_episode += 1
            if self._steps_in_episode == self._episode_len:
                self._launch_device_reset(None, self._obs_ptr)
        elif self._host_next:  # ... or leaves it to the next call
            self._steps_in_episode += 1
            if self._steps_in_episode == self._episode_len:
                self._pending_reset = True
        elif self.autoreset == "same_step" and not self._dev_auto:
            d = done.cpu().numpy()
            if d.any():  # host-side autoreset (numpy_parity): fresh global-RNG seeds like reset(seed=None)
                self._reset_numpy_parity(None, self._last_opts, d, torch.as_tensor(d.astype(np.uint8),
                                         device=self.device), self._obs_ptr)
        return self._obs, self._reward, done, self._truncated, _LazyInfo(self)

    def record_steps(self, one_day, days: int, warmup: bool = True):
        """Record `days` calls of one_day() -- a policy on device tensors + self.step(actions) -- into a hipGraph and
        return a weather2alert_amd.graph.RecordedSteps: .replay() launches the block and reads the device status word
        BEHIND every replay (asynchronously: replay k checks what replay k-1 left), so a block that could not run -- the
        mirror of a packed lock-step batch marked stale by a masked reset or a restored checkpoint, W2A_ST_STALE_GRAPH --
        or bad actions raise at the next replay instead of training on frozen observations."""
        from .graph import RecordedSteps

        return RecordedSteps(self, one_day, days, warmup)

    # ------------------------------------------------------------------ rollout
    def rollout(self, policy: dict, n_steps: int | None = None, alert_mask: bool = False) -> dict:
        """Run a built-in policy inside the kernel for ``n_steps`` days (default: to the end of the episode)
        without returning to Python between days (replaces loops like env.py:265-277). With
        reward_mode="posterior_mean" the whole rollout is one launch of k_pm_rollout (vector kernel, <= 112 posterior
        draws, reference schema); otherwise a host loop of policy kernel + reward kernels + step kernel per day (same
        policies, same outputs): the legacy eval mode's evaluation sweep.

        policy: {"kind": "never" | "always"} |
                {"kind": "bernoulli", "p": 0.1, "seed": 0} |
                {"kind": "threshold", "feature": "heat_qi", "threshold": 0.9, "lag": 1} |
                {"kind": "table", "table": uint8 [T, R]}   (action = table[day][min(remaining_budget, R-1)])
                plus optional "require_budget": True (never attempt an alert with no budget left).
        The threshold policy sees the lagging observation the reference's agent would see (Q6; lag=0 reads
        today's row instead). Returns device tensors: "return" (rewards summed over the days run), "alerts",
        "attempts_over_budget", "final_return" (episode return of envs that finished), "done", and with
        alert_mask=True "alert_days" bool [N, T]. In lock-step same_step-autoreset mode a finished batch is
        reset, so consecutive calls evaluate consecutive episodes."""
        if self._needs_reset:
            raise RuntimeError("call reset() before rollout()")
        if self._pending_reset:  # next_step autoreset in lock step: the finished batch restarts before anything runs
            self._launch_device_reset(None, self._obs_ptr)
        ct = self.ct
        kind = policy.get("kind")
        if kind not in _ffi.POLICY_KINDS:
            raise ValueError(f"policy kind {kind!r}")
        p = _ffi.Policy()
        p.kind = _ffi.POLICY_KINDS[kind]
        p.p = float(policy.get("p", 0.0))
        p.require_budget = int(bool(policy.get("require_budget", False)))
        p.seed = int(policy.get("seed", 0)) & (2**64 - 1)
        p.obs_lag = int(policy.get("lag", 1))
        keep = None
        if kind == "threshold":
            feat = policy["feature"]
            if feat not in ct.feature_names:
                raise KeyError(feat)
            p.obs_col, p.threshold = ct.feature_names.index(feat), float(policy["threshold"])
        if kind == "table":
            keep = torch.as_tensor(policy["table"], device=self.device).to(torch.uint8).contiguous()
            if keep.dim() != 2 or keep.shape[0] < ct.T:
                raise ValueError(f"policy table must be [T >= {ct.T}, R]")
            p.table, p.table_R = keep.data_ptr(), int(keep.shape[1])
        n, dev = self.num_envs, self.device
        steps = int(n_steps) if n_steps is not None else ct.T
        out = {"return": torch.empty(n, dtype=torch.float32, device=dev),
               "alerts": torch.empty(n, dtype=torch.int32, device=dev),
               "attempts_over_budget": torch.empty(n, dtype=torch.int32, device=dev)}
        words = (ct.T + 31) // 32
        mask = torch.empty((n, words), dtype=torch.int32, device=dev) if alert_mask else None
        amask = torch.empty((n, words), dtype=torch.int32, device=dev) if alert_mask else None
        snap = torch.full((n,), float("nan"), dtype=torch.float32, device=dev) if alert_mask else None
        st0 = self.state() if (alert_mask or self._pm) else None
        with torch.cuda.device(dev):
            if not self._pm and getattr(self, "_order_stale", True) and self.rollout_order:
                if self._order_ws is None:
                    # attached from here on: every later whole-batch reset leaves the feature-row counts and per-env ranks
                    # in it (the first pass of the counting sort, inside k_reset), w2a_rollout_order only scans and places
                    self._order_ws = torch.empty(self._lib.w2a_rollout_order_workspace_bytes(n, ct.S_w * ct.Y), dtype=torch.uint8, device=dev)
                    _ffi.check(self._lib.w2a_rollout_order_attach(self._h, self._order_ws.data_ptr(), self._order_ws.numel()),
                               "w2a_rollout_order_attach")
                _ffi.check(self._lib.w2a_rollout_order(self._h, self._order_ws.data_ptr(), self._order_ws.numel(),
                                                       self._stream()), "w2a_rollout_order")
                self._order_stale = False
                if self.rollout_mfma and not (self.fixes - {"budget"}):
                    if self._mfma_ws is None:
                        self._mfma_ws = torch.empty(self._lib.w2a_rollout_mfma_workspace_bytes(
                            n, ct.S_w * ct.Y, ct.S, ct.n_samples), dtype=torch.uint8, device=dev)
                    _ffi.check(self._lib.w2a_rollout_mfma_prepare(self._h, self._mfma_ws.data_ptr(), self._mfma_ws.numel(),
                                                                  self._stream()), "w2a_rollout_mfma_prepare")
            if self._pm:
                steps = self._rollout_posterior_mean(p, steps, out, mask, amask, words, snap, st0)
            else:
                _ffi.check(self._lib.w2a_rollout(self._h, C.byref(p), steps, out["return"].data_ptr(),
                                                 out["alerts"].data_ptr(), out["attempts_over_budget"].data_ptr(),
                                                 None if mask is None else mask.data_ptr(),
                                                 None if amask is None else amask.data_ptr(), words, self._fr_ptr,
                                                 None if snap is None else snap.data_ptr(), self._stream()), "w2a_rollout")
        self._keep_pol = keep
        # only what this call returns is decoded (sixteen arrays of N int32 otherwise: 64 MB of writes at 1 M envs)
        st = self.state() if mask is not None else self._state_packed(("finished",))[1]
        out["done"] = st["finished"].bool()  # the terminal step has run (t stops at n_days-1 before AND after it)
        out["final_return"] = self._final_return.clone()  # meaningful where out["done"]
        if mask is not None:
            bits = torch.arange(32, device=dev, dtype=torch.int32)
            unpack = lambda m: (((m.unsqueeze(-1) >> bits) & 1).reshape(n, words * 32)[:, : ct.T]).bool()  # noqa: E731
            out["alert_days"] = unpack(mask)      # the reference's actual_alert_buffer (env.py:248), per day
            out["attempt_days"] = unpack(amask)   # its attempted_alert_buffer (env.py:239)
            out["return_snapshot"] = snap         # running return when the callbacks read the env (t == n_days - 2)
            out["n_days"], out["budget"] = st["n_days"], st["budget"]
            out["first_day"] = st0["t"]           # day index this call started from (0 for whole episodes)
            out["year"] = torch.as_tensor(ct.years, dtype=torch.int32, device=dev)[st["year_i"].long()]
        if self._host_auto or self._host_next:
            self._steps_in_episode = min(self._steps_in_episode + steps, self._episode_len)
            if self._steps_in_episode >= self._episode_len:
                if self._host_auto:
                    self._launch_device_reset(None, self._obs_ptr)
                else:
                    self._pending_reset = True
        return out

    def _rollout_posterior_mean(self, p, steps, out, mask, amask, words, snap, st0) -> int:
        """rollout() with reward_mode="posterior_mean": one launch of k_pm_rollout when it applies, else per day
        w2a_policy_actions (the policy and counters of k_rollout), w2a_posterior_mean_reward, w2a_step(REWARD_GIVEN |
        SKIP_FINISHED). Every env is handled on its own day and its own episode length (batches that left lock step
        under autoreset="disabled", ragged episode lengths): envs whose episode is over take no part. Returns the
        number of days run = the most any env had left, capped by `steps`."""
        for k in ("return", "alerts", "attempts_over_budget"):
            out[k].zero_()
        for m in (mask, amask):
            if m is not None:
                m.zero_()
        left = torch.where(st0["finished"].bool(), torch.zeros_like(st0["t"]), st0["n_days"] - st0["t"])
        steps = min(steps, int(left.max()))
        lib, h, stream = self._lib, self._h, self._stream()
        if steps and self.pm_rollout_kernel:  # the whole rollout in one launch, when the kernel applies
            rc = lib.w2a_rollout_posterior_mean(h, C.byref(p), steps, out["return"].data_ptr(), out["alerts"].data_ptr(),
                                                out["attempts_over_budget"].data_ptr(),
                                                None if mask is None else mask.data_ptr(),
                                                None if amask is None else amask.data_ptr(), words, self._fr_ptr,
                                                None if snap is None else snap.data_ptr(), stream)
            if rc == 0:
                return steps
            if rc != 1:
                _ffi.check(rc, "w2a_rollout_posterior_mean")
        act = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        flags = self._step_flags | _ffi.STEP_NO_OBS | _ffi.STEP_SKIP_FINISHED
        nd2 = st0["n_days"] - 2
        for _ in range(steps):
            _ffi.check(lib.w2a_policy_actions(h, C.byref(p), act.data_ptr(), out["alerts"].data_ptr(),
                                              out["attempts_over_budget"].data_ptr(),
                                              None if mask is None else mask.data_ptr(),
                                              None if amask is None else amask.data_ptr(), words, stream),
                       "w2a_policy_actions")
            _ffi.check(lib.w2a_posterior_mean_reward(h, act.data_ptr(), _ffi.ACT_I32, self._rew_ptr, stream),
                       "w2a_posterior_mean_reward")
            # like k_rollout, no observation rows are written (the next step() or reset() brings them up to date);
            # finished envs are skipped: their reward comes back as 0 and their state stays
            _ffi.check(self._w2a_step(h, act.data_ptr(), _ffi.ACT_I32, None, self._rew_ptr, self._done_ptr,
                                      self._fr_ptr, flags, stream), "w2a_step")
            out["return"] += self._reward
            if snap is not None:  # per env: the moment the reference's callbacks read it (t == n_days - 2 after a step)
                st = self.state()
                hit = (st["t"] == nd2) & (st["finished"] == 0) & torch.isnan(snap)
                torch.where(hit, st["episode_return"], snap, out=snap)
        self._keep_act = act
        return steps

    # ------------------------------------------------------------------ statistics (weather2alert_amd/stats.py)
    episode_stats = staticmethod(_stats.episode_stats)
    callback_stats = staticmethod(_stats.callback_stats)
    episode_rows = staticmethod(_stats.episode_rows)
    write_episode_csv = staticmethod(_stats.write_episode_csv)
    CSV_FIELDS = _stats.CSV_FIELDS

    def _coerce_actions(self, actions):
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.device)
        elif actions.device != self.device:
            actions = actions.to(self.device)
        if actions.dtype not in _ACT_CODES:
            actions = actions.to(torch.int32)
        if actions.numel() != self.num_envs:
            raise ValueError(f"expected {self.num_envs} actions, got {actions.numel()}")
        return actions.contiguous()

    def _info(self):
        return _LazyInfo(self)


def __getattr__(name):  # weather2alert_amd.env.HeatAlertEnv: the drop-in lives in dropin.py (it builds on this module)
    if name == "HeatAlertEnv":
        from .dropin import HeatAlertEnv

        return HeatAlertEnv
    raise AttributeError(name)
