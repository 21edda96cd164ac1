"""Model of one libw2a handle driven through weather2alert_amd.HeatAlertVecEnv (seed_mode="device") under ARBITRARY call
sequences -- TEST INFRASTRUCTURE ONLY (like everything under oracle/): imported by tests/ and tools/sequence_fuzz.py as
the checker, never by the product path.

The reference env allows reset() and step() to interleave arbitrarily (/root/reference/src/weather2alert/env.py:133-184,
238-262: reset re-draws the episode and zeroes the buffers whenever it is called; step after `done` keeps recomputing
the last day). `HandleModel` states what N such envs hold after any sequence of the build's calls:

  * per-env arithmetic is the float64 `VectorOracle` (heatalert_oracle.py, pinned bit-exact to the reference's goldens);
  * the episode draws of device-RNG resets and autoresets are the restatement of the build's counter RNG
    (`draw_episodes`: csrc/w2a_common.hip.h draw_episode, vectorised; no reference counterpart -- the distributions are
    the reference's, env.py:145-177);
  * masked resets, injected tuples, the three autoreset modes (same_step / next_step / disabled), partial rollouts,
    episode_order="sorted" (a stable relabelling), checkpoints;
  * and a mirror of what the HANDLE should know after each call (`known_day`, which step / rollout kernel a call must
    launch, whether the packed lock-step form is current) so that w2a_query can be compared with what the sequence
    implies -- a wrong validity flag in the library means silently wrong rewards.
"""
from __future__ import annotations

import copy

import numpy as np

from . import heatalert_oracle as O

_M64 = (1 << 64) - 1
_PHI = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)
BUDGET_FIXED, BUDGET_LESS_THAN, BUDGET_CENTERED = 0, 1, 2
S64_MIN_ENVS = 131072
FIX_BITS = ("alert_2wks", "lag", "penalty", "obs", "augment")  # the W2A_FIX_* bits; "budget" is the sticky=0 argument


def _streams(seed: int, gid: np.ndarray, episode_no: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        h = O._mix64_np(np.uint64(seed & _M64) + _PHI * (gid.astype(np.uint64) + np.uint64(1)))
        return O._mix64_np(h ^ (episode_no.astype(np.uint64) * _C1 + _C2))


def _bounded(stream: np.ndarray, slot: int, n) -> np.ndarray:
    with np.errstate(over="ignore"):
        u = O._mix64_np(stream + np.uint64(slot + 1) * _PHI) >> np.uint64(32)
        return ((u * np.asarray(n, np.uint64)) >> np.uint64(32)).astype(np.int64)


def draw_episodes(ct, cfg, gid, episode_no, sticky_in, fix_augment: bool):
    """csrc/w2a_common.hip.h draw_episode for arrays of envs. cfg = (seed, location, augment, budget_kw, sample_mode,
    sticky) as HeatAlertVecEnv._device_cfg builds it. Returns county_w, year_i, coef_col, sample, budget, sticky_out."""
    seed, loc, aug, budget_kw, mode, sticky = cfg
    n = len(gid)
    st = _streams(seed, np.asarray(gid), np.asarray(episode_no))
    county = _bounded(st, 0, ct.S) if loc < 0 else np.full(n, loc, np.int64)
    coef_col = county.copy()
    if aug:
        ns = np.asarray(ct.sim_cnt)[county].astype(np.int64)
        assert (ns > 0).all(), "model: county without similar counties"
        coef_col = _bounded(st, 1, ns)
        if fix_augment:
            county = np.asarray(ct.sim_idx)[np.asarray(ct.sim_ptr)[county] + coef_col].astype(np.int64)
            coef_col = county.copy()
    year_i = _bounded(st, 2, ct.Y)
    sample = _bounded(st, 3, ct.n_samples)
    cw = np.asarray(ct.fips_to_weather)[county].astype(np.int64)
    row = cw * ct.Y + year_i
    b = np.asarray(ct.B0)[row].astype(np.int64) if budget_kw < 0 else np.full(n, budget_kw, np.int64)
    if sticky:
        b = np.where(sticky_in >= 0, sticky_in, b)
    b = np.maximum(b, 0)
    if mode == BUDGET_LESS_THAN:
        b = _bounded(st, 4, b + 1)
    elif mode == BUDGET_CENTERED:
        lo = (0.5 * b.astype(np.float64)).astype(np.int64)
        hi = (1.5 * b.astype(np.float64) + 1.0).astype(np.int64)
        b = lo + _bounded(st, 4, hi - lo)
    sticky_out = b.copy() if sticky else np.full(n, -1, np.int64)
    return cw, year_i, coef_col, sample, b, sticky_out


_V_FIELDS = ("t", "used", "streak", "hist", "last_actual", "at_budget", "obs")


class HandleModel:
    def __init__(self, sd, ct, n, gid0=0, fixes=(), reward_mode="sampled", autoreset="same_step", augment=False,
                 ctor_budget=None, episode_order="iid", lockstep=None, write_obs=True, step_kernel="auto",
                 rollout_order=True, rollout_mfma=True, pm_kernel="matrix_i8"):
        self.ct, self.n, self.gid0 = ct, int(n), int(gid0)
        self.fixes = set(fixes)
        self.fixbits = bool(self.fixes & set(FIX_BITS))
        self.reward_mode, self.autoreset, self.augment = reward_mode, autoreset, bool(augment)
        self.pm = reward_mode == "posterior_mean"
        self.ctor_budget, self.episode_order, self.write_obs = ctor_budget, episode_order, bool(write_obs)
        self.step_kernel, self.rollout_order, self.rollout_mfma, self.pm_kernel = step_kernel, rollout_order, rollout_mfma, pm_kernel
        self.V = O.VectorOracle(O.RefData.from_synth(sd), sd.fips_weather, sd.years,
                                fixes=self.fixes & {"alert_2wks", "lag", "penalty", "obs"}, reward_mode=reward_mode)
        nd = np.unique(np.asarray(ct.n_days))
        self.uniform = len(nd) == 1 and nd[0] > 0
        self.uni_nd = int(nd[0]) if self.uniform else -1
        self.lockstep = bool(self.uniform and lockstep is not False)
        self.V.reset(*(np.zeros(self.n, np.int64) for _ in range(5)))  # five separate arrays: they are written in place
        self.V.n_days = np.ones(self.n, np.int64)          # k_init_state: n_days 1, finished
        self.sticky = np.full(self.n, -1, np.int64)
        self.episode_no = np.full(self.n, -1, np.int64)
        self.finished = np.ones(self.n, bool)
        self.ret32 = np.zeros(self.n, np.float32)          # the handle's running episode return (f32 accumulation)
        self.ret_tol = np.zeros(self.n, np.float64)        # slack on it where per-day rewards were not visible
        self.final_return = np.zeros(self.n, np.float32)
        self.final_tol = np.zeros(self.n, np.float64)
        self.obs = np.zeros((self.n, ct.n_obs), np.float32)
        self.pending_reset = False
        self.reset_cfg = None
        self.was_reset = False
        # ---- mirror of the handle's bookkeeping (csrc/w2a_bookkeeping.h)
        self.known_day = -1        # W2A_Q_LOCKSTEP_DAY
        self.lock = False          # W2A_Q_LOCKSTEP
        self.graph_canon = self.graph_packed = self.graph_autoreset = False
        self.pk_valid, self.canon_valid, self.poisoned = False, True, False
        self.order_set = False
        self.rm_valid = False
        self.py_order_stale = True
        self.last_step_kernel = -1
        self.last_rollout_kernel = -1

    # ------------------------------------------------------------------------------------------ helpers
    @property
    def packed_current(self) -> bool:
        """HeatAlertVecEnv.packed_state: the mirror is current and the canonical words are not."""
        return self.pk_valid and not self.canon_valid

    @property
    def any_graph(self) -> bool:
        return self.graph_canon or self.graph_packed

    def _can_pack(self) -> bool:
        return self.uni_nd > 0 and self.lock and not self.graph_canon  # (budgets do not matter: the kernel serves any)

    def _ensure_canonical(self):
        self.canon_valid = True

    def _canonical_modified(self, keeps_lock=True):
        self.pk_valid = False
        if not keeps_lock:
            self.lock, self.known_day = False, -1

    def _end_call(self):
        """bk_end_call: after a recorded packed step the mirror stays the primary form."""
        if not self.graph_packed:
            return
        if self.pk_valid:
            self.canon_valid = False
        elif self._can_pack():
            self.pk_valid, self.canon_valid, self.poisoned = True, False, False
        else:
            self.poisoned = True

    def _mode(self) -> str:
        auto = self.autoreset in ("same_step", "next_step")
        if not auto:
            return "none"
        if self.lockstep:
            return "host_auto" if self.autoreset == "same_step" else "host_next"
        return "dev_same" if self.autoreset == "same_step" else "dev_next"

    def _vsnap(self):
        return {k: getattr(self.V, k).copy() for k in _V_FIELDS}

    def _vrestore(self, snap, mask):
        if mask.any():
            for k in _V_FIELDS:
                getattr(self.V, k)[mask] = snap[k][mask]

    def _assign(self, sel, cw, yi, cc, sm, b):
        V = self.V
        V.county_w[sel], V.year_i[sel], V.coef_col[sel], V.sample[sel], V.budget[sel] = cw, yi, cc, sm, b
        V.n_days[sel] = V.n_days_tab[cw, yi]
        for k in ("t", "used", "streak", "last_actual"):
            getattr(V, k)[sel] = 0
        V.hist[sel] = 0
        V.at_budget[sel] = False
        rows = V._get_obs()
        V.obs[sel] = rows[sel]
        self.finished[sel] = False
        self.ret32[sel] = 0.0
        self.ret_tol[sel] = 0.0
        if self.write_obs:
            self.obs[sel] = rows[sel].astype(np.float32)

    def _note_launch_reset(self, masked):
        """launch_reset for from_tuples != 2 (bk_reset + bk_end_call)."""
        self.rm_valid = False
        if masked:
            self._ensure_canonical()
        self.canon_valid = True
        self._canonical_modified(keeps_lock=False)
        if not masked and self.uni_nd > 0:
            self.lock = True
            if not self.any_graph:
                self.known_day = 0
        self._end_call()

    def _device_reset(self, sel, restart, masked=False):
        """w2a_reset_device_rng on the selected envs with self.reset_cfg (+ the relabelling sort in sorted mode).
        masked: a mask was PASSED (whatever it selects: the handle cannot see its contents)."""
        idx = np.nonzero(sel)[0]
        ep = np.zeros(len(idx), np.int64) if restart else self.episode_no[idx] + 1
        cw, yi, cc, sm, b, so = draw_episodes(self.ct, self.reset_cfg, self.gid0 + idx, ep, self.sticky[idx],
                                              "augment" in self.fixes)
        self.episode_no[idx] = ep
        self.sticky[idx] = so
        self._assign(idx, cw, yi, cc, sm, b)
        self._note_launch_reset(masked=masked)
        if self.episode_order == "sorted":
            assert sel.all()
            self._relabel()
        self.py_order_stale = True
        self.pending_reset = False

    def _relabel(self):
        """w2a_sort_episodes: stable sort of the whole per-env record by coefficient row (column, draw)."""
        V = self.V
        key = (V.coef_col.astype(np.uint64) << np.uint64(12)) | V.sample.astype(np.uint64)
        p = np.argsort(key, kind="stable")
        for k in ("county_w", "year_i", "coef_col", "sample", "budget", "n_days") + _V_FIELDS:
            setattr(V, k, getattr(V, k)[p])
        for k in ("sticky", "episode_no", "finished", "ret32", "ret_tol"):
            setattr(self, k, getattr(self, k)[p])
        if self.write_obs:
            self.obs = V.obs.astype(np.float32)  # w2a_observe re-emits every first observation
        self.rm_valid = False  # w2a_sort_episodes; ensure_canonical + canonical_modified(keeps lock step)
        self._ensure_canonical()
        self._canonical_modified()
        self._end_call()
        if self.write_obs:  # w2a_observe
            self._ensure_canonical()
            self._end_call()

    # ------------------------------------------------------------------------------------------ reset
    def device_cfg(self, seed, options):
        """HeatAlertVecEnv._device_cfg."""
        o = options or {}
        loc = o.get("location")
        loc_i = -1 if loc is None else self.ct.fips_list.index(loc)
        aug = o.get("similar_climate_counties")
        aug = self.augment if aug is None else bool(aug)
        bk = self.ctor_budget if self.ctor_budget is not None else o.get("budget")
        mode = BUDGET_FIXED
        if o.get("sample_budget"):
            mode = {"less_than": BUDGET_LESS_THAN, "centered": BUDGET_CENTERED}[o.get("sample_budget_type") or "less_than"]
        return int(seed) & _M64, loc_i, int(aug), -1 if bk is None else int(bk), mode, int("budget" not in self.fixes)

    def _leave_lockstep(self):
        self.lockstep = False
        self.pending_reset = False  # _set_step_mode: not host_next any more

    def reset_device(self, seed, options=None, mask=None):
        """env.reset(seed=seed, options={..., "mask": mask}) in device seed mode."""
        sel = np.ones(self.n, bool) if mask is None else np.asarray(mask, bool)
        if self.lockstep and mask is not None:
            self._leave_lockstep()
        self.reset_cfg = self.device_cfg(seed, options)
        self._device_reset(sel, restart=True, masked=mask is not None)
        self.was_reset = True

    def reset_tuples(self, seed, ep, mask=None, options=None):
        """env.reset(seed=seed, options={"episodes": ep, "mask": mask, ...}): caller-chosen tuples (w2a_reset)."""
        sel = np.ones(self.n, bool) if mask is None else np.asarray(mask, bool)
        if self.lockstep:
            self._leave_lockstep()
        idx = np.nonzero(sel)[0]
        arr = {k: np.broadcast_to(np.asarray(ep[k], np.int64), (self.n,)) for k in ("county_w", "year_i", "coef_col", "sample")}
        if ep.get("budget") is None:
            bud = np.asarray(self.ct.B0)[arr["county_w"] * self.ct.Y + arr["year_i"]].astype(np.int64)
        else:
            bud = np.broadcast_to(np.asarray(ep["budget"], np.int64), (self.n,))
        self.episode_no[idx] += 1  # cold.w + 1; the sticky budget stays
        self._assign(idx, arr["county_w"][idx], arr["year_i"][idx], arr["coef_col"][idx], arr["sample"][idx], bud[idx])
        self._note_launch_reset(masked=mask is not None)
        self.py_order_stale = True
        if self.autoreset in ("same_step", "next_step"):
            self.reset_cfg = self.device_cfg(seed, options)
        self.was_reset = True

    # ------------------------------------------------------------------------------------------ step
    def _wide(self) -> bool:
        return (self.pm or self.step_kernel == "wide" or (self.step_kernel in ("auto", "unpacked") and self.n >= S64_MIN_ENVS)) \
            and self.step_kernel != "classic"

    def expect_step_kernel(self, flags_autoreset: bool, capturing: bool = False) -> int:
        """bk_step's plan: 0 k_step, 1 k_step64 on the canonical words, 2 k_step64 on the mirror; -1 refused (a capture
        that would have to record a conversion of the state's form)."""
        packed = self._wide() and not self.pm and self.step_kernel != "unpacked" and self._can_pack()  # autoreset or not
        if capturing:
            if packed and not self.pk_valid:
                packed = False
            if not packed and not self.canon_valid:
                return -1
        return 2 if packed else (1 if self._wide() else 0)

    def _note_step(self, flags_autoreset: bool, capturing: bool = False) -> int:
        k = self.expect_step_kernel(flags_autoreset, capturing)
        if k < 0:
            return k
        if flags_autoreset:
            self.rm_valid = False
        if capturing:
            if k == 2:
                self.graph_packed = True
            else:
                self.graph_canon = True
            if flags_autoreset:
                self.graph_autoreset = True
        nxt = self.known_day + 1 if (not flags_autoreset and self.known_day >= 0 and self.known_day + 1 < self.uni_nd) else -1
        if self.any_graph:
            nxt = -1
        if k == 2:
            self.pk_valid, self.poisoned = True, False
            self.canon_valid = False
        else:
            self._ensure_canonical()
            self._canonical_modified()
        self.known_day = nxt
        self.last_step_kernel = k
        if k == 2 or not capturing:
            self._end_call()
        return k

    def step(self, actions, dev_reward=None, note=True) -> dict:
        """env.step(actions). dev_reward: the f32 rewards the device returned for this call (each compared with the
        expected one by the caller): the handle adds those into its f32 episode return, so with them the model tracks
        the return to the last ulp or two (the compiler contracts `ret + c * base * (1 - eff)` into an FMA, so the
        kernel's sum is not always the sum of the ROUNDED reward it stored: 2 ulp of slack per step); without them
        (graph replays) the per-step reward tolerance accumulates instead. note=False: a REPLAYED step -- the device
        advances, the handle's bookkeeping does not run."""
        n, V = self.n, self.V
        a = np.asarray(actions).astype(np.int64).copy()
        if self.pending_reset:  # host-driven next_step: this call restarts the whole batch, nothing is stepped
            self._device_reset(np.ones(n, bool), restart=False)
            return {"obs": self.obs.copy(), "reward": np.zeros(n), "done": np.zeros(n, bool),
                    "final_return": self.final_return.copy(), "status": 0, "stepped": np.zeros(n, bool)}
        mode = self._mode()
        restart_in = self.finished & (mode == "dev_next")
        stepped = ~restart_in
        status = 0
        bad = stepped & (a != 0) & (a != 1)
        if bad.any():
            status |= 2
            a[bad] = 1
        a[restart_in] = 0
        if (self.finished & stepped).any():
            status |= 4
        snap = self._vsnap()
        _, r, done, _ = V.step(a)
        self._vrestore(snap, restart_in)
        done = done & stepped
        r = np.where(stepped, r, 0.0)
        add = np.asarray(dev_reward, np.float32) if dev_reward is not None else r.astype(np.float32)
        self.ret32 = np.where(stepped, (self.ret32 + add).astype(np.float32), self.ret32)
        per_step = (1e-5 + 2e-6 * np.abs(self.ret32)) if dev_reward is None else 2.4e-7 * np.maximum(np.abs(self.ret32), 1.0)
        self.ret_tol = self.ret_tol + np.where(stepped, per_step, 0.0)
        self.final_return = np.where(done, self.ret32, self.final_return)
        self.final_tol = np.where(done, self.ret_tol, self.final_tol)
        self.finished = np.where(stepped, done, self.finished)
        if self.write_obs:
            # the caller's buffer: a stepped env's row is written unless its terminal step has just run (the stale row
            # stays, Q6; W2A_FIX_OBS writes the last row) -- what "stale" is depends on the buffer, e.g. rollouts advance
            # days without writing rows
            written = stepped & (~done | ("obs" in self.fixes))
            self.obs[written] = V.obs[written].astype(np.float32)
        if note:
            self._note_step(mode in ("dev_same", "dev_next"))
        rs = done if mode == "dev_same" else (restart_in if mode == "dev_next" else np.zeros(n, bool))
        if rs.any():
            idx = np.nonzero(rs)[0]
            ep = self.episode_no[idx] + 1
            cw, yi, cc, sm, b, so = draw_episodes(self.ct, self.reset_cfg, self.gid0 + idx, ep, self.sticky[idx],
                                                  "augment" in self.fixes)
            self.episode_no[idx], self.sticky[idx] = ep, so
            self._assign(idx, cw, yi, cc, sm, b)
        if mode in ("host_auto", "host_next"):
            assert done.all() or not done.any(), "model: a lock-step batch whose envs do not finish together"
            if done.all():
                if mode == "host_auto":
                    self._device_reset(np.ones(n, bool), restart=False)
                else:
                    self.pending_reset = True
        return {"obs": self.obs.copy(), "reward": r, "done": done, "final_return": self.final_return.copy(),
                "status": status, "stepped": stepped}

    # ------------------------------------------------------------------------------------------ rollout
    def rollout(self, policy: dict, n_steps=None) -> dict:
        """env.rollout(policy, n_steps, alert_mask=True): per env up to n_steps days or to the end of its episode."""
        n, V, ct = self.n, self.V, self.ct
        if self.pending_reset:
            self._device_reset(np.ones(n, bool), restart=False)
        steps = int(n_steps) if n_steps is not None else ct.T
        left = np.where(self.finished, 0, V.n_days - V.t)
        if self.pm:
            steps = min(steps, int(left.max()))
        first_day = V.t.copy()
        pol = dict(policy)
        if pol["kind"] == "threshold":
            pol["col"] = ct.columns.index(pol["feature"])
        if pol["kind"] == "table":
            pol["table"] = np.asarray(pol["table"])

        class _Draws:
            def __init__(s2, outer):
                s2.o = outer

            def vec(s2, days):
                return O.devrng_policy_uniform_vec(int(pol.get("seed", 0)), s2.o.gid0 + np.arange(n), s2.o.episode_no, days)

        draws = _Draws(self)
        ret = np.zeros(n)
        alerts, over = np.zeros(n, np.int64), np.zeros(n, np.int64)
        days = np.zeros((n, ct.T), bool)
        att = np.zeros((n, ct.T), bool)
        snapv = np.full(n, np.nan)
        run = np.zeros(n, np.int64)
        fin_now = np.zeros(n, bool)
        ret_run = self.ret32.astype(np.float64)
        for _ in range(steps):
            live = ~self.finished
            if not live.any():
                break
            act = O._policy_actions(V, pol, draws)
            act = np.where(live, act, 0)
            tday = V.t.copy()
            atb = V.used == V.budget
            snap = self._vsnap()
            _, r, done, actual = V.step(act)
            self._vrestore(snap, ~live)
            ret += np.where(live, r, 0.0)
            ret_run += np.where(live, r, 0.0)
            alerts += np.where(live, actual, 0)
            over += np.where(live & (act == 1) & atb, 1, 0)
            li = np.nonzero(live & (actual == 1))[0]
            days[li, tday[li]] = True
            ai = np.nonzero(live & (act == 1))[0]
            att[ai, tday[ai]] = True
            run += live
            newly = live & done
            hit = live & ~newly & (V.t == V.n_days - 2) & np.isnan(snapv)
            snapv = np.where(hit, ret_run, snapv)
            self.finished = self.finished | newly
            fin_now |= newly
        tol = run * 1e-5 + 2e-6 * np.abs(ret_run) + 1e-6
        self.ret_tol = self.ret_tol + np.where(run > 0, tol, 0.0)
        self.ret32 = np.where(run > 0, ret_run.astype(np.float32), self.ret32)
        self.final_return = np.where(fin_now, self.ret32, self.final_return)
        self.final_tol = np.where(fin_now, self.ret_tol, self.final_tol)
        out = {"return": ret, "alerts": alerts, "attempts_over_budget": over, "alert_days": days, "attempt_days": att,
               "done": self.finished.copy(), "final_return": self.final_return.copy(), "finished_now": fin_now,
               "return_snapshot": snapv, "first_day": first_day, "days_run": run, "steps": steps,
               "tol": tol, "n_days": V.n_days.copy(), "budget": V.budget.copy()}
        # ---- what the handle knows afterwards / which kernel ran
        if not self.pm:
            if self.py_order_stale and self.rollout_order:
                self.order_set, self.rm_valid, self.py_order_stale = True, False, False
                if self.rollout_mfma and not self.fixbits:
                    self.rm_valid = True
            if self.graph_autoreset:
                self.rm_valid = False
            self.last_rollout_kernel = 2 if (self.rm_valid and self.order_set and not self.fixbits and
                                             self.lock) else (1 if self.order_set else 0)
            self._note_rollout_begin(steps)
            self._end_call()
        else:
            one_launch = self.pm_kernel != "matrix" and steps > 0
            if one_launch:
                self._note_rollout_begin(steps)
                self._end_call()
            else:
                for _ in range(steps):  # w2a_policy_actions, w2a_posterior_mean_reward (reads), w2a_step(REWARD_GIVEN)
                    self._ensure_canonical()
                    self._end_call()
                    self._note_step(False)
        mode = self._mode()
        if mode in ("host_auto", "host_next") and self.finished.all():
            if mode == "host_auto":
                self._device_reset(np.ones(n, bool), restart=False)
            else:
                self.pending_reset = True
        return out

    def _note_rollout_begin(self, steps):
        """bk_rollout_begin."""
        self._ensure_canonical()
        self._canonical_modified()
        self.known_day = self.known_day + steps if (self.known_day >= 0 and self.known_day + steps < self.uni_nd and
                                                    not self.any_graph) else -1

    # ------------------------------------------------------------------------------------------ read-backs
    def state(self) -> dict:
        V = self.V
        hist14 = np.zeros(self.n, np.int64)
        for k in range(14):
            hist14 |= V.hist[:, 13 - k].astype(np.int64) << k
        self._ensure_canonical()  # w2a_get_state
        self._end_call()
        return {"t": V.t, "used": V.used, "streak": V.streak, "hist14": hist14, "last_actual": V.last_actual,
                "at_budget": V.at_budget.astype(np.int64), "budget": V.budget, "n_days": V.n_days,
                "county_w": V.county_w, "year_i": V.year_i, "coef_col": V.coef_col, "sample": V.sample,
                "sticky_budget": self.sticky, "episode_no": self.episode_no, "finished": self.finished.astype(np.int64),
                "episode_return": self.ret32}

    def sync_returns(self, dev_return, dev_final=None):
        """After a tolerance comparison: continue from the handle's own f32 values (bit-exact tracking resumes)."""
        self.ret32 = np.asarray(dev_return, np.float32).copy()
        self.ret_tol[:] = 0.0
        if dev_final is not None:
            self.final_return = np.asarray(dev_final, np.float32).copy()
            self.final_tol[:] = 0.0

    def lockstep_truth(self):
        """The day every env is on if the batch really is in lock step (same day, nobody finished, one episode
        length), else -1: what a non-negative W2A_Q_LOCKSTEP_DAY must equal."""
        V = self.V
        if self.finished.any() or len(np.unique(V.t)) != 1 or len(np.unique(V.n_days)) != 1:
            return -1
        return int(V.t[0])

    def uniform_truth(self) -> bool:
        """Every env on the same day of an episode of the same length, all finished or none: what W2A_Q_LOCKSTEP = 1 claims."""
        V = self.V
        return len(np.unique(V.t)) == 1 and len(np.unique(V.n_days)) == 1 and len(np.unique(self.finished)) == 1

    # ------------------------------------------------------------------------------------------ checkpoints etc.
    _CKPT = ("V", "sticky", "episode_no", "finished", "ret32", "ret_tol", "final_return", "final_tol", "obs",
             "pending_reset", "reset_cfg", "lockstep", "was_reset")

    def snapshot(self) -> dict:
        self._ensure_canonical()  # state_dict() starts with state()
        self._end_call()
        d = {}
        for k in self._CKPT:
            v = getattr(self, k)
            d[k] = {f: getattr(v, f).copy() for f in ("county_w", "year_i", "coef_col", "sample", "budget", "n_days") + _V_FIELDS} \
                if k == "V" else copy.deepcopy(v)
        return d

    def note_invalidate(self):
        self.known_day, self.lock = -1, False
        self.rm_valid = False
        self.pk_valid, self.canon_valid = False, True
        self.poisoned = False  # the restored buffer covers the mirror's day words: the handle poisons them again if it must
        self._end_call()

    def restore(self, d: dict):
        """env.load_state_dict(...): the arrays come back; the handle forgets what it knew (w2a_invalidate) and gets the
        checkpoint's autoreset parameters again."""
        for k in self._CKPT:
            if k == "V":
                for f, v in d["V"].items():
                    setattr(self.V, f, v.copy())
            else:
                setattr(self, k, copy.deepcopy(d[k]))
        if not (self.lockstep and self.autoreset == "next_step"):
            self.pending_reset = False
        self.note_invalidate()
        self.py_order_stale = True
