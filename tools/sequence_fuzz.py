#!/usr/bin/env python3
"""Model-based call-sequence fuzz of the libw2a handle (through HeatAlertVecEnv, i.e. through the C ABI) on the GPU.

    python tools/sequence_fuzz.py [--sequences 400] [--seed 0] [--only I] [--verbose]

Per sequence: a random small table set (lock-step or ragged episode lengths, 3 .. 153 days), a random env configuration
(autoreset same_step / next_step / disabled, step kernel classic / wide / unpacked / auto, lockstep on / off, corrected-
semantics flags, sampled or posterior-mean reward with each of its kernels, episode_order iid / sorted, rollout order and
matrix cores on / off, observations on / off) and 30-80 random operations:

    unmasked / masked reset() with the device RNG (location, augmentation, budget, both sample_budget types),
    unmasked / masked reset() with injected tuples (budgets in device memory),
    step() singly and in bursts (now and then with an action outside {0, 1}; stepping finished envs is part of it),
    partial and whole rollout() with every policy kind, state(), state_dict() -> load_state_dict(),
    w2a_invalidate, a hipGraph capture of a block of steps + replays.

Every operation is mirrored on oracle/sequence_model.HandleModel (float64 VectorOracle arithmetic, the restated device
RNG) and after EVERY operation the output buffers are compared: observations bit-exact, rewards <= 1e-5, done / final
returns, the status word -- plus w2a_query against what the sequence implies (lock-step day, which step / rollout kernel
ran, packed form current), and the soundness of a non-negative lock-step day against the model's per-env truth. The full
decoded state is compared when the sequence itself reads it (state() is an operation: reading it after every call
would bring the canonical words up to date each time and hide exactly the stale-flag bugs this is after) and at the
end of each sequence. oracle/ is test infrastructure: used here as the checker only.

Exit code 1 and the operation log of the failing sequence (replay: --only I) on the first violation."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle.sequence_model import HandleModel  # noqa: E402
from weather2alert_amd import HeatAlertVecEnv, _ffi, synth, tables  # noqa: E402

REWARD_TOL = 1e-5
EDGE_N = [1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513]
_TABLE_CACHE: dict = {}


class SequenceFailure(AssertionError):
    pass


def table_set(key):
    """(SynthData, CompiledTables) for key = (n_fips, n_years, n_samples, n_days, ragged, seed); cached."""
    if key not in _TABLE_CACHE:
        n_fips, n_years, n_samples, n_days, ragged, seed = key
        sd = synth.make_synth("linear", n_fips=n_fips, years=list(range(2006, 2006 + n_years)), n_samples=n_samples,
                              n_days=n_days, seed=seed, extra_confounder_fips=2)
        if ragged:
            r = np.random.default_rng(seed + 1)
            sd.meta["n_days_per_episode"] = r.integers(max(2, n_days - 6), n_days + 1, size=(n_fips, n_years))
        _TABLE_CACHE[key] = (sd, tables.compile_from_synth(sd))
    return _TABLE_CACHE[key]


BIG = False  # --big: every sequence at >= 131 072 envs (the size from which step_kernel="auto" goes wide by itself)


def draw_config(rng):
    n_days = int(rng.choice([3, 4, 5, 6, 8, 11, 16, 24, 40, 153], p=[.08, .1, .12, .14, .14, .14, .1, .08, .06, .04]))
    ragged = bool(rng.random() < 0.25)
    key = (int(rng.integers(3, 20)), int(rng.integers(1, 4)), int(rng.integers(1, 8)), n_days, ragged, int(rng.integers(0, 6)))
    u = rng.random()
    n = int(rng.choice(EDGE_N)) if u < 0.6 else int(rng.integers(1, 400)) if u < 0.95 else int(rng.integers(1025, 6000))
    if (u > 0.995 or BIG) and n_days <= 16:
        n = 131072 + int(rng.integers(0, 300))  # step_kernel="auto" picks the 64-envs-per-wave kernel by itself from here
    pm = bool(rng.random() < 0.2)
    autoreset = str(rng.choice(["same_step", "next_step", "disabled"], p=[.45, .3, .25]))
    lockstep = None if rng.random() < 0.7 else False
    if ragged:
        lockstep = None
    if pm and (ragged or lockstep is False) and autoreset != "disabled":
        autoreset = "disabled"  # posterior_mean cannot run with the in-kernel autoreset
    fixes = ()
    if not pm and rng.random() < 0.3:
        fixes = tuple(f for f in ("alert_2wks", "lag", "penalty", "obs", "augment", "budget") if rng.random() < 0.35)
    sk = str(rng.choice(["auto", "classic", "wide", "unpacked"], p=[.15, .25, .45, .15]))
    if pm and sk == "classic":
        sk = "wide"
    order = "iid"
    if not ragged and lockstep is None and rng.random() < 0.15:
        order = "sorted"
    cfg = dict(n=n, autoreset=autoreset, lockstep=lockstep, fixes=fixes, step_kernel=sk,
               reward_mode="posterior_mean" if pm else "sampled", episode_order=order,
               pm_kernel=str(rng.choice(["vector", "matrix", "matrix_i8"])),
               augment=bool(rng.random() < 0.5) or "augment" in fixes,
               ctor_budget=None if rng.random() < 0.75 else int(rng.integers(0, 7)) if (pm or rng.random() < 0.85) else 66000,
               gid0=int(rng.integers(0, 1 << 20)), write_obs=bool(rng.random() < 0.92),
               rollout_order=bool(rng.random() < 0.8), rollout_mfma=bool(rng.random() < 0.8))
    return key, cfg


class Runner:
    def __init__(self, i, rng, dev, verbose=False):
        self.i, self.rng, self.dev, self.verbose = i, rng, dev, verbose
        self.key, c = draw_config(rng)
        self.cfg = c
        sd, ct = table_set(self.key)
        self.ct = ct
        n = c["n"]
        self.log = [f"sequence {i}: tables (n_fips, years, draws, n_days, ragged, seed) = {self.key}; {c}"]
        self.env = HeatAlertVecEnv(n, tables=ct, device=dev, env_gid0=c["gid0"], similar_climate_counties=c["augment"],
                                   autoreset=c["autoreset"], lockstep=c["lockstep"], fixes=c["fixes"] or None,
                                   step_kernel=c["step_kernel"], reward_mode=c["reward_mode"], pm_kernel=c["pm_kernel"],
                                   episode_order=c["episode_order"], budget=c["ctor_budget"], write_obs=c["write_obs"],
                                   rollout_order=c["rollout_order"], rollout_mfma=c["rollout_mfma"])
        self.m = HandleModel(sd, ct, n, gid0=c["gid0"], fixes=c["fixes"], reward_mode=c["reward_mode"],
                             autoreset=c["autoreset"], augment=c["augment"], ctor_budget=c["ctor_budget"],
                             episode_order=c["episode_order"], lockstep=c["lockstep"], write_obs=c["write_obs"],
                             step_kernel=c["step_kernel"], rollout_order=c["rollout_order"], rollout_mfma=c["rollout_mfma"],
                             pm_kernel=c["pm_kernel"])
        self.n = n
        # even sequences read the full decoded state after EVERY operation (complete comparison, but each read brings
        # the canonical state words up to date); odd ones only where the sequence itself asks for it (a stale form
        # then has to show in what later operations compute)
        self.state_every_op = i % 2 == 0
        self.log[0] += f"; state after every op: {self.state_every_op}"
        self.ckpt = None
        self.graph = None
        self.stats = {"ops": 0, "steps": 0, "resets": 0, "rollouts": 0, "graphs": 0, "ckpt": 0, "worst": 0.0,
                      "packed_steps": 0, "mfma_rollouts": 0, "after_done": 0, "autoresets": 0, "packed_graphs": 0,
                      "replays": 0, "stale_replays": 0}
        self.expect_status = 0

    # ------------------------------------------------------------------ checking
    def fail(self, what):
        raise SequenceFailure("\n".join(self.log[:1] + self.log[1:][-60:]) + f"\n>>> {what}")

    def q(self, what):
        return self.env._lib.w2a_query(self.env._h, what)

    def check_flags(self, where):
        m, e = self.m, self.env
        d = self.q(_ffi.Q_LOCKSTEP_DAY)
        if d >= 0 and d != m.lockstep_truth():
            self.fail(f"{where}: handle claims lock-step day {d}, the envs are at {m.lockstep_truth()} (-1 = not in lock step)")
        if d != m.known_day:
            self.fail(f"{where}: W2A_Q_LOCKSTEP_DAY {d}, the sequence implies {m.known_day}")
        lk = self.q(_ffi.Q_LOCKSTEP)
        if lk and not m.uniform_truth():
            self.fail(f"{where}: handle claims lock step, the envs are not on one day")
        if lk != int(m.lock):
            self.fail(f"{where}: W2A_Q_LOCKSTEP {lk}, the sequence implies {int(m.lock)}")
        if e.packed_state != m.packed_current:
            self.fail(f"{where}: packed_state {e.packed_state}, the sequence implies {m.packed_current}")
        if self.q(_ffi.Q_LAST_STEP_KERNEL) != m.last_step_kernel:
            self.fail(f"{where}: last step kernel {self.q(_ffi.Q_LAST_STEP_KERNEL)}, the sequence implies {m.last_step_kernel}")
        if not m.pm and self.q(_ffi.Q_LAST_ROLLOUT_KERNEL) != m.last_rollout_kernel:
            self.fail(f"{where}: last rollout kernel {self.q(_ffi.Q_LAST_ROLLOUT_KERNEL)}, the sequence implies {m.last_rollout_kernel}")
        elig = self.q(_ffi.Q_PACKED_ELIGIBLE)
        if elig != int(m.uni_nd > 0):  # (the synthetic tables' dims always fit the mirror's bit fields)
            self.fail(f"{where}: W2A_Q_PACKED_ELIGIBLE {elig}, uniform length {m.uni_nd}")
        py = (e._lockstep, e._pending_reset)
        if py != (m.lockstep, m.pending_reset):
            self.fail(f"{where}: host (lockstep, pending_reset) = {py}, model {(m.lockstep, m.pending_reset)}")

    def check_obs(self, where):
        got = self.env._obs.cpu().numpy()
        if not np.array_equal(got, self.m.obs):
            bad = np.nonzero((got != self.m.obs).any(axis=1))[0]
            j = int(bad[0])
            cols = np.nonzero(got[j] != self.m.obs[j])[0]
            self.fail(f"{where}: observation rows differ for {len(bad)} envs, first env {j} columns {cols.tolist()} "
                      f"got {got[j, cols].tolist()} want {self.m.obs[j, cols].tolist()}")

    def check_status(self, where, want):
        try:
            bits = self.env.check_status()
        except ValueError:
            bits = 2 | (4 if (want & 4) else 0)  # raised for the bad action; other bits were cleared with it
            if not (want & 2):
                self.fail(f"{where}: BAD_ACTION raised, not expected")
            return
        except RuntimeError:  # W2A_ST_STALE_GRAPH: a replayed packed step found its mirror poisoned
            if not (want & 8) or (want & 2):
                self.fail(f"{where}: STALE_GRAPH raised, status word wanted {want}")
            return
        if bits != want:
            self.fail(f"{where}: status word {bits}, want {want}")

    def check_state(self, where):
        st = {k: v.cpu().numpy() for k, v in self.env.state().items()}
        ms = self.m.state()
        for k, want in ms.items():
            if k == "episode_return":
                err = np.abs(st[k].astype(np.float64) - want.astype(np.float64))
                if (err > self.m.ret_tol).any():
                    j = int(np.argmax(err - self.m.ret_tol))
                    self.fail(f"{where}: episode_return of env {j}: {st[k][j]!r} vs {want[j]!r} (tolerance {self.m.ret_tol[j]:.2e})")
            elif not np.array_equal(st[k], want):
                j = int(np.nonzero(st[k] != want)[0][0])
                self.fail(f"{where}: state[{k}] differs, first env {j}: got {st[k][j]} want {want[j]}")
        self.m.sync_returns(st["episode_return"])
        return st

    def check_final(self, where, mask=None):
        got = self.env._final_return.cpu().numpy()
        want, tol = self.m.final_return, self.m.final_tol
        sel = np.ones(self.n, bool) if mask is None else mask
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        if (err[sel] > tol[sel]).any():
            j = int(np.nonzero(sel & (err > tol))[0][0])
            self.fail(f"{where}: final_return of env {j}: {got[j]!r} vs {want[j]!r} (tolerance {tol[j]:.2e})")
        self.m.final_return = np.where(sel, got, self.m.final_return).astype(np.float32)
        self.m.final_tol = np.where(sel, 0.0, self.m.final_tol)

    # ------------------------------------------------------------------ operations
    def op_reset_device(self, masked):
        rng, ct = self.rng, self.ct
        opts = {}
        if rng.random() < 0.15:
            cand = [f for i, f in enumerate(ct.fips_list) if ct.fips_to_weather[i] >= 0 and ct.sim_cnt[i] > 0]
            opts["location"] = str(rng.choice(cand))
        if rng.random() < 0.25:
            opts["similar_climate_counties"] = bool(rng.random() < 0.5) or "augment" in self.cfg["fixes"]
        if rng.random() < 0.4:
            # now and then a budget the mirror's 16-bit field cannot hold (65535 is the escape value itself): the packed
            # kernel reads those from the canonical words
            opts["budget"] = int(rng.integers(0, 8)) if (self.m.pm or rng.random() < 0.75) else int(rng.integers(65530, 70000))
        if rng.random() < 0.3:
            opts["sample_budget"] = True
            opts["sample_budget_type"] = str(rng.choice(["less_than", "centered"]))
        seed = int(rng.integers(0, 1 << 40))
        mask = (rng.random(self.n) < rng.choice([0.1, 0.5, 0.9])) if masked else None
        self.log.append(f"reset(seed={seed}, options={opts}, mask={'%d of %d' % (int(mask.sum()), self.n) if masked else None})")
        o = dict(opts)
        if masked:
            o["mask"] = mask
        self.env.reset(seed=seed, options=o)
        self.m.reset_device(seed, opts, mask)
        self.stats["resets"] += 1

    def op_reset_tuples(self, masked):
        rng, ct, n = self.rng, self.ct, self.n
        county = rng.integers(0, ct.S, n)
        cw = np.asarray(ct.fips_to_weather)[county].astype(np.int64)
        yi = rng.integers(0, ct.Y, n)
        ep = dict(county_w=cw, year_i=yi, coef_col=rng.integers(0, ct.S, n), sample=rng.integers(0, ct.n_samples, n),
                  budget=None if rng.random() < 0.3 else rng.integers(0, 9 if (self.m.pm or rng.random() < 0.75) else 70000, n))
        seed = int(rng.integers(0, 1 << 40))
        mask = (rng.random(n) < rng.choice([0.1, 0.5, 0.9])) if masked else None
        self.log.append(f"reset(seed={seed}, episodes=<tuples, budget {'table' if ep['budget'] is None else 'array'}>, "
                        f"mask={'%d of %d' % (int(mask.sum()), n) if masked else None})")
        o = {"episodes": ep}
        if masked:
            o["mask"] = mask
        self.env.reset(seed=seed, options=o)
        self.m.reset_tuples(seed, ep, mask, {})
        self.stats["resets"] += 1

    def _actions(self):
        rng, n = self.rng, self.n
        a = (rng.random(n) < rng.choice([0.05, 0.3, 0.7, 1.0])).astype(np.int64)
        if rng.random() < 0.03:
            a[int(rng.integers(0, n))] = int(rng.choice([2, -1, 7]))
        dt = [torch.int32, torch.int64, torch.uint8][int(rng.integers(0, 3))]
        if dt == torch.uint8 and (a < 0).any():
            dt = torch.int32
        return a, torch.as_tensor(a, device=self.dev).to(dt)

    def op_step(self):
        a, at = self._actions()
        was_fin = int(self.m.finished.sum())
        obs, r, done, trunc, _ = self.env.step(at)
        rd = r.cpu().numpy()
        exp = self.m.step(a, rd)
        self.log.append(f"step(alerts {int((a == 1).sum())}/{self.n}, dtype {at.dtype}) -> done {int(exp['done'].sum())}, "
                        f"finished before {was_fin}, kernel {self.m.last_step_kernel}")
        where = self.log[-1]
        err = np.abs(rd.astype(np.float64) - exp["reward"])
        if err.max() > REWARD_TOL:
            j = int(err.argmax())
            self.fail(f"{where}: reward of env {j}: {rd[j]!r} vs {exp['reward'][j]!r} (|diff| {err[j]:.2e})")
        self.stats["worst"] = max(self.stats["worst"], float(err.max()))
        if not np.array_equal(done.cpu().numpy(), exp["done"]):
            self.fail(f"{where}: done differs: got {done.cpu().numpy().nonzero()[0][:8]} want {exp['done'].nonzero()[0][:8]}")
        if bool(trunc.any()):
            self.fail(f"{where}: truncated set")
        self.check_final(where)
        self.expect_status |= exp["status"]
        self.stats["steps"] += 1
        self.stats["packed_steps"] += self.m.last_step_kernel == 2
        self.stats["after_done"] += bool(exp["status"] & 4)
        self.stats["autoresets"] += int(exp["done"].sum()) if self.cfg["autoreset"] != "disabled" else 0

    def op_rollout(self, whole):
        rng, ct = self.rng, self.ct
        kind = str(rng.choice(["never", "always", "bernoulli", "threshold", "threshold_lag0", "table"]))
        table = (rng.random((ct.T, int(rng.integers(1, 7)))) < 0.4).astype(np.uint8)
        pol = {"always": dict(kind="always"), "never": dict(kind="never"),
               "bernoulli": dict(kind="bernoulli", p=float(rng.choice([0.1, 0.5])), seed=int(rng.integers(0, 1 << 30))),
               "threshold": dict(kind="threshold", feature="heat_qi", threshold=float(rng.choice([0.3, 0.75])),
                                 require_budget=bool(rng.random() < 0.5)),
               "threshold_lag0": dict(kind="threshold", feature="heat_qi", threshold=0.6, lag=0),
               "table": dict(kind="table", table=table)}[kind]
        n_steps = None if whole else int(rng.integers(1, max(2, ct.T)))
        masks = bool(rng.random() < 0.7)  # the kernels are compiled with and without the day bitmaps / snapshot
        out = self.env.rollout(pol, n_steps=n_steps, alert_mask=masks)
        exp = self.m.rollout(pol, n_steps)
        self.log.append(f"rollout({kind}, n_steps={n_steps}, alert_mask={masks}) -> days run max {int(exp['days_run'].max())}, "
                        f"finished now {int(exp['finished_now'].sum())}, kernel {self.m.last_rollout_kernel}")
        where = self.log[-1]
        g = {k: v.cpu().numpy() for k, v in out.items() if torch.is_tensor(v)}
        if not masks:
            for k in ("alert_days", "attempt_days", "first_day", "return_snapshot"):
                g[k] = np.asarray(exp[k])
        for k in ("alerts", "attempts_over_budget", "alert_days", "attempt_days", "done"):
            if not np.array_equal(g[k], exp[k]):
                j = int(np.nonzero((g[k] != exp[k]).reshape(self.n, -1).any(axis=1))[0][0])
                self.fail(f"{where}: rollout output {k} differs, first env {j}: got {g[k][j]} want {exp[k][j]}")
        err = np.abs(g["return"].astype(np.float64) - exp["return"])
        if (err > exp["tol"]).any():
            j = int(np.argmax(err - exp["tol"]))
            self.fail(f"{where}: rollout return of env {j}: {g['return'][j]!r} vs {exp['return'][j]!r}")
        if not np.array_equal(g["first_day"], exp["first_day"]):
            self.fail(f"{where}: first_day differs")
        sn, se = g["return_snapshot"].astype(np.float64), exp["return_snapshot"]
        if not np.array_equal(np.isnan(sn), np.isnan(se)):
            j = int(np.nonzero(np.isnan(sn) != np.isnan(se))[0][0])
            self.fail(f"{where}: return_snapshot presence differs at env {j}: got {sn[j]} want {se[j]}")
        ok = ~np.isnan(se)
        if ok.any() and (np.abs(sn[ok] - se[ok]) > (exp["tol"] + self.m.ret_tol + 1e-4)[ok]).any():
            self.fail(f"{where}: return_snapshot values differ")
        # final returns of the envs that finished inside this call (the clone rollout() took before a lock-step reset)
        fin = exp["finished_now"]
        if fin.any():
            fe = np.abs(g["final_return"].astype(np.float64) - exp["final_return"].astype(np.float64))
            if (fe[fin] > (self.m.final_tol + exp["tol"] + 1e-6)[fin]).any():
                j = int(np.nonzero(fin & (fe > self.m.final_tol + exp["tol"] + 1e-6))[0][0])
                self.fail(f"{where}: rollout final_return of env {j}: {g['final_return'][j]!r} vs {exp['final_return'][j]!r}")
        self.check_final(where)
        self.check_state(where + " [state after rollout]")  # rollout() has just read the state itself: no perturbation
        self.stats["rollouts"] += 1
        self.stats["mfma_rollouts"] += self.m.last_rollout_kernel == 2 and not self.m.pm

    def op_state(self):
        self.log.append("state()")
        self.check_state("state()")

    def op_checkpoint(self):
        self.log.append("state_dict()")
        self.ckpt = (self.env.state_dict(), self.m.snapshot())
        self.stats["ckpt"] += 1

    def op_restore(self):
        self.log.append("load_state_dict(<last checkpoint>)")
        self.env.load_state_dict(self.ckpt[0])
        self.m.restore(self.ckpt[1])

    def op_invalidate(self):
        self.log.append("state(); w2a_invalidate()")
        self.check_state("state() before w2a_invalidate")
        e = self.env
        _ffi.check(e._lib.w2a_invalidate(e._h, e._stream()), "w2a_invalidate")
        self.m.note_invalidate()
        e._regroup()  # posterior_mean: the column grouping was dropped with everything else
        self.m.py_order_stale = True

    def op_switch_kernel(self):
        """A/B switches a user may flip between calls: the step kernel form and the rollout options."""
        e, m, rng = self.env, self.m, self.rng
        sk = str(rng.choice(["auto", "wide", "unpacked"] + ([] if m.pm else ["classic"])))
        e.step_kernel = m.step_kernel = sk
        e._set_step_mode()
        e.rollout_order = m.rollout_order = bool(rng.random() < 0.8)
        e.rollout_mfma = m.rollout_mfma = bool(rng.random() < 0.8)
        self.log.append(f"switch: step_kernel={sk}, rollout_order={e.rollout_order}, rollout_mfma={e.rollout_mfma}")

    def op_c_level_empty_reset(self):
        """Straight at the C ABI, past the Python class: a masked device-RNG reset whose mask selects NO env. No env
        changes, but the handle cannot know that: it must drop what it knew (lock-step day, packed form, tile list) and
        -- posterior-mean mode -- refuse the reward kernel until the envs are grouped by column again (the flag whose
        staleness the round-2 advisor found by reading; the Python class regroups after every reset of its own, so only
        a direct call can see it)."""
        e, m = self.env, self.m
        with torch.cuda.device(self.dev):
            if e._sort_ws is not None and self.rng.random() < 0.5:
                # episode_order="sorted": the relabelling sort once more. The envs are already in key order and the sort
                # is stable, so nobody moves -- but the handle must again treat every index as holding another episode
                self.log.append("C level: w2a_sort_episodes (already sorted: the identity)")
                _ffi.check(e._lib.w2a_sort_episodes(e._h, e._sort_ws.data_ptr(), e._sort_ws.numel(), e._stream()),
                           "w2a_sort_episodes")
                m.rm_valid = False  # bk_sort + bk_end_call
                m._ensure_canonical()
                m._canonical_modified()
                m._end_call()
            else:
                self.log.append("C level: w2a_reset_device_rng(mask = nobody)")
                zero = torch.zeros(self.n, dtype=torch.uint8, device=self.dev)
                _ffi.check(e._lib.w2a_reset_device_rng(e._h, *e._reset_cfg, 1, zero.data_ptr(), e._obs_ptr, e._stream()),
                           "w2a_reset_device_rng")
                m._note_launch_reset(masked=True)
            torch.cuda.synchronize()
            if m.pm:
                act = torch.zeros(self.n, dtype=torch.int32, device=self.dev)
                rc = e._lib.w2a_posterior_mean_reward(e._h, act.data_ptr(), _ffi.ACT_I32, e._rew_ptr, e._stream())
                if rc == 0:
                    self.fail("C level: w2a_posterior_mean_reward ran on a column grouping that a reset had outdated")
        e._regroup()
        m.py_order_stale = True

    def op_graph(self):
        """A block of K step() calls captured into a hipGraph and replayed R times (autoreset in the kernel or disabled:
        the host must have nothing to do between the steps of a replayed block). Whatever form of the state the handle
        steps right now is the form the recorded kernels step -- the packed one too (round 5): the capture executes
        nothing and the handle keeps the recorded form current from then on, so the sequence simply goes on afterwards
        (more replays come as their own operation, after whatever else the sequence does to the handle)."""
        rng, n, e, m = self.rng, self.n, self.env, self.m
        K, R = int(rng.integers(1, 5)), int(rng.integers(1, 4))
        acts = [(rng.random(n) < 0.3).astype(np.int64) for _ in range(K)]
        at = [torch.as_tensor(a, device=self.dev).to(torch.int32) for a in acts]
        if (rng.random() < 0.5 or (m._can_pack() and m._wide())) and not m.pending_reset:
            self.op_step()  # a batch that can be packed enters the packed form here: the capture then records packed steps
            self.after_op()
        self.log.append(f"hipGraph: capture {K} steps, replay {R} times")
        where = self.log[-1]
        self.check_state(where + " [state() before capture]")  # either form may be recorded, but no conversion: both current now
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for t in at:
                e.step(t)
        # what the capture calls did to the handle's bookkeeping (no kernel ran): mirrored by stepping the model's flags
        kinds = [m._note_step(m._mode() in ("dev_same", "dev_next"), capturing=True) for _ in range(K)]
        if min(kinds) < 0 or len(set(kinds)) != 1:
            self.fail(f"{where}: the model expects the capture to be refused / to mix kernels: {kinds}")
        self.log[-1] += f" (recorded kernel {kinds[0]})"
        self.graph = dict(g=g, acts=acts, at=at, kind=kinds[0], mode=m._mode(), step_kernel=m.step_kernel, write_obs=m.write_obs,
                          reset_cfg=m.reset_cfg)
        self.check_flags(where + " [after capture]")
        self.stats["graphs"] += 1
        self.stats["packed_graphs"] += kinds[0] == 2
        for rep in range(R):
            self.op_replay()

    def op_replay(self):
        """One more replay of the recorded block -- after whatever the sequence did to the handle since the capture."""
        e, m, G = self.env, self.m, self.graph
        self.log.append(f"hipGraph: replay ({len(G['acts'])} recorded steps, kernel {G['kind']})")
        where = self.log[-1]
        G["g"].replay()
        torch.cuda.synchronize()
        if G["kind"] == 2 and m.poisoned:
            # the mirror could not be kept current: the replayed packed steps do nothing but raise W2A_ST_STALE_GRAPH
            self.log[-1] += " -> poisoned mirror: nothing stepped"
            self.expect_status |= 8
            self.stats["stale_replays"] += 1
            if self.state_every_op:
                return
            self.check_state(where + " [state after a refused replay]")
            return
        if G["kind"] == 2 and not m.pk_valid:
            self.fail(f"{where}: model: a recorded packed step on a mirror that is neither current nor poisoned")
        for a in G["acts"]:
            exp = m.step(a, None, note=False)
            self.expect_status |= exp["status"]
        rd = e._reward.cpu().numpy()
        err = np.abs(rd.astype(np.float64) - exp["reward"])
        if err.max() > REWARD_TOL:
            self.fail(f"{where}: reward after the last replayed step differs by {err.max():.2e}")
        if not np.array_equal(e._done_bool.cpu().numpy(), exp["done"]):
            self.fail(f"{where}: done after the last replayed step differs")
        self.check_final(where)
        self.stats["steps"] += len(G["acts"])
        self.stats["replays"] += 1

    def replay_possible(self) -> bool:
        G, m = self.graph, self.m
        # a recorded call keeps its ARGUMENTS: the step flags of the mode it was recorded in and -- with the in-kernel
        # autoreset -- the w2a_set_autoreset parameters of that moment (include/w2a.h); the model replays with today's
        return (G is not None and not m.pending_reset and m._mode() == G["mode"] and m.step_kernel == G["step_kernel"]
                and m.write_obs == G["write_obs"] and (G["mode"] == "none" or m.reset_cfg == G["reset_cfg"]))

    # ------------------------------------------------------------------ sequence
    def run(self):
        rng, m = self.rng, self.m
        self.op_reset_device(False)
        self.after_op()
        n_ops = int(rng.integers(30, 81))
        pm_auto = m.pm and self.cfg["autoreset"] != "disabled"
        sorted_mode = self.cfg["episode_order"] == "sorted"
        for _ in range(n_ops):
            partial_ok = not sorted_mode and not pm_auto  # masks / tuples: refused there by the env (by design)
            ops = [("step", 40), ("burst", 6), ("rollout_part", 6), ("rollout_whole", 3), ("reset", 5), ("state", 6),
                   ("ckpt", 3), ("status", 3), ("invalidate", 2), ("switch", 2)]
            if not m.pending_reset:
                ops.append(("c_reset", 2))
            if partial_ok:
                ops += [("reset_masked", 5), ("tuples", 3), ("tuples_masked", 3)]
            if self.ckpt is not None:
                ops.append(("restore", 3))
            if self.graph is None and not m.pm and m._mode() in ("dev_same", "dev_next", "none") and not m.pending_reset:
                ops.append(("graph", 3))
            if self.replay_possible():
                ops.append(("replay", 5))
            names, w = zip(*ops)
            op = str(rng.choice(names, p=np.asarray(w, float) / sum(w)))
            if op == "step":
                self.op_step()
            elif op == "burst":
                for _ in range(int(rng.integers(2, 12))):
                    self.op_step()
                    self.after_op()
            elif op == "rollout_part":
                self.op_rollout(False)
            elif op == "rollout_whole":
                self.op_rollout(True)
            elif op == "reset":
                self.op_reset_device(False)
            elif op == "reset_masked":
                self.op_reset_device(True)
            elif op == "tuples":
                self.op_reset_tuples(False)
            elif op == "tuples_masked":
                self.op_reset_tuples(True)
            elif op == "state":
                self.op_state()
            elif op == "ckpt":
                self.op_checkpoint()
            elif op == "restore":
                self.op_restore()
            elif op == "invalidate":
                self.op_invalidate()
            elif op == "graph":
                self.op_graph()
            elif op == "replay":
                self.op_replay()
            elif op == "switch":
                self.op_switch_kernel()
            elif op == "c_reset":
                self.op_c_level_empty_reset()
            elif op == "status":
                self.log.append("check_status()")
                self.check_status("check_status()", self.expect_status)
                self.expect_status = 0
            self.after_op()
        self.check_status("end of sequence", self.expect_status)
        self.check_state("end of sequence")
        self.env.close()
        return self.stats

    def after_op(self):
        self.stats["ops"] += 1
        where = self.log[-1]
        self.check_obs(where)
        self.check_flags(where)
        if self.state_every_op:
            self.check_state(where + " [state after the operation]")
            self.check_flags(where + " [after state()]")
        if self.verbose:
            print("   ", where, flush=True)


def run_sequence(i, master_seed, dev, verbose=False):
    rng = np.random.default_rng([master_seed, i])
    r = Runner(i, rng, dev, verbose)
    try:
        return r.run()
    except SequenceFailure:
        raise
    except Exception as e:  # noqa: BLE001  (an exception inside the env is a finding too: show the sequence)
        raise SequenceFailure("\n".join(r.log[:1] + r.log[1:][-60:]) + f"\n>>> exception {e!r}") from e
    finally:
        try:
            r.env.close()
        except Exception:  # noqa: BLE001
            pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--keep-going", type=int, default=0, help="report up to this many failing sequences instead of stopping at the first")
    ap.add_argument("--big", action="store_true", help="batches of >= 131 072 envs in every sequence with episodes <= 16 days")
    a = ap.parse_args()
    global BIG
    BIG = a.big
    dev = torch.device("cuda:0")
    t0 = time.time()
    tot: dict = {}
    todo = [a.only] if a.only is not None else range(a.sequences)
    failures = []
    for i in todo:
        try:
            s = run_sequence(i, a.seed, dev, a.verbose)
        except SequenceFailure as e:
            print(f"FAILED sequence {i} (replay: python tools/sequence_fuzz.py --seed {a.seed} --only {i} --verbose)\n{e}", flush=True)
            failures.append(i)
            if len(failures) > a.keep_going:
                return 1
            continue
        for k, v in s.items():
            tot[k] = max(tot.get(k, 0.0), v) if k == "worst" else tot.get(k, 0) + v
        if (i + 1) % 25 == 0:
            print(f"  {i + 1} sequences, {tot['ops']} operations, {time.time() - t0:.0f} s", flush=True)
    print(f"sequence_fuzz: {len(list(todo)) - len(failures)} of {len(list(todo))} sequences OK in {time.time() - t0:.0f} s "
          f"(seed {a.seed}): {tot}" + (f"; FAILED: {failures}" if failures else ""))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
