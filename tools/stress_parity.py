#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random table shapes, batch sizes around every tile boundary, random policies and
budgets -- every step kernel form (4-lanes-per-env, 64-envs-per-wave on the canonical and on the packed state), both
sampled-reward rollout kernels (k_rollout64, k_rollout_mfma) and the three posterior-mean kernels against the float64
oracle; every other case also the in-kernel autoreset with random corrected-semantics flags, 64-envs-per-wave kernel
against 4-lanes-per-env kernel (oracle/heatalert_oracle.py; test infrastructure, used here as the checker only).

    python tools/stress_parity.py [--cases 40] [--seed 0]

Bars: integers / observations / done exact, reward <= 1e-5, rollout returns <= 2e-6 relative + 2e-5 absolute. Prints one line per case
and a summary; exits non-zero on the first violation (with the case's parameters, so that it can be replayed)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from oracle import heatalert_oracle as O  # noqa: E402
from weather2alert_amd import HeatAlertVecEnv, synth, tables  # noqa: E402

EDGE_N = [1, 2, 15, 16, 17, 63, 64, 65, 127, 255, 256, 257, 1023, 1025, 4097]
# returns summed over up to 153 days in f32 inside a kernel against the float64 oracle: the north star's 1e-5 is a per-step
# reward bound; a sum of n rewards may differ by n x 1e-5 at most (1.5e-3 per episode). Measured: <= 5.3e-7 relative
# (returns of magnitude 10^2..10^3), so the suite holds the kernels to 2e-6 relative + 2e-5 absolute
RETURN_RTOL, RETURN_ATOL = 2e-6, 2e-5


def oracle_for(env, V):
    st = {k: v.cpu().numpy() for k, v in env.state().items()}
    V.reset(st["county_w"], st["year_i"], st["coef_col"], st["sample"], st["budget"])
    V._finished = np.zeros(len(st["t"]), bool)
    return st


def run_case(i, rng, dev):
    n_fips = int(rng.integers(3, 40))
    years = list(range(2006, 2006 + int(rng.integers(1, 4))))
    n_samples = int(rng.integers(1, 14))
    n_days = int(rng.choice([153, 153, 153, 40, 97]))
    sd = synth.make_synth("linear", n_fips=n_fips, years=years, n_samples=n_samples, n_days=n_days, seed=int(rng.integers(1 << 30)),
                          extra_confounder_fips=int(rng.integers(0, 4)))
    ct = tables.compile_from_synth(sd)
    n = int(rng.choice(EDGE_N)) if rng.random() < 0.6 else int(rng.integers(1, 6000))
    augment = bool(rng.random() < 0.5)
    gid0 = int(rng.integers(0, 1 << 20))
    seed = int(rng.integers(1 << 30))
    budget = None if rng.random() < 0.5 else int(rng.integers(0, 25))
    kernel = str(rng.choice(["classic", "wide", "unpacked"]))
    p_act = float(rng.choice([0.05, 0.2, 0.6, 1.0]))
    desc = (f"case {i}: S={ct.S} Y={ct.Y} T={ct.T} draws={n_samples} n={n} augment={augment} budget={budget} "
            f"kernel={kernel} p={p_act} gid0={gid0} seed={seed}")
    V = O.VectorOracle(O.RefData.from_synth(sd), sd.fips_weather, sd.years)
    kw = dict(tables=ct, device=dev, env_gid0=gid0, similar_climate_counties=augment, autoreset="disabled")
    opts = None if budget is None else {"budget": budget}
    worst = 0.0
    # ---- step(): one kernel form against the oracle, a whole episode
    env = HeatAlertVecEnv(n, step_kernel="auto" if kernel == "unpacked" else kernel, **kw)
    if kernel == "unpacked":
        env._step_flags |= 128  # W2A_STEP_UNPACKED: the 64-envs-per-wave kernel on the canonical words
        env._step_flags |= 32   # W2A_STEP_WIDE
    obs, _ = env.reset(seed=seed, options=opts)
    oracle_for(env, V)
    steps = int(rng.integers(1, ct.T + 1))
    for t in range(steps):
        a = (rng.random(n) < p_act).astype(np.int32)
        obs, r, done, _, _ = env.step(torch.as_tensor(a, device=dev))
        obs_o, r_o, done_o, _ = V.step(a)
        err = float(np.abs(r.cpu().numpy().astype(np.float64) - r_o).max())
        worst = max(worst, err)
        assert err <= 1e-5, (desc, "reward", t, err)
        assert np.array_equal(done.cpu().numpy(), done_o), (desc, "done", t)
        assert np.array_equal(obs.cpu().numpy(), obs_o.astype(np.float32)), (desc, "obs", t)
    st = env.state()
    for k, ref in (("used", V.used), ("streak", V.streak), ("t", V.t)):
        assert np.array_equal(st[k].cpu().numpy(), ref), (desc, k)
    # ---- the rest of the episode by rollout() on a restored checkpoint (the handle has lost its lock-step knowledge:
    # k_rollout64 on both settings), same policy
    kindp = str(rng.choice(["always", "bernoulli", "threshold", "threshold_lag0", "table", "never"]))
    table = (rng.random((ct.T, 6)) < 0.3).astype(np.uint8)
    pol = {"always": dict(kind="always"), "never": dict(kind="never"), "bernoulli": dict(kind="bernoulli", p=0.2, seed=seed & 0xFFFF),
           "threshold": dict(kind="threshold", feature="heat_qi", threshold=0.75, require_budget=bool(rng.random() < 0.5)),
           "threshold_lag0": dict(kind="threshold", feature="heat_qi", threshold=0.6, lag=0),
           "table": dict(kind="table", table=table)}[kindp]
    if steps < ct.T:
        sdict = env.state_dict()
        outs = {}
        for mfma in (True, False):
            e2 = HeatAlertVecEnv(n, rollout_mfma=mfma, **kw)
            e2.reset(seed=seed, options=opts)
            e2.load_state_dict(sdict)
            outs[mfma] = e2.rollout(pol, alert_mask=True)
            outs[mfma]["kernel"] = e2.last_rollout_kernel
            e2.close()
        epno = st["episode_no"].cpu().numpy()
        draw = (lambda j, t: O.devrng_policy_uniform(seed & 0xFFFF, gid0 + j, int(epno[j]), t)) if kindp == "bernoulli" else None
        V._finished = np.zeros(n, bool)
        ret_o, al_o, ov_o, days_o = O.oracle_rollout(V, dict(pol, col=ct.columns.index("heat_qi")), ct.T, draw)
        for mfma, out in outs.items():
            tag = (desc, kindp, out["kernel"])
            assert np.array_equal(out["alerts"].cpu().numpy(), al_o), tag + ("alerts",)
            assert np.array_equal(out["attempts_over_budget"].cpu().numpy(), ov_o), tag + ("over",)
            assert np.array_equal(out["alert_days"].cpu().numpy(), days_o), tag + ("days",)
            np.testing.assert_allclose(out["return"].cpu().numpy(), ret_o, rtol=RETURN_RTOL, atol=RETURN_ATOL, err_msg=str(tag))
            assert bool(out["done"].all()), tag + ("done",)
    env.close()
    # ---- a fresh lock-step batch through the matrix-core rollout (the restored one above has lost lock-step knowledge)
    e3 = HeatAlertVecEnv(n, rollout_mfma=True, **kw)
    e3.reset(seed=seed + 1, options=opts)
    st3 = oracle_for(e3, V)
    k = int(rng.integers(0, 20))
    for t in range(min(k, ct.T - 1)):
        a = (rng.random(n) < p_act).astype(np.int32)
        e3.step(torch.as_tensor(a, device=dev))
        V.step(a)
    out = e3.rollout(pol, alert_mask=True)
    used_mfma = e3.last_rollout_kernel
    epno = st3["episode_no"]
    draw = (lambda j, t: O.devrng_policy_uniform(seed & 0xFFFF, gid0 + j, int(epno[j]), t)) if kindp == "bernoulli" else None
    ret_o, al_o, ov_o, days_o = O.oracle_rollout(V, dict(pol, col=ct.columns.index("heat_qi")), ct.T, draw)
    tag = (desc, kindp, used_mfma, "fresh")
    assert np.array_equal(out["alerts"].cpu().numpy(), al_o), tag + ("alerts",)
    assert np.array_equal(out["alert_days"].cpu().numpy(), days_o), tag + ("days",)
    np.testing.assert_allclose(out["return"].cpu().numpy(), ret_o, rtol=RETURN_RTOL, atol=RETURN_ATOL, err_msg=str(tag))
    assert e3.check_status() == 0, tag
    e3.close()
    # ---- posterior-mean reward, one of the three kernels, a few days
    pmk = str(rng.choice(["vector", "matrix", "matrix_i8"]))
    Vp = O.VectorOracle(O.RefData.from_synth(sd), sd.fips_weather, sd.years, reward_mode="posterior_mean")
    e4 = HeatAlertVecEnv(n, reward_mode="posterior_mean", pm_kernel=pmk, **kw)
    e4.reset(seed=seed + 2, options=opts)
    oracle_for(e4, Vp)
    worst_pm = 0.0
    for t in range(int(rng.integers(1, 8))):
        a = (rng.random(n) < p_act).astype(np.int32)
        _, r, _, _, _ = e4.step(torch.as_tensor(a, device=dev))
        _, r_o, _, _ = Vp.step(a)
        worst_pm = max(worst_pm, float(np.abs(r.cpu().numpy().astype(np.float64) - r_o).max()))
    assert worst_pm <= 1e-5, (desc, "posterior_mean", pmk, worst_pm)
    e4.close()
    # ---- in-kernel same-step autoreset (+ a random set of corrected-semantics flags): the 64-envs-per-wave kernel's
    # epilogue / FIXES variants against the 4-lanes-per-env kernel's, ragged episode lengths, envs restarting on their own
    if i % 2 == 0:
        sd2 = synth.make_synth("linear", n_fips=n_fips, years=years, n_samples=n_samples, n_days=n_days, seed=seed & 0xFFFF,
                               extra_confounder_fips=2)
        sd2.meta["n_days_per_episode"] = rng.integers(max(n_days - 30, 2), n_days + 1, size=(n_fips, len(years)))
        ct2 = tables.compile_from_synth(sd2)
        fx = tuple(f for f in ("alert_2wks", "lag", "penalty", "obs", "augment") if rng.random() < 0.3)
        mode = str(rng.choice(["same_step", "next_step"]))
        kw2 = dict(tables=ct2, device=dev, env_gid0=gid0, similar_climate_counties=augment or "augment" in fx, lockstep=False, fixes=fx,
                   autoreset=mode)
        ea, eb = HeatAlertVecEnv(n, step_kernel="wide", **kw2), HeatAlertVecEnv(n, step_kernel="classic", **kw2)
        oa, _ = ea.reset(seed=seed)
        ob, _ = eb.reset(seed=seed)
        assert torch.equal(oa, ob), (desc, "autoreset reset obs", fx)
        for t in range(int(rng.integers(n_days, 2 * n_days + 20))):
            at = torch.as_tensor((rng.random(n) < p_act).astype(np.int32), device=dev)
            oa, ra, da, _, _ = ea.step(at)
            ob, rb, db, _, _ = eb.step(at)
            assert torch.equal(oa, ob) and torch.equal(da, db), (desc, "autoreset obs/done", mode, fx, t)
            assert torch.allclose(ra, rb, rtol=0, atol=1e-6), (desc, "autoreset reward", fx, t)
        sa, sb = ea.state(), eb.state()
        for k in sa:
            if k != "episode_return":
                assert torch.equal(sa[k], sb[k]), (desc, "autoreset state", k, fx)
        assert ea.check_status() == eb.check_status() == 0, (desc, fx)
        ea.close()
        eb.close()
    print(f"{desc} steps={steps} policy={kindp} rollout={used_mfma} pm={pmk}: reward {worst:.2e}, posterior mean {worst_pm:.2e}",
          flush=True)
    return worst, worst_pm, used_mfma


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    w = wp = 0.0
    kinds = {}
    for i in range(a.cases):
        x, y, k = run_case(i, rng, dev)
        w, wp = max(w, x), max(wp, y)
        kinds[k] = kinds.get(k, 0) + 1
    print(f"stress_parity: {a.cases} cases OK in {time.time() - t0:.0f} s; max |reward - oracle| {w:.2e}, posterior mean {wp:.2e}; "
          f"fresh-batch rollouts by kernel: {kinds}")


if __name__ == "__main__":
    main()
