"""Model-based call-sequence fuzz of the libw2a handle: 1000 random sequences of 30-80 operations each
-- resets (device RNG / injected tuples, masked / unmasked), steps in every kernel form and autoreset mode, partial and
whole rollouts, state(), checkpoints, w2a_invalidate, episode_order="sorted", the posterior-mean
reward with each kernel, hipGraph captures (of canonical and of packed steps) with replays right away and later in the
sequence -- mirrored on oracle/sequence_model.HandleModel; outputs compared
after every operation, w2a_query against what the sequence implies (tools/sequence_fuzz.py is the long form).
The reference allows reset/step to interleave arbitrarily (env.py:133-184,238-262); round 3 added seven validity flags
to the handle whose interleavings had one scripted test."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

N_SEQUENCES = 1000  # ~0.1 s each on MI355X (even ones read the state after every operation)
MASTER_SEED = 2024


def _load():
    spec = importlib.util.spec_from_file_location("sequence_fuzz", os.path.join(os.path.dirname(__file__), "..", "tools",
                                                                               "sequence_fuzz.py"))
    sf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sf)
    return sf


@pytest.mark.timeout(900)
def test_call_sequence_fuzz_against_the_model():
    assert torch.cuda.is_available()
    sf = _load()
    dev = torch.device("cuda:0")
    tot: dict = {}
    for i in range(N_SEQUENCES):
        s = sf.run_sequence(i, MASTER_SEED, dev)  # raises SequenceFailure with the operation log on a violation
        for k, v in s.items():
            tot[k] = max(tot.get(k, 0.0), v) if k == "worst" else tot.get(k, 0) + v
    print(f"sequence fuzz: {N_SEQUENCES} sequences, {tot}")
    assert tot["worst"] <= 1e-5
    # the sweep must actually have visited what it is for
    assert tot["ops"] >= 60 * N_SEQUENCES and tot["steps"] > 40 * N_SEQUENCES and tot["resets"] > 8 * N_SEQUENCES
    assert tot["rollouts"] > 4 * N_SEQUENCES and tot["packed_steps"] > N_SEQUENCES // 2 and tot["mfma_rollouts"] > N_SEQUENCES // 4
    assert tot["graphs"] > N_SEQUENCES // 4 and tot["ckpt"] > N_SEQUENCES and tot["after_done"] > 5 * N_SEQUENCES
    assert tot["autoresets"] > 100 * N_SEQUENCES
    # round 5: recorded graphs are replayed again later in the sequence, after whatever else happened to the handle; some
    # of them hold PACKED steps, and some replays find their mirror poisoned (and must then do nothing but say so)
    assert tot["replays"] > N_SEQUENCES and tot["packed_graphs"] >= 5 and tot["stale_replays"] >= 3, tot


def test_policy_loop_over_a_ragged_batch_raises_no_status_bit():
    """Finding 1 of the fuzz (round 4, sequence 82 of seed 2024). reward_mode='posterior_mean' with the fp64 matrix kernel
    has no one-launch rollout, so rollout() runs the per-day calls w2a_policy_actions / w2a_posterior_mean_reward /
    w2a_step(REWARD_GIVEN | SKIP_FINISHED). After a masked reset the envs finish on different days; the step skips the
    finished ones silently, but the reward pre-pass (k_pm_prep) raised W2A_ST_STEP_AFTER_DONE for them: the status word
    came back 4 after a rollout in which nothing was wrong."""
    import numpy as np

    from weather2alert_amd import HeatAlertVecEnv, synth, tables

    sd = synth.make_synth("linear", n_fips=12, years=[2006, 2007, 2008], n_samples=6, n_days=16, seed=0, extra_confounder_fips=2)
    ct = tables.compile_from_synth(sd)
    n = 127
    for pmk in ("matrix", "vector", "matrix_i8"):
        env = HeatAlertVecEnv(n, tables=ct, device="cuda:0", autoreset="disabled", reward_mode="posterior_mean", pm_kernel=pmk,
                              step_kernel="wide")
        env.pm_rollout_kernel = pmk == "matrix_i8"  # the two others through the per-day calls
        env.reset(seed=1)
        for _ in range(9):
            env.step(torch.ones(n, dtype=torch.int32, device="cuda:0"))
        env.reset(seed=2, options={"mask": np.arange(n) % 11 == 0})
        out = env.rollout(dict(kind="bernoulli", p=0.3, seed=5), alert_mask=True)
        assert bool(out["done"].all()) and env.check_status() == 0, pmk
        env.close()


def test_restored_sticky_budget_above_16_bits_is_served_by_the_packed_kernel():
    """Finding 5 of the fuzz (round 4, sequence 77 of seed 99): episodes with a sticky budget of 65 721, a checkpoint
    restore (w2a_invalidate) after which nothing was stated about budgets, a reset with injected tuples whose small
    budgets the caller then stated -- the handle took that as covering the whole buffer, and the next sticky device
    reset handed 65 721 to the 16-bit packed form: rewards off by 5. Rounds 4 and 5 kept such budgets away from the packed
    form (a scan of the restored buffer, a bound, statements); since round 6 the packed kernel reads budgets its 16-bit
    field cannot hold from the canonical words, nothing is stated or scanned, and the same sequence runs PACKED and right."""
    import numpy as np

    from oracle import heatalert_oracle as O
    from weather2alert_amd import HeatAlertVecEnv, _ffi, synth, tables

    sd = synth.make_synth("linear", n_fips=18, years=[2006, 2007, 2008], n_samples=6, n_days=5, seed=0, extra_confounder_fips=2)
    ct = tables.compile_from_synth(sd)
    n, dev = 64, torch.device("cuda:0")
    env = HeatAlertVecEnv(n, tables=ct, device=dev, autoreset="disabled", step_kernel="wide")
    env.reset(seed=1, options={"budget": 65721})  # sticky from here on (Q9)
    env.step(torch.ones(n, dtype=torch.int32, device=dev))
    assert env.packed_state
    env.state()
    _ffi.check(env._lib.w2a_invalidate(env._h, env._stream()), "w2a_invalidate")  # as after a restore
    assert env._lib.w2a_query(env._h, _ffi.Q_PACKED_ELIGIBLE) == 1  # a property of the tables, not of budgets
    rng = np.random.default_rng(0)
    county = rng.integers(0, ct.S, n)
    env.reset(seed=2, options={"episodes": dict(county_w=np.asarray(ct.fips_to_weather)[county], year_i=rng.integers(0, ct.Y, n),
                                                coef_col=county, sample=rng.integers(0, ct.n_samples, n),
                                                budget=rng.integers(0, 9, n))})
    env.reset(seed=3)  # device RNG, sticky: every env gets 65 721 again
    st = {k: v.cpu().numpy() for k, v in env.state().items()}
    assert (st["budget"] == 65721).all()
    V = O.VectorOracle(O.RefData.from_synth(sd), sd.fips_weather, sd.years)
    V.reset(st["county_w"], st["year_i"], st["coef_col"], st["sample"], st["budget"])
    for t in range(3):
        a = (rng.random(n) < 0.5).astype(np.int32)
        o, r, _, _, _ = env.step(torch.as_tensor(a, device=dev))
        o_o, r_o, _, _ = V.step(a)
        assert np.abs(r.cpu().numpy() - r_o).max() <= 1e-5 and env.packed_state
        assert np.array_equal(o.cpu().numpy(), o_o.astype(np.float32)), t
    assert env.check_status() == 0
    env.close()


def test_dropin_env_random_call_sequences():
    """The num_envs = 1 drop-in (`HeatAlertEnv`, NumPy-seed parity) under random interleavings of reset(**kwargs) and
    step() against the scalar `OracleEnv` -- the line-by-line restatement of env.py:107-262 that is pinned bit-exact to
    the reference's own trajectories: resets in the middle of an episode, steps after `done` (the reference keeps
    recomputing the last day, env.py:256-262), every reset kwarg incl. seed=None (the global NumPy RNG, env.py:143-144),
    the sticky budget through sample_budget of both types (Q9), augmentation (Q8), ragged episode lengths. The golden
    tests replay reset + 153 steps; this is what the reference's API allows beyond that."""
    import numpy as np

    from oracle import heatalert_oracle as O
    from weather2alert_amd import HeatAlertEnv, synth, tables
    from weather2alert_amd.tables import DeviceTables

    dev = torch.device("cuda:0")
    worst, n_ops, n_after_done = 0.0, 0, 0
    for variant in range(3):
        sd = synth.make_synth("linear", n_fips=10 + variant, years=[2006, 2007, 2008][: 1 + variant], n_samples=4,
                              n_days=[7, 12, 30][variant], seed=40 + variant, extra_confounder_fips=2)
        if variant == 1:
            sd.meta["n_days_per_episode"] = np.random.default_rng(5).integers(5, 13, size=(len(sd.fips_weather), len(sd.years)))
        rd = O.RefData.from_synth(sd)
        dt = DeviceTables(tables.compile_from_synth(sd), dev)
        for seq in range(25):
            rng = np.random.default_rng([variant, seq])
            ctor = dict(similar_climate_counties=bool(rng.random() < 0.4),
                        budget=None if rng.random() < 0.6 else int(rng.integers(0, 6)))
            env, orc = HeatAlertEnv(weights="linear", tables=dt, device="cuda:0", **ctor), O.OracleEnv(rd, **ctor)
            log = [f"variant {variant} sequence {seq} ctor {ctor}"]
            fresh = True
            for _ in range(int(rng.integers(40, 90))):
                if fresh or rng.random() < 0.12:
                    kw = dict(location=None if rng.random() < 0.5 else str(rng.choice(sd.fips_list)),
                              similar_climate_counties=[None, True, False][int(rng.integers(0, 3))],
                              seed=None if rng.random() < 0.2 else int(rng.integers(0, 10000)),
                              budget=None if rng.random() < 0.5 else int(rng.integers(0, 8)),
                              sample_budget=bool(rng.random() < 0.35),
                              sample_budget_type=str(rng.choice(["less_than", "centered"])))
                    log.append(f"reset({kw})")
                    g = int(rng.integers(0, 1 << 30))  # seed=None draws from the GLOBAL generator: same state for both
                    np.random.seed(g)
                    o1, i1 = env.reset(**kw)
                    np.random.seed(g)
                    o2, i2 = orc.reset(**kw)
                    r1 = r2 = None
                    d1 = d2 = False
                    fresh = False
                else:
                    a = int(rng.random() < 0.45)
                    n_after_done += orc.t >= orc.n_days - 1 and len(orc.actual_alert_buffer) >= orc.n_days
                    o1, r1, d1, tr, i1 = env.step(a)
                    o2, r2, d2, _, i2 = orc.step(a)
                    log.append(f"step({a}) -> done {d2}, t {orc.t}")
                    assert tr is False
                n_ops += 1
                ctx = "\n".join(log[:1] + log[-12:])
                assert np.array_equal(o1, o2.astype(np.float32)), ctx
                if r1 is not None:
                    assert abs(r1 - r2) <= 1e-5, (ctx, r1, r2)
                    worst = max(worst, abs(r1 - r2))
                assert d1 == d2, ctx
                for k in ("episode_index", "remaining_budget", "at_budget", "location", "location_index", "feature_names"):
                    assert i1[k] == i2[k], (ctx, k, i1[k], i2[k])
                assert (env.t, env.alert_streak, env.budget, env.coef_index, env.n_days, env.remaining_budget, env.at_budget) == \
                    (orc.t, orc.alert_streak, orc.budget, orc.coef_index, orc.n_days, orc.remaining_budget, orc.at_budget), ctx
            env.close()
    print(f"drop-in sequences: {n_ops} operations, {n_after_done} steps on finished episodes, max |reward - oracle| {worst:.2e}")
    assert n_ops > 3000 and n_after_done > 100 and worst <= 1e-5


def test_vector_env_numpy_parity_sequences():
    """seed_mode="numpy_parity" (the host replays NumPy's Generator per env, env.py:143-178) under random sequences
    against one scalar OracleEnv per env: per-env reset kwargs, masked resets, the per-env sticky budget (Q9), and the
    host-driven same_step autoreset, which re-seeds finished envs from the GLOBAL NumPy generator in env order
    (env.py:143-144) -- replayed on the oracle side by re-seeding that generator identically."""
    import numpy as np

    from oracle import heatalert_oracle as O
    from weather2alert_amd import HeatAlertVecEnv, synth, tables
    from weather2alert_amd.tables import DeviceTables

    dev = torch.device("cuda:0")
    sd = synth.make_synth("linear", n_fips=12, years=[2006, 2007], n_samples=3, n_days=6, seed=61, extra_confounder_fips=2)
    rd = O.RefData.from_synth(sd)
    dt = DeviceTables(tables.compile_from_synth(sd), dev)
    n_ops = n_auto = 0
    for seq in range(30):
        rng = np.random.default_rng([11, seq])
        n = int(rng.integers(1, 7))
        autoreset = str(rng.choice(["same_step", "disabled"]))
        aug, cb = bool(rng.random() < 0.4), (None if rng.random() < 0.6 else int(rng.integers(0, 5)))
        env = HeatAlertVecEnv(n, tables=dt, device=dev, seed_mode="numpy_parity", autoreset=autoreset,
                              similar_climate_counties=aug, budget=cb)
        orcs = [O.OracleEnv(rd, similar_climate_counties=aug, budget=cb) for _ in range(n)]
        last = {}
        first = True
        for _ in range(int(rng.integers(30, 70))):
            if first or rng.random() < 0.15:
                mask = None if (first or rng.random() < 0.5) else (rng.random(n) < 0.6)
                opts = dict(location=[None if rng.random() < 0.5 else str(rng.choice(sd.fips_list)) for _ in range(n)],
                            budget=None if rng.random() < 0.5 else int(rng.integers(0, 7)),
                            sample_budget=bool(rng.random() < 0.3),
                            sample_budget_type=str(rng.choice(["less_than", "centered"])))
                seeds = [int(x) for x in rng.integers(0, 10000, n)]
                o = dict(opts)
                if mask is not None:
                    o["mask"] = mask
                obs, info = env.reset(seed=seeds, options=o)
                last = {k: v for k, v in opts.items() if k != "location"}
                last["location"] = opts["location"]
                for i in range(n):
                    if mask is None or mask[i]:
                        orcs[i].reset(location=opts["location"][i], seed=seeds[i], budget=opts["budget"],
                                      sample_budget=opts["sample_budget"], sample_budget_type=opts["sample_budget_type"])
                first = False
                want_r = want_d = None
            else:
                a = (rng.random(n) < 0.5).astype(np.int32)
                g = int(rng.integers(0, 1 << 30))
                np.random.seed(g)
                obs, r, d, _, info = env.step(torch.as_tensor(a, device=dev))
                np.random.seed(g)
                res = [orcs[i].step(int(a[i])) for i in range(n)]
                want_r = np.asarray([x[1] for x in res])
                want_d = np.asarray([x[2] for x in res])
                if autoreset == "same_step":
                    for i in range(n):  # DummyVecEnv-style: finished envs restart with a fresh global-RNG seed, in env order
                        if want_d[i]:
                            orcs[i].reset(location=last["location"][i], budget=last["budget"], sample_budget=last["sample_budget"],
                                          sample_budget_type=last["sample_budget_type"])
                            n_auto += 1
                assert np.abs(r.cpu().numpy() - want_r).max() <= 1e-5 and np.array_equal(d.cpu().numpy(), want_d)
            n_ops += 1
            want_obs = np.stack([x.observation for x in orcs]).astype(np.float32)
            assert np.array_equal(obs.cpu().numpy(), want_obs), (seq, n_ops)
            st = {k: v.cpu().numpy() for k, v in env.state().items()}
            assert list(st["t"]) == [x.t for x in orcs] and list(st["streak"]) == [x.alert_streak for x in orcs]
            assert list(st["budget"]) == [x.budget for x in orcs] and list(st["sample"]) == [x.coef_index for x in orcs]
            assert list(info["location"]) == [x.location for x in orcs]
            assert list(info["episode_index"]) == [x.ep_index for x in orcs]
            assert list(info["remaining_budget"].cpu().numpy()) == [x.budget - sum(x.actual_alert_buffer) for x in orcs]
        env.close()
    print(f"numpy_parity sequences: {n_ops} operations, {n_auto} host autoresets")
    assert n_ops > 1000 and n_auto > 100
