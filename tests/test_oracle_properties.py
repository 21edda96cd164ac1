"""Property tests (hypothesis) of the budget / history state machine on the CPU oracle -- the invariants the
GPU suite then checks at full batch size (tests/test_env_gpu.py::test_budget_invariants_at_scale)."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import heatalert_oracle as O
from weather2alert_amd import synth


@pytest.fixture(scope="module")
def small():
    sd = synth.make_synth("linear", n_fips=6, years=[2006, 2007], n_samples=3, seed=5)
    rd = O.RefData.from_synth(sd)
    return sd, rd


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 10_000), budget=st.integers(0, 20), p=st.floats(0.0, 1.0), aseed=st.integers(0, 2**31 - 1))
def test_scalar_env_invariants(small, seed, budget, p, aseed):
    sd, rd = small
    env = O.OracleEnv(rd)
    obs, info = env.reset(seed=seed, budget=budget)
    assert info["remaining_budget"] == budget and not info["at_budget"]
    acts = (np.random.default_rng(aseed).random(153) < p).astype(int)
    i_rem, i_lag, i_streak = (env.feat_names.index(k) for k in ("remaining_budget", "alert_lag1", "alert_streak"))
    prev_obs, streak = obs, 0
    for t, a in enumerate(acts):
        used_before = sum(env.actual_alert_buffer)
        obs, r, done, trunc, info = env.step(int(a))
        actual = env.actual_alert_buffer[-1]
        assert actual == (1 if (a == 1 and used_before < budget) else 0)          # budget gate (env.py:242-246)
        assert sum(env.actual_alert_buffer) <= budget
        assert info["remaining_budget"] == budget - sum(env.actual_alert_buffer)
        assert info["at_budget"] == (used_before == budget)
        assert -1000 / 152 - 1e-12 <= r <= 0 and r != -1                            # Q5: the penalty never fires
        assert done == (t == 152) and trunc is False
        if not done:
            assert obs[i_rem] == info["remaining_budget"]
            assert obs[i_lag] == (actual if t > 0 else 0)                            # Q3
            assert obs[i_streak] == streak                                           # Q4: streak before today's action
            assert obs[-1] == sum(env.actual_alert_buffer[-14:])                     # Q1 slot
            streak = streak + 1 if actual else 0
            assert env.alert_streak == streak and env.t == t + 1
        else:
            np.testing.assert_array_equal(obs, prev_obs)                             # Q6: stale terminal observation
        prev_obs = obs


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 40), p=st.floats(0.0, 0.6))
def test_vector_oracle_equals_scalar_oracle(small, seed, n, p):
    """The vectorised restatement and the line-by-line one agree bit for bit on random batches."""
    sd, rd = small
    rng = np.random.default_rng(seed)
    V = O.VectorOracle(rd, sd.fips_weather, sd.years)
    county = rng.integers(0, len(sd.fips_list), n)
    year_i = rng.integers(0, len(sd.years), n)
    sample = rng.integers(0, rd.n_samples, n)
    budget = rng.integers(0, 8, n)
    V.reset(county, year_i, county, sample, budget)
    envs = []
    for i in range(n):
        e = O.OracleEnv(rd, budget=int(budget[i]))
        # force the episode tuple instead of drawing it
        e.rng = np.random.default_rng(0)
        e.location, e.location_index = sd.fips_list[county[i]], int(county[i])
        e.ep = rd.episodes[(sd.fips_weather[county[i]], sd.years[year_i[i]])]
        e.n_days, e.coef_index = e.ep.shape[0], int(sample[i])
        e.actual_alert_buffer, e.attempted_alert_buffer, e.alert_streak, e.t = [], [], 0, 0
        e.remaining_budget, e.at_budget, e.ep_index = e.budget, False, "x"
        e.observation = e._get_obs()
        envs.append(e)
    for t in range(153):
        a = (rng.random(n) < p).astype(np.int64)
        obs, r, done, actual = V.step(a)
        for i, e in enumerate(envs):
            o, ri, di, _, _ = e.step(int(a[i]))
            assert ri == r[i] and di == done[i] and e.actual_alert_buffer[-1] == actual[i]
            np.testing.assert_array_equal(o, obs[i])
