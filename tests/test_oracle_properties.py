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


def test_episode_csv_fields_are_the_reference_callbacks():
    """The per-episode CSV (HeatAlertVecEnv.write_episode_csv, oracle CSV_FIELDS) has the field names and order of the
    reference's FinalEvalCallback (callbacks.py:136-146, written by csv.DictWriter at :151-157). The expected list
    is committed; when the reference tree is present (build container) it is also read from the source text."""
    import os
    import re

    from oracle import heatalert_oracle as O
    from weather2alert_amd.env import HeatAlertVecEnv

    expect = ["year", "alert_budget", "sum_alerts", "reward", "average_t_alerts", "stdev_t_alerts", "average_streak",
              "stdev_streak", "alerts"]
    assert list(O.CSV_FIELDS) == expect == list(HeatAlertVecEnv.CSV_FIELDS)
    src = "/root/reference/src/weather2alert/callbacks.py"
    if os.path.exists(src):
        text = open(src).read()
        block = text[text.index("self.data.append({"): text.index("# Reset:")]
        assert re.findall(r'"([a-z_0-9%]+)":', block) == expect
        logged = re.findall(r'"([a-z_0-9%]+)":', text[text.index("summary = {"): text.index("for k, v in summary.items()")])
        assert logged == ["training_rewards", "over_budget_freq", "alerts_freq", "average_t_alerts", "stdev_t_alerts",
                          "average_streak", "stdev_streak", "alert_t_50%", "alert_t_80%", "alert_t_100%"]
        v, log = O._EnvView(n_days=4, year=2006, budget=1), O.AlertLoggingOracle()
        v.after_step(1, 1, False, -1.0, 1)
        log.on_step([v])
        assert list(log.on_rollout_end()) == logged


def test_callback_restatement_on_a_hand_checked_episode():
    """AlertLoggingOracle / FinalEvalOracle on a 7-day episode worked out by hand from callbacks.py."""
    from oracle import heatalert_oracle as O

    v = O._EnvView(n_days=7, year=2010, budget=2)
    log, fin = O.AlertLoggingOracle(), O.FinalEvalOracle()
    attempted = [1, 1, 1, 0, 1, 0, 0]
    used = 0
    for k, a in enumerate(attempted):
        atb = used == 2
        actual = 0 if (a and atb) else a
        used += actual
        v.after_step(a, actual, atb, -1.0, min(k + 1, 6))
        fin.on_step(v)
        log.on_step([v])
    s = log.on_rollout_end()
    # attempted alerts on days 0,1,2,4 -> env.t after the step = 1,2,3,5; streaks ended: 3 (days 0-2) and 1 (day 4)
    assert s["average_t_alerts"] == 2.75 and s["average_streak"] == 2.0 and s["stdev_streak"] == 1.0
    assert s["alerts_freq"] == 4 / 7 and s["over_budget_freq"] == 2 / 7  # days 2 and 4 were attempted at budget
    # read at t == 5 (after day 4): granted list [1,1,0,0,0] -> 50 % at index 0, 80 % and 100 % at index 1
    assert (s["alert_t_50%"], s["alert_t_80%"], s["alert_t_100%"]) == (0.0, 1.0, 1.0) and s["training_rewards"] == -5.0
    r = fin.row()
    assert r["year"] == 2010 and r["alert_budget"] == 2 and r["sum_alerts"] == 2 and r["reward"] == -5.0
    assert r["alerts"] == [1, 1, 0, 0, 0, 0, 0] and r["average_t_alerts"] == 1.5 and r["average_streak"] == 2.0
