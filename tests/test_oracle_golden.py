"""Pin the CPU oracle against vectors captured from the unmodified reference env
(tests/golden/make_golden.py). Integer state must be bit-exact; the float64 rewards and
observations are required bit-exact too (same arithmetic, same summation order)."""
import json
import os

import numpy as np
import pytest

from oracle import heatalert_oracle as O


def _load(golden_dir, name):
    d = dict(np.load(os.path.join(golden_dir, name)))  # NpzFile re-reads on every access
    return d, json.loads(str(d["meta_json"]))


@pytest.fixture(scope="module")
def mini(golden_dir, mini_root):
    d, meta = _load(golden_dir, "mini_traj.npz")
    data = O.RefData.from_files(mini_root, weights="linear", split="65k")
    return d, meta, data


def test_loader_matches_reference_metadata(mini):
    d, meta, data = mini
    assert data.fips_list == meta["fips_list"]
    assert data.valid_years == meta["valid_years"]
    assert data.n_samples == meta["n_samples"]
    assert data.baseline_keys == meta["baseline_keys"]
    assert data.effectiveness_keys == meta["effectiveness_keys"]
    assert data.columns + ["alert_2wks"] == meta["feature_names"]
    assert data.sig_categories == meta["significance_categories"]
    assert len(meta["feature_names"]) == 29 and meta["declared_obs_shape"] == [33]  # Q12


def test_scalar_oracle_reproduces_every_golden_episode(mini):
    d, meta, data = mini
    envs = {}
    for i, e in enumerate(meta["episodes"]):
        key = e["env_key"]
        if key not in envs:
            envs[key] = O.OracleEnv(data, **e["ctor"])
        env = envs[key]
        obs, info = env.reset(**e["reset"])
        assert info["location"] == e["info_location"]
        assert info["episode_index"] == e["episode_index"]
        assert info["location_index"] == d["location_index"][i]
        assert env.coef_index == d["coef_index"][i]
        assert env.budget == d["budget"][i]
        assert info["remaining_budget"] == d["reset_remaining_budget"][i]
        assert env.n_days == d["n_days"][i] == 153  # Q7
        np.testing.assert_array_equal(obs, d["obs0"][i])
        for t, a in enumerate(d["actions"][i]):
            obs, r, done, trunc, info = env.step(int(a))
            assert env.actual_alert_buffer[-1] == d["actual"][i, t]
            assert r == d["reward"][i, t], (i, t, r, d["reward"][i, t])
            assert done == d["done"][i, t]
            np.testing.assert_array_equal(obs, d["obs"][i, t])
            assert info["remaining_budget"] == d["remaining_budget"][i, t]
            assert info["at_budget"] == d["at_budget"][i, t]
            assert env.alert_streak == d["streak_after"][i, t]
            assert env.t == d["t_after"][i, t]
        assert done and trunc is False


def test_quirks_visible_in_goldens(mini):
    d, meta, data = mini
    names = meta["feature_names"]
    ones = [i for i, e in enumerate(meta["episodes"]) if e["env_key"] == "ones"][0]
    b = d["budget"][ones]
    # Q5: over-budget attempts are silently dropped, never penalised with -1
    assert d["actual"][ones].sum() == b and (d["actual"][ones][:b] == 1).all()
    assert (d["reward"][ones] != -1).all()
    # Q3: alert_lag1 in the returned obs equals today's actual action for t>0
    lag = names.index("alert_lag1")
    np.testing.assert_array_equal(d["obs"][ones][1:-1, lag], d["actual"][ones][1:-1])
    # Q6: terminal step returns the previous observation
    np.testing.assert_array_equal(d["obs"][ones][-1], d["obs"][ones][-2])
    # Q1: the agent's 14-day count lives in the appended slot; 'alerts_2wks' stays historical
    assert names[-1] == "alert_2wks" and "alerts_2wks" in names[:-1]
    # Q9: budget sticks to the first episode's value on a long-lived env
    st = [i for i, e in enumerate(meta["episodes"]) if e["env_key"] == "sticky_plain"]
    assert len(set(d["budget"][st].tolist())) == 1
    kw = [i for i, e in enumerate(meta["episodes"]) if e["env_key"] == "sticky_kw"]
    assert set(d["budget"][kw].tolist()) == {4}


def test_vector_oracle_equals_goldens(mini):
    d, meta, data = mini
    fw = sorted({k[0] for k in data.episodes})
    V = O.VectorOracle(data, fw, data.valid_years)
    E = len(meta["episodes"])
    cw = [fw.index(e["episode_index"].split("_")[0]) for e in meta["episodes"]]
    yi = [data.valid_years.index(int(e["episode_index"].split("_")[1])) for e in meta["episodes"]]
    obs0 = V.reset(cw, yi, d["location_index"], d["coef_index"], d["budget"])
    np.testing.assert_array_equal(obs0, d["obs0"])
    for t in range(153):
        obs, r, done, actual = V.step(d["actions"][:, t])
        np.testing.assert_array_equal(actual, d["actual"][:, t])
        np.testing.assert_array_equal(r, d["reward"][:, t])
        np.testing.assert_array_equal(done, d["done"][:, t])
        np.testing.assert_array_equal(obs, d["obs"][:, t])
        np.testing.assert_array_equal(V.budget - V.used, d["remaining_budget"][:, t])
        np.testing.assert_array_equal(V.at_budget, d["at_budget"][:, t])
        np.testing.assert_array_equal(V.streak, d["streak_after"][:, t])
    assert E == 58


def test_numpy_parity_reset_tuples(mini):
    d, meta, data = mini
    sticky = {}
    for i, e in enumerate(meta["episodes"]):
        key, kw = e["env_key"], e["reset"]
        if key not in sticky:
            sticky[key] = e["ctor"].get("budget")
        aug = kw.get("similar_climate_counties", e["ctor"].get("similar_climate_counties", False))
        loc, li, year, ci, b, info_loc = O.numpy_parity_reset_tuple(
            data, kw["seed"], kw.get("location"), aug, sticky[key], kw.get("budget"),
            kw.get("sample_budget", False), kw.get("sample_budget_type", "less_than"))
        sticky[key] = b
        assert f"{loc}_{year}" == e["episode_index"]
        assert (li, ci, b, info_loc) == (d["location_index"][i], d["coef_index"][i], d["budget"][i],
                                         e["info_location"])


def test_full_size_anchors(golden_dir):
    """Known answers on the full 746-county weight list + the step arithmetic on
    self-contained episodes captured from the reference."""
    d, meta = _load(golden_dir, "full_anchor.npz")
    assert meta["n_fips"] == 746 and meta["index_06037"] == 84 and meta["n_similar_06037"] == 111
    got = [(e["reset"]["seed"], e["episode_index"], int(c), int(li))
           for e, c, li in zip(meta["episodes"], d["coef_index"], d["location_index"])]
    assert got[0] == (0, "06037_2015", 63, 84)
    assert got[1] == (1, "06037_2011", 51, 84)
    assert got[2] == (123, "06037_2006", 68, 84)
    assert got[3] == (5, "06037_2014", 2, 74)
    # pure NumPy replay of the draw order (env.py:145-160)
    for s, ep, c, li in got[:3]:
        r = np.random.default_rng(s)
        assert int(r.choice(meta["valid_years"])) == int(ep.split("_")[1]) and int(r.integers(0, 100)) == c
    r = np.random.default_rng(5)
    assert (int(r.choice(range(111))), int(r.choice(meta["valid_years"])), int(r.integers(0, 100))) == (74, 2014, 2)
    # step arithmetic on the captured episode tables/weights
    cols = meta["feature_names"][:-1]
    data = O.RefData()
    data.columns = cols
    data.baseline_keys, data.effectiveness_keys = meta["baseline_keys"], meta["effectiveness_keys"]
    data.n_samples, data.fips_list, data.valid_years = 1, ["x"], [0]
    for i in range(len(meta["episodes"])):
        data.episodes = {("x", 0): d["episode_table"][i]}
        data.wb = d["episode_weights"][i, 0].reshape(-1, 1, 1)
        data.we = d["episode_weights"][i, 1].reshape(-1, 1, 1)
        env = O.OracleEnv(data, budget=int(d["budget"][i]))
        obs, _ = env.reset(location="x", seed=0)
        np.testing.assert_array_equal(obs, d["obs0"][i])
        for t, a in enumerate(d["actions"][i]):
            obs, r, done, _, info = env.step(int(a))
            assert r == d["reward"][i, t]
            np.testing.assert_array_equal(obs, d["obs"][i, t])
