"""Full-size parity on the GPU (BASELINE configs[1..3] shapes): the reference's own full-size anchor episodes
(tests/golden/full_anchor.npz, captured from the unmodified reference env at S = 746) through libw2a.so, and the
HIP path against the float64 oracle on the exact BASELINE table shapes -- S = 746 / 720 counties, 11 years,
100 posterior draws, device-RNG episode tuples -- at 65 536 envs and on a strided sample (including the last env)
of 1 048 576-env batches. 32-bit offset arithmetic and tile-tail handling only break at these sizes.

Bars (north_star): integer state and observations bit-exact, reward within 1e-5."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import heatalert_oracle as O
from weather2alert_amd import synth, tables

pytestmark = pytest.mark.gpu
REWARD_TOL = 1e-5
YEARS = list(range(2006, 2017))
_CACHE = {}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    return torch.device("cuda:0")


def full_tables(weights, dev):
    """(synth data, compiled tables, device tables, vector oracle) of one BASELINE table shape, built once."""
    if weights not in _CACHE:
        from weather2alert_amd.tables import DeviceTables

        sd = synth.make_synth(weights, years=YEARS, n_samples=100, seed=0, extra_confounder_fips=60)
        ct = tables.compile_from_synth(sd)
        V = O.VectorOracle(O.RefData.from_synth(sd), sd.fips_weather, sd.years)
        _CACHE[weights] = (sd, ct, DeviceTables(ct, dev), V)
    return _CACHE[weights]


def test_reference_full_size_anchor_episodes(golden_dir, dev):
    """The six episodes the reference produced on the 746-county weight list (make_golden.py: 41 weather counties,
    11 years, 100 draws). The data set is regenerated from the same seed, checked value by value against the episode
    tables and coefficient vectors the reference actually used, then stepped through the drop-in env (NumPy seed
    parity at full size: '06037' is column 84, 111 similar counties) and through the vector env."""
    from weather2alert_amd import HeatAlertEnv, HeatAlertVecEnv
    from weather2alert_amd.tables import DeviceTables

    d = dict(np.load(os.path.join(golden_dir, "full_anchor.npz")))
    meta = json.loads(str(d["meta_json"]))
    sp = meta["synth"]
    sd = synth.make_synth(sp["weights_name"], n_counties_weather=sp["n_counties_weather"], years=YEARS, n_samples=100,
                          seed=sp["seed"], extra_confounder_fips=sp["extra_confounder_fips"])
    ct = tables.compile_from_synth(sd)
    assert ct.S == meta["n_fips"] == 746 and ct.fips_list.index("06037") == meta["index_06037"] == 84
    assert int(ct.sim_cnt[84]) == meta["n_similar_06037"] == 111
    names = meta["feature_names"]
    assert ct.feature_names == names
    E = len(meta["episodes"])
    cw = [ct.fips_weather.index(e["episode_index"].split("_")[0]) for e in meta["episodes"]]
    yi = [ct.years.index(int(e["episode_index"].split("_")[1])) for e in meta["episodes"]]
    # pin the regenerated tables to what the reference saw
    runtime = {"alert_lag1", "alert_streak", "remaining_budget"}
    for i in range(E):
        row = cw[i] * ct.Y + yi[i]
        for j, c in enumerate(names[:-1]):
            if c not in runtime:
                np.testing.assert_array_equal(ct.X[:, row, ct.slot_of[c]].astype(np.float64), d["episode_table"][i][:, j])
        assert ct.B0[row] == d["episode_table"][i][0, names.index("remaining_budget")]
        wrow = ct.W[int(d["location_index"][i]) * ct.n_samples + int(d["coef_index"][i])]
        for head, keys, pre in ((0, meta["baseline_keys"], "baseline_"), (1, meta["effectiveness_keys"], "effectiveness_")):
            for j, k in enumerate(keys):
                assert wrow[head, ct.slot_of[k.replace(pre, "")]] == d["episode_weights"][i, head, j]
    dt = DeviceTables(ct, dev)
    # (1) drop-in env: seeds -> the reference's episodes, then its trajectories
    worst = 0.0
    for i, e in enumerate(meta["episodes"]):
        env = HeatAlertEnv(weights="linear", tables=dt, device=dev)  # each anchor episode ran on a fresh budget
        obs, info = env.reset(**e["reset"])
        assert info["episode_index"] == e["episode_index"] and info["location"] == e["info_location"]
        assert info["location_index"] == d["location_index"][i] and env.coef_index == d["coef_index"][i]
        assert env.budget == d["budget"][i] and info["remaining_budget"] == d["reset_remaining_budget"][i]
        np.testing.assert_array_equal(obs, d["obs0"][i].astype(np.float32))
        for t in range(153):
            obs, r, done, trunc, info = env.step(int(d["actions"][i, t]))
            worst = max(worst, abs(r - d["reward"][i, t]))
            assert abs(r - d["reward"][i, t]) <= REWARD_TOL and done == d["done"][i, t]
            np.testing.assert_array_equal(obs, d["obs"][i, t].astype(np.float32))
            assert info["remaining_budget"] == d["remaining_budget"][i, t] and info["at_budget"] == d["at_budget"][i, t]
            assert env.alert_streak == d["streak_after"][i, t] and env.t == d["t_after"][i, t]
        env.close()
    # (2) the same six episodes as one batch of injected tuples, on both step kernels
    for kernel in ("wide", "classic"):
        v = HeatAlertVecEnv(E, tables=dt, device=dev, autoreset="disabled", step_kernel=kernel)
        obs, _ = v.reset(options={"episodes": dict(county_w=cw, year_i=yi, coef_col=d["location_index"],
                                                   sample=d["coef_index"], budget=d["budget"])})
        np.testing.assert_array_equal(obs.cpu().numpy(), d["obs0"].astype(np.float32))
        for t in range(153):
            obs, r, done, _, info = v.step(torch.as_tensor(d["actions"][:, t], device=dev))
            np.testing.assert_allclose(r.cpu().numpy(), d["reward"][:, t], rtol=0, atol=REWARD_TOL)
            np.testing.assert_array_equal(done.cpu().numpy(), d["done"][:, t])
            np.testing.assert_array_equal(obs.cpu().numpy(), d["obs"][:, t].astype(np.float32))
            np.testing.assert_array_equal(info["remaining_budget"].cpu().numpy(), d["remaining_budget"][:, t])
            st = v.state()
            np.testing.assert_array_equal(st["streak"].cpu().numpy(), d["streak_after"][:, t])
            np.testing.assert_array_equal(st["last_actual"].cpu().numpy(), d["actual"][:, t])
        assert v.check_status() == 0
        v.close()
    print(f"full-size anchors: max |reward - reference| = {worst:.3e}")


@pytest.mark.parametrize("config,weights,n,augment,kernel", [
    ("configs[1]", "linear", 65536, False, "auto"),
    ("configs[1]", "linear", 65536, False, "wide"),
    ("configs[2]", "linear", 1 << 20, True, "auto"),
    ("configs[3]", "nn_full_medicare_all", 1 << 20, False, "classic"),
    ("configs[3]", "nn_full_medicare_all", 1 << 20, False, "auto"),
])
def test_baseline_shapes_vs_oracle(dev, config, weights, n, augment, kernel):
    """HIP path vs the float64 vector oracle on the exact BASELINE shape: the batch resets with the device RNG
    (random county per env / similar_climate_counties), the drawn tuples are read back with state() and replayed on
    the oracle for a strided sample of <= 65 537 envs that includes env 0 and the LAST env; one whole episode."""
    from weather2alert_amd import HeatAlertVecEnv

    sd, ct, dt, V = full_tables(weights, dev)
    assert ct.S == (746 if weights == "linear" else 720) and ct.Y == 11 and ct.n_samples == 100 and ct.T == 153
    env = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=augment, autoreset="disabled",
                          step_kernel=kernel)
    assert env.step_kernel_name == ("k_step" if kernel == "classic" or (kernel == "auto" and n < 131072) else "k_step64")
    obs, _ = env.reset(seed=20 + n % 7)
    idx = np.unique(np.concatenate([np.arange(0, n, max(n // 65536, 1)), [n - 1]]))
    it = torch.as_tensor(idx, device=dev)
    st = {k: v[it].cpu().numpy() for k, v in env.state().items()}
    if augment:  # Q8: the coefficient column is a position inside the filtered similar list, weather stays
        assert (st["coef_col"] < ct.sim_cnt.max()).all() and len(np.unique(st["county_w"])) > 700
    else:
        np.testing.assert_array_equal(ct.fips_to_weather[st["coef_col"]], st["county_w"])
    obs_o = V.reset(st["county_w"], st["year_i"], st["coef_col"], st["sample"], st["budget"])
    np.testing.assert_array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32))
    g = torch.Generator(device=dev).manual_seed(5)
    worst = 0.0
    for t in range(153):
        a = (torch.rand(n, device=dev, generator=g) < 0.12).to(torch.int32)
        obs, r, done, _, _ = env.step(a)
        obs_o, r_o, done_o, _ = V.step(a[it].cpu().numpy())
        err = np.abs(r[it].cpu().numpy().astype(np.float64) - r_o).max()
        worst = max(worst, err)
        assert err <= REWARD_TOL, (t, err)
        np.testing.assert_array_equal(done[it].cpu().numpy(), done_o)
        np.testing.assert_array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32))
    assert done.all()
    s2 = {k: v[it].cpu().numpy() for k, v in env.state().items()}
    np.testing.assert_array_equal(s2["used"], V.used)
    np.testing.assert_array_equal(s2["streak"], V.streak)
    np.testing.assert_array_equal(s2["t"], V.t)
    np.testing.assert_array_equal(s2["budget"] - s2["used"], V.budget - V.used)
    assert env.check_status() == 0
    print(f"{config} {weights} n={n} kernel={kernel}: sample {len(idx)} envs, max |reward - oracle| = {worst:.3e}")
    env.close()
