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
# returns summed over up to 153 days in f32 inside a kernel against the float64 oracle: the north star's 1e-5 is a per-step
# reward bound; a sum of n rewards may differ by n x 1e-5 at most (1.5e-3 per episode). Measured: <= 5.3e-7 relative
# (returns of magnitude 10^2..10^3), so the suite holds the kernels to 2e-6 relative + 2e-5 absolute
RETURN_RTOL, RETURN_ATOL = 2e-6, 2e-5
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


# ---------------------------------------------------------------------------------------------------------------
# every other path bench.py quotes at 1 048 576 envs: the on-device rollout with its visiting order, the
# posterior-mean reward (step and whole-episode rollout, vector and matrix kernels) and episode_order="sorted".
# Position arithmetic, tile-list sizing and the per-XCD tile walk only break at this size.
# ---------------------------------------------------------------------------------------------------------------
def _sample(n, k):
    """~k env ids, strided, always including env 0 and the LAST env."""
    return np.unique(np.concatenate([np.arange(0, n, max(n // k, 1)), [n - 1]]))


def test_eight_million_envs_on_one_gpu(dev):
    """BASELINE configs[4]'s TOTAL batch (8 388 608 envs) + a ragged tail of 13 on one GPU: observation offsets pass
    2^31 bytes (973 MB of rows), the last wave is partial. 20 days of step() and a whole-episode rollout on the
    matrix-core kernel against the oracle on a strided sample that includes env 0 and the LAST env."""
    from weather2alert_amd import HeatAlertVecEnv

    sd, ct, dt, V = full_tables("nn_full_medicare_all", dev)
    V.reward_mode = "sampled"
    n = 8388608 + 13
    env = HeatAlertVecEnv(n, tables=dt, device=dev, autoreset="disabled")
    obs, _ = env.reset(seed=77)
    idx = _sample(n, 4096)
    assert idx[-1] == n - 1
    it = torch.as_tensor(idx, device=dev)
    st = {k: v[it].cpu().numpy() for k, v in env.state().items()}
    obs_o = V.reset(st["county_w"], st["year_i"], st["coef_col"], st["sample"], st["budget"])
    np.testing.assert_array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32))
    g = torch.Generator(device=dev).manual_seed(9)
    worst = 0.0
    for t in range(20):
        a = (torch.rand(n, device=dev, generator=g) < 0.15).to(torch.uint8)
        obs, r, done, _, _ = env.step(a)
        obs_o, r_o, done_o, _ = V.step(a[it].cpu().numpy().astype(np.int32))
        worst = max(worst, float(np.abs(r[it].cpu().numpy().astype(np.float64) - r_o).max()))
        np.testing.assert_array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32))
    assert worst <= REWARD_TOL and env.step_kernel_name == "k_step64" and env.packed_state
    # the rest of the episode in one launch (lock step known to the handle: the matrix-core kernel serves it)
    V._finished = np.zeros(len(idx), bool)
    pol = dict(kind="threshold", feature="heat_qi", threshold=0.85, require_budget=True)
    out = env.rollout(pol, alert_mask=True)
    assert env.last_rollout_kernel == "k_rollout_mfma"
    ret_o, al_o, ov_o, days_o = O.oracle_rollout(V, dict(pol, col=ct.columns.index("heat_qi")), ct.T, None)
    np.testing.assert_array_equal(out["alerts"][it].cpu().numpy(), al_o)
    np.testing.assert_array_equal(out["alert_days"][it].cpu().numpy(), days_o)
    np.testing.assert_allclose(out["return"][it].cpu().numpy(), ret_o, rtol=RETURN_RTOL, atol=RETURN_ATOL)
    sa = env.state()
    assert bool(out["done"].all()) and bool((sa["used"] <= sa["budget"]).all()) and env.check_status() == 0
    print(f"8 388 621 envs: sample {len(idx)} envs, max |reward - oracle| = {worst:.3e}")
    env.close()
    del env, obs, out, sa
    torch.cuda.empty_cache()


class _Draw:
    """Bernoulli-policy uniforms of the sampled envs (the build's counter RNG, restated in the oracle)."""

    def __init__(self, seed, gids, episode_no):
        self.seed, self.gids, self.ep = seed, np.asarray(gids, np.uint64), np.asarray(episode_no, np.uint64)

    def vec(self, t):
        return O.devrng_policy_uniform_vec(self.seed, self.gids, self.ep, t)

    def __call__(self, i, t):
        return O.devrng_policy_uniform(self.seed, int(self.gids[i]), int(self.ep[i]), t)


@pytest.mark.parametrize("kind,kernel", [("threshold", "k_rollout_mfma"), ("bernoulli", "k_rollout_mfma"),
                                         ("threshold", "k_rollout64"), ("bernoulli", "k_rollout64")])
def test_full_size_rollout_with_visiting_order_vs_oracle(dev, kind, kernel):
    """rollout() at BASELINE configs[2] size (1 048 576 envs, S = 746, 100 draws, similar_climate_counties) through
    the counting-sort visiting order (w2a_rollout_order) and either the matrix-core kernel (k_rollout_mfma: feature-row
    tile list, int8 MFMAs for the table part of the logits) or k_rollout64, whole episode in one launch, against the
    oracle's policy loop on ~8 192 sampled envs (incl. env 0 and the last one): alerts, over-budget attempts, alert-day
    and attempt-day bitmaps exact, returns <= 2e-5 relative; every env finished; the order is a permutation."""
    from weather2alert_amd import HeatAlertVecEnv

    sd, ct, dt, V = full_tables("linear", dev)
    V.reward_mode = "sampled"
    n, gid0 = 1 << 20, 12345
    env = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=True, autoreset="disabled", env_gid0=gid0,
                          rollout_order=True, rollout_mfma=kernel == "k_rollout_mfma")
    env.reset(seed=31)
    idx = _sample(n, 8192)
    it = torch.as_tensor(idx, device=dev)
    st = {k: v[it].cpu().numpy() for k, v in env.state().items()}
    V.reset(st["county_w"], st["year_i"], st["coef_col"], st["sample"], st["budget"])
    V._finished = np.zeros(len(idx), bool)
    if kind == "threshold":
        pol = dict(kind="threshold", feature="heat_qi", threshold=0.9, require_budget=True)
        draw = None
    else:
        pol = dict(kind="bernoulli", p=0.1, seed=99)
        draw = _Draw(99, gid0 + idx, st["episode_no"])
        np.testing.assert_array_equal(draw.vec(np.full(len(idx), 3))[:5], [draw(i, 3) for i in range(5)])
    out = env.rollout(pol, alert_mask=True)
    assert env._order_ws is not None and not env._order_stale  # the kernel ran on the visiting order
    assert env.last_rollout_kernel == kernel
    order = env._order_ws[: 4 * n].view(torch.int32)
    assert torch.equal(torch.sort(order.long()).values, torch.arange(n, device=dev))  # a permutation of the env ids
    ret_o, al_o, ov_o, days_o = O.oracle_rollout(V, dict(pol, col=ct.columns.index("heat_qi")), ct.T, draw)
    np.testing.assert_array_equal(out["alerts"][it].cpu().numpy(), al_o)
    np.testing.assert_array_equal(out["attempts_over_budget"][it].cpu().numpy(), ov_o)
    np.testing.assert_array_equal(out["alert_days"][it].cpu().numpy(), days_o)
    got = out["return"][it].cpu().numpy().astype(np.float64)
    rel = np.abs(got - ret_o).max() / np.abs(ret_o).max()
    np.testing.assert_allclose(got, ret_o, rtol=RETURN_RTOL, atol=RETURN_ATOL)
    np.testing.assert_allclose(out["final_return"][it].cpu().numpy(), ret_o, rtol=RETURN_RTOL, atol=RETURN_ATOL)
    assert bool(out["done"].all()) and int(out["alerts"].sum()) > n // 2  # all 1 048 576 envs ran to their last day
    s2 = {k: v[it].cpu().numpy() for k, v in env.state().items()}
    np.testing.assert_array_equal(s2["used"], V.used)
    np.testing.assert_array_equal(s2["streak"], V.streak)
    np.testing.assert_array_equal(s2["t"], V.t)
    # invariants over ALL envs: alerts never exceed the budget, attempts over budget only once the budget is used up
    sa = env.state()
    assert bool((sa["used"] <= sa["budget"]).all()) and bool((out["alerts"] == sa["used"]).all())
    assert bool(((out["attempts_over_budget"] == 0) | (sa["used"] == sa["budget"])).all())
    assert env.check_status() == 0
    print(f"full-size rollout [{kind}, {kernel}]: sample {len(idx)} envs, max rel |return - oracle| = {rel:.3e}")
    env.close()


@pytest.mark.parametrize("pm_kernel", ["vector", "matrix", "matrix_i8"])
def test_full_size_posterior_mean_step_vs_oracle(dev, pm_kernel):
    """reward_mode='posterior_mean' step() at 1 048 576 envs on the nn_full_medicare_all shape (S = 720, 100 draws:
    BASELINE configs[3], the 'dense reward GEMM'), both kernels of the library, 20 days against the oracle's mean over
    all 100 draws on ~4 096 sampled envs (incl. env 0 and the last): w2a_group_by_column's radix sort, tile list and
    per-XCD tile walk at full size. Reward <= 1e-5, observations and integer state exact."""
    from weather2alert_amd import HeatAlertVecEnv

    sd, ct, dt, V = full_tables("nn_full_medicare_all", dev)
    V.reward_mode = "posterior_mean"
    try:
        n = 1 << 20
        env = HeatAlertVecEnv(n, tables=dt, device=dev, autoreset="disabled", reward_mode="posterior_mean",
                              pm_kernel=pm_kernel)
        obs, _ = env.reset(seed=17)
        idx = _sample(n, 4096)
        it = torch.as_tensor(idx, device=dev)
        st = {k: v[it].cpu().numpy() for k, v in env.state().items()}
        assert len(np.unique(st["coef_col"])) > 700  # the sample spans the coefficient table
        obs_o = V.reset(st["county_w"], st["year_i"], st["coef_col"], st["sample"], st["budget"])
        np.testing.assert_array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32))
        g = torch.Generator(device=dev).manual_seed(9)
        worst = 0.0
        for t in range(20):
            a = (torch.rand(n, device=dev, generator=g) < 0.3).to(torch.int32)
            obs, r, done, _, _ = env.step(a)
            obs_o, r_o, done_o, _ = V.step(a[it].cpu().numpy())
            err = np.abs(r[it].cpu().numpy().astype(np.float64) - r_o).max()
            worst = max(worst, err)
            assert err <= REWARD_TOL, (t, err)
            np.testing.assert_array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32))
            assert torch.isfinite(r).all() and float(r.max()) < 0.0  # every env got a reward (none skipped)
        s2 = {k: v[it].cpu().numpy() for k, v in env.state().items()}
        np.testing.assert_array_equal(s2["used"], V.used)
        np.testing.assert_array_equal(s2["streak"], V.streak)
        assert env.check_status() == 0
        print(f"full-size posterior mean step [{pm_kernel}]: sample {len(idx)} envs, max |reward - oracle| = {worst:.3e}")
        env.close()
    finally:
        V.reward_mode = "sampled"


@pytest.mark.parametrize("pm_kernel", ["vector", "matrix_i8"])
def test_full_size_posterior_mean_rollout_vs_oracle(dev, pm_kernel):
    """rollout() through k_pm_rollout / k_pm_rollout_i8 (one launch per episode) at 1 048 576 envs, S = 746, 100 draws, augmented:
    whole episode against the oracle's policy loop on the all-draws reward for ~1 024 sampled envs; the per-day
    sequence on the matrix kernel agrees with it on every env for the first days."""
    from weather2alert_amd import HeatAlertVecEnv

    sd, ct, dt, V = full_tables("linear", dev)
    V.reward_mode = "posterior_mean"
    try:
        n, gid0 = 1 << 20, 777
        env = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=True, autoreset="disabled",
                              reward_mode="posterior_mean", pm_kernel=pm_kernel, env_gid0=gid0)
        env.reset(seed=23)
        idx = _sample(n, 1024)
        it = torch.as_tensor(idx, device=dev)
        st = {k: v[it].cpu().numpy() for k, v in env.state().items()}
        V.reset(st["county_w"], st["year_i"], st["coef_col"], st["sample"], st["budget"])
        V._finished = np.zeros(len(idx), bool)
        pol = dict(kind="bernoulli", p=0.1, seed=5)
        draw = _Draw(5, gid0 + idx, st["episode_no"])
        # a second env with the same episodes: the matrix kernel through the per-day sequence, first 3 days
        e2 = HeatAlertVecEnv(n, tables=dt, device=dev, similar_climate_counties=True, autoreset="disabled",
                             reward_mode="posterior_mean", pm_kernel="matrix", env_gid0=gid0)
        e2.reset(seed=23)
        o1 = env.rollout(pol, n_steps=3)
        o2 = e2.rollout(pol, n_steps=3)
        assert torch.equal(o1["alerts"], o2["alerts"])
        torch.testing.assert_close(o1["return"], o2["return"], rtol=1e-5, atol=1e-5)
        e2.close()
        r1, a1, v1, _ = O.oracle_rollout(V, pol, 3, draw)
        out = env.rollout(pol, alert_mask=True)
        r2, a2, v2, days_o = O.oracle_rollout(V, pol, ct.T, draw)
        assert bool(out["done"].all()) and bool((out["first_day"] == 3).all())
        np.testing.assert_array_equal((o1["alerts"] + out["alerts"])[it].cpu().numpy(), a1 + a2)
        np.testing.assert_array_equal((o1["attempts_over_budget"] + out["attempts_over_budget"])[it].cpu().numpy(), v1 + v2)
        np.testing.assert_array_equal(out["alert_days"][it].cpu().numpy()[:, 3:], days_o[:, 3:])
        np.testing.assert_allclose(o1["return"][it].cpu().numpy(), r1, rtol=RETURN_RTOL, atol=RETURN_ATOL)
        np.testing.assert_allclose(out["return"][it].cpu().numpy(), r2, rtol=RETURN_RTOL, atol=RETURN_ATOL)
        np.testing.assert_allclose(out["final_return"][it].cpu().numpy(), r1 + r2, rtol=RETURN_RTOL, atol=RETURN_ATOL)
        assert env.check_status() == 0
        env.close()
    finally:
        V.reward_mode = "sampled"


def test_full_size_sorted_reset_paths_agree(dev):
    """episode_order='sorted' on BASELINE's tables at 1 048 576 envs and just above: the fused reset (32-bit keys, rocprim
    onesweep with 9-bit passes) and the three-call sequence (64-bit keys through hipcub: a merge sort up to 1 048 576
    items, a radix sort above -- the two library paths take the bit range differently, see w2a_sort_episodes) give the same
    batch, which is THE stable sort of the iid batch by coefficient row; through a second, sticky episode as well."""
    from weather2alert_amd import HeatAlertVecEnv

    sd, ct, dt, V = full_tables("linear", dev)
    kw = dict(tables=dt, device=dev, similar_climate_counties=True)
    opts = {"sample_budget": True, "sample_budget_type": "centered"}
    for n in (1 << 20, (1 << 20) + 64 * 3 + 5):
        iid = HeatAlertVecEnv(n, autoreset="disabled", **kw)
        fused = HeatAlertVecEnv(n, episode_order="sorted", sorted_reset="fused", **kw)
        relab = HeatAlertVecEnv(n, episode_order="sorted", sorted_reset="relabel", **kw)
        for episode in range(2):
            for e in (iid, fused, relab):
                e.reset(seed=11 + episode, options=opts)
            si, sf, sr = iid.state(), fused.state(), relab.state()
            for k in sf:
                assert torch.equal(sf[k], sr[k]), (n, episode, k)
            if episode == 0:  # (afterwards a sticky budget follows its RECORD, which the relabelling has moved to another index)
                perm = torch.sort(si["coef_col"].long() << 12 | si["sample"].long(), stable=True).indices
                for k in ("county_w", "year_i", "coef_col", "sample", "budget", "sticky_budget"):
                    assert torch.equal(sf[k], si[k][perm]), (n, episode, k)
            assert torch.equal(fused._obs, relab._obs)
        assert fused.check_status() == 0 and relab.check_status() == 0
        for e in (iid, fused, relab):
            e.close()


def test_full_size_sorted_episode_order_vs_oracle(dev):
    """episode_order='sorted' at 1 048 576 envs (S = 746, augmented): the relabelled batch holds the same multiset of
    episode records as the iid order for the same seed, env indices follow the coefficient rows, and one whole
    episode steps like the oracle on a strided sample (incl. env 0 and the last env)."""
    from weather2alert_amd import HeatAlertVecEnv

    sd, ct, dt, V = full_tables("linear", dev)
    V.reward_mode = "sampled"
    n = 1 << 20
    kw = dict(tables=dt, device=dev, similar_climate_counties=True)
    iid = HeatAlertVecEnv(n, autoreset="disabled", **kw)
    srt = HeatAlertVecEnv(n, episode_order="sorted", **kw)
    iid.reset(seed=5)
    obs, _ = srt.reset(seed=5)
    keys = ("county_w", "year_i", "coef_col", "sample", "budget", "sticky_budget", "episode_no")

    def packed(e):  # one int64 key per env: the whole record (field widths: 10, 4, 10, 7, 16, 16 bits; episode_no = 0)
        s = e.state()
        assert int(s["episode_no"].max()) == 0 and int(s["budget"].max()) < 65535 and int(s["sticky_budget"].max()) < 65535
        k = s["county_w"].long()
        for name, bits in (("year_i", 4), ("coef_col", 10), ("sample", 7), ("budget", 16), ("sticky_budget", 16)):
            k = (k << bits) | (s[name].long() + (1 if name == "sticky_budget" else 0))
        return k

    ka, kb = packed(iid), packed(srt)
    assert torch.equal(torch.sort(ka).values, torch.sort(kb).values)  # same multiset of episode records
    sb = srt.state()
    row = sb["coef_col"].long() << 12 | sb["sample"].long()
    assert bool((row[1:] >= row[:-1]).all()) and not torch.equal(ka, kb)
    sa = iid.state()  # envs of one coefficient row keep the iid order's order: the permutation is THE stable sort
    assert torch.equal(kb, ka[torch.sort(sa["coef_col"].long() << 12 | sa["sample"].long(), stable=True).indices])
    iid.close()
    idx = _sample(n, 16384)
    it = torch.as_tensor(idx, device=dev)
    st = {k: v[it].cpu().numpy() for k, v in sb.items()}
    obs_o = V.reset(st["county_w"], st["year_i"], st["coef_col"], st["sample"], st["budget"])
    np.testing.assert_array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32))
    g = torch.Generator(device=dev).manual_seed(2)
    ret = np.zeros(len(idx))
    for t in range(153):
        a = (torch.rand(n, device=dev, generator=g) < 0.15).to(torch.int32)
        obs, r, done, _, info = srt.step(a)
        obs_o, r_o, done_o, _ = V.step(a[it].cpu().numpy())
        assert np.abs(r[it].cpu().numpy() - r_o).max() <= REWARD_TOL
        np.testing.assert_array_equal(done[it].cpu().numpy(), done_o)
        ret += r_o
        if t < 152:
            np.testing.assert_array_equal(obs[it].cpu().numpy(), obs_o.astype(np.float32))
    assert bool(done.all())
    np.testing.assert_allclose(info["final_return"][it].cpu().numpy(), ret, rtol=2e-5)
    # the lock-step autoreset relabelled the next episode too
    s3 = srt.state()
    assert bool((s3["episode_no"] == 1).all()) and bool((s3["t"] == 0).all())
    row = s3["coef_col"].long() << 12 | s3["sample"].long()
    assert bool((row[1:] >= row[:-1]).all())
    assert srt.check_status() == 0
    srt.close()
