"""CPU ORACLE for the HeatAlertEnv reset/step hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy / pure-Python restatement of the reference algorithm
(`/root/reference/src/weather2alert/env.py:107-262`, `datautils.py:103-126`). It exists so
that tests, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg can CHECK the HIP
path; nothing under `weather2alert_amd/` may import it and the product never falls back
to it.

Parity pin: the reference has no tests or golden vectors of its own, so this
oracle is pinned by vectors captured from the unmodified reference env run in the build
container (`tests/golden/make_golden.py` -> `tests/golden/*.npz`): `tests/test_oracle_golden.py`
requires integer state bit-exact and float64 rewards/observations bit-exact against them.

Layout of the restatement (each function cites what it follows):

  RefData.from_files     env.py:39-85,104-105   table/weight loading (pandas, like the reference)
  similar_counties       datautils.py:103-126   climate-zone augmentation set
  OracleEnv.reset        env.py:133-184, 107-131
  OracleEnv._get_obs     env.py:186-195
  OracleEnv._get_reward  env.py:197-226
  OracleEnv.step         env.py:238-262
  VectorOracle           the same step arithmetic vectorised over envs (float64, same
                         summation order), used for large-N checks and the CPU baseline
  devrng_*               restatement of the build's own counter-based device RNG
                         (weather2alert_amd/csrc/w2a_common.hip.h: w2a_mix64 / draw slots);
                         this part has no reference counterpart
"""
from __future__ import annotations

import os

import numpy as np
from scipy.special import expit as _expit  # the reference's sigmoid (env.py:12)

WESTERN_STATE_FIPS = {  # datautils.py:3-17 via FIPS2STATE (:42-100): AZ CA CO ID MT NM NV OR WA ND SD NE KS
    "03", "04", "06", "08", "16", "30", "35", "32", "41", "53", "38", "46", "31", "20",
}
_KNOWN_STATE_PREFIXES = {
    "01", "02", "03", "04", "05", "06", "08", "09", "10", "11", "12", "13", "15", "16", "17", "18", "19", "20",
    "21", "22", "23", "24", "25", "26", "27", "28", "29", "30", "31", "32", "33", "34", "35", "36", "37", "38",
    "39", "40", "41", "42", "44", "45", "46", "47", "48", "49", "50", "51", "53", "54", "55", "56", "72", "60",
    "66", "69", "78",
}


def similar_counties(fips: str, conf_fips: list[str], conf_zone: list[str]) -> list[str]:
    """datautils.py:103-126. Counties of a WESTERN_STATES state are all 'Cold-West' whatever
    their zone (:113-117); other 'Cold' ones become 'Cold-East'; result keeps the confounders'
    row order (:124)."""

    def zone_of(f, z):
        if f[:2] in WESTERN_STATE_FIPS:
            return "Cold-West"
        return "Cold-East" if z == "Cold" else z

    zones = [zone_of(f, z) for f, z in zip(conf_fips, conf_zone)]
    mine = zones[conf_fips.index(fips)]  # confounders.loc[fips] -> KeyError/ValueError if absent
    return [f for f, z in zip(conf_fips, zones) if z == mine]


class RefData:
    """Everything `HeatAlertEnv.__init__` holds (env.py:39-105), as plain arrays."""

    def __init__(self):
        self.columns: list[str] = []  # the 28 episode columns (21 exo + 7 endo), env.py:128-130
        self.episodes: dict[tuple[str, int], np.ndarray] = {}  # (fips, year) -> f64 [n_days, 28]
        self.fips_list: list[str] = []
        self.valid_years: list[int] = []
        self.n_samples = 0
        self.baseline_keys: list[str] = []
        self.effectiveness_keys: list[str] = []
        self.wb = None  # f32 [K, n_samples, S] rows in baseline_keys order
        self.we = None
        self.conf_fips: list[str] = []
        self.conf_zone: list[str] = []
        self.sig_categories: list[str] = []

    # ---- loading from the reference's on-disk format ---------------------------------
    @classmethod
    def from_files(cls, data_dir: str, weights: str = "nn_full_medicare_all", split: str = "65k",
                   years: list | None = None) -> "RefData":
        import pandas as pd
        import yaml
        from safetensors import safe_open

        self = cls()
        ddir = os.path.join(data_dir, "data", split)
        merged = pd.merge(  # env.py:49-53
            pd.read_parquet(os.path.join(ddir, "exogenous_states.parquet")),
            pd.read_parquet(os.path.join(ddir, "endogenous_states_actions.parquet")),
            on=["fips", "date"],
        )
        merged["year"] = merged.date.str[:4].astype(int)  # env.py:54
        conf = pd.read_parquet(os.path.join(ddir, "confounders.parquet"))  # env.py:57
        self.conf_fips = [str(x) for x in conf["fips"].tolist()]
        self.conf_zone = [str(x) for x in conf["ba_zone"].tolist()]
        self.columns = [c for c in merged.columns if c not in ("fips", "date", "year")]
        cats = sorted(x for x in merged["significance"].dropna().unique()) if "significance" in merged else []
        self.sig_categories = list(cats)
        num = merged[self.columns].copy()
        if "significance" in num:
            num["significance"] = merged["significance"].map(lambda v: 0.0 if v is None or v != v
                                                             else float(cats.index(v) + 1))
        vals = num.astype(np.float64).values
        fy = list(zip(merged["fips"].tolist(), merged["year"].tolist()))
        # rows of one (fips, year) in file order, like merged.loc[(location, year)] (env.py:127)
        groups: dict[tuple[str, int], list[int]] = {}
        for i, k in enumerate(fy):
            groups.setdefault(k, []).append(i)
        self.episodes = {k: vals[idx] for k, idx in groups.items()}
        # env.py:104-105: years in order of first appearance, unless given
        if years is None:
            self.valid_years = [int(y) for y in pd.unique(merged["year"])]
        else:
            self.valid_years = list(years)
        post = {}
        with safe_open(os.path.join(data_dir, weights, "posterior_samples.safetensors"), framework="np") as f:
            for k in f.keys():  # env.py:69-72 -- dict order = file key order
                post[k] = f.get_tensor(k)
        cfg = yaml.safe_load(open(os.path.join(data_dir, weights, "config.yaml")))
        self.fips_list = [str(x) for x in cfg["fips_list"]]  # env.py:75
        self._set_weights(post)
        return self

    def _set_weights(self, post: dict):
        self.baseline_keys = [k for k in post if k.startswith("baseline")]  # env.py:77-79
        self.effectiveness_keys = [k for k in post if k.startswith("effectiveness")]  # env.py:80-82
        self.n_samples = int(post["baseline_bias"].shape[0])  # env.py:85
        self.wb = np.stack([np.asarray(post[k], np.float32)[:, 0, :] for k in self.baseline_keys])
        self.we = np.stack([np.asarray(post[k], np.float32)[:, 0, :] for k in self.effectiveness_keys])

    @classmethod
    def from_synth(cls, d, sorted_keys: bool = True) -> "RefData":
        """Same content from a dense synthetic data set (duck-typed: weather2alert_amd.synth.SynthData),
        without the parquet round trip. Column order = exogenous then endogenous file order."""
        self = cls()
        exo_cols, endo_cols = d.meta["exo_cols"], d.meta["endo_cols"]
        self.columns = list(exo_cols) + list(endo_cols)
        endo = np.stack([np.asarray(getattr(d, c), np.float64) for c in endo_cols], axis=-1)
        full = np.concatenate([d.exo.astype(np.float64), endo], axis=-1)
        ragged = d.meta.get("n_days_per_episode")  # optional [S_w, Y] episode lengths, 0 = pair absent
        for ci, f in enumerate(d.fips_weather):
            for yi, y in enumerate(d.years):
                nd = full.shape[2] if ragged is None else int(ragged[ci][yi])
                if nd > 0:
                    self.episodes[(f, int(y))] = full[ci, yi, :nd]
        self.valid_years = [int(y) for y in d.years]
        self.fips_list = list(d.fips_list)
        self.conf_fips, self.conf_zone = list(d.confounder_fips), list(d.confounder_zone)
        self.sig_categories = list(d.meta.get("sig_categories", []))
        keys = sorted(d.weights) if sorted_keys else list(d.weights)  # safetensors lists keys sorted
        self._set_weights({k: d.weights[k] for k in keys})
        return self

    # feature position of each coefficient key inside an episode row (+ 'bias')
    def key_columns(self, keys: list[str], prefix: str) -> list[int]:
        out = []
        for k in keys:
            name = k.replace(prefix, "")  # env.py:208,215
            out.append(-1 if name == "bias" else self.columns.index(name))
        return out


class OracleEnv:
    """Scalar restatement of HeatAlertEnv; attribute names follow the reference."""

    def __init__(self, data: RefData, similar_climate_counties: bool = False, budget: int | None = None,
                 eval_mode: bool = False):
        # eval_mode: the LEGACY env's evaluation switch (_deprecated/env.py:64,332-342): the step reward is the mean
        # over ALL posterior draws instead of the episode's one draw. The current env.py has no such switch; the
        # restatement applies the legacy averaging to today's reward form (env.py:197-226). Not pinned by goldens
        # (the legacy env computes a different functional form), only by agreement of the two restatements below.
        self.eval_mode = eval_mode
        self.d = data
        self.similar_climate_counties = similar_climate_counties
        self.budget = budget  # env.py:34 (sticky once set, :167-170)
        self.fips_list = data.fips_list
        self.valid_years = data.valid_years
        self.n_samples = data.n_samples
        self._bcols = data.key_columns(data.baseline_keys, "baseline_")
        self._ecols = data.key_columns(data.effectiveness_keys, "effectiveness_")
        self._i_lag1 = data.columns.index("alert_lag1")
        self._i_streak = data.columns.index("alert_streak")
        self._i_rem = data.columns.index("remaining_budget")
        self._i_hq = data.columns.index("heat_qi")
        self.feat_names = list(data.columns) + ["alert_2wks"]  # env.py:191 adds a NEW key (Q1)

    # env.py:107-131
    def _get_episode(self, location: str, augment: bool):
        if augment:
            locations = similar_counties(location, self.d.conf_fips, self.d.conf_zone)
            locations = [x for x in locations if x in self.fips_list]
            self.location_index = int(self.rng.choice(range(len(locations))))
            self.location = locations[self.location_index]
        else:
            self.location = location
            self.location_index = self.fips_list.index(location)
        year = int(self.rng.choice(self.valid_years))
        return self.d.episodes[(location, year)], year  # weather of the REQUESTED county (Q8)

    # env.py:133-184
    def reset(self, location=None, similar_climate_counties=None, seed=None, budget=None,
              sample_budget=False, sample_budget_type="less_than"):
        if seed is None:
            seed = np.random.randint(0, 10000)
        self.rng = np.random.default_rng(seed)
        if similar_climate_counties is None:
            similar_climate_counties = self.similar_climate_counties
        if location is None:
            location = str(self.rng.choice(self.fips_list))
        self.ep, year = self._get_episode(location, similar_climate_counties)
        self.year = year
        self.ep_index = location + "_" + str(year)
        self.n_days = self.ep.shape[0]
        self.coef_index = int(self.rng.integers(0, self.n_samples))
        self.actual_alert_buffer = []
        self.attempted_alert_buffer = []
        self.alert_streak = 0
        self.t = 0
        if self.budget is None:
            self.budget = int(self.ep[0, self._i_rem]) if budget is None else budget
        if sample_budget:
            b = self.budget
            if sample_budget_type == "less_than":
                self.budget = int(self.rng.integers(0, b + 1))
            elif sample_budget_type == "centered":
                self.budget = int(self.rng.integers(0.5 * b, 1.5 * b + 1))
        self.remaining_budget = self.budget
        self.at_budget = False
        self.observation = self._get_obs()
        return self.observation, self._get_info()

    # env.py:186-195 -> numeric f64 [29]
    def _get_obs(self):
        row = np.empty(len(self.d.columns) + 1, np.float64)
        row[:-1] = self.ep[self.t]
        row[self._i_lag1] = self.actual_alert_buffer[-1] if self.t > 0 else 0
        row[-1] = sum(self.actual_alert_buffer[-14:])  # 'alert_2wks' (Q1)
        row[self._i_streak] = self.alert_streak
        row[self._i_rem] = self.budget - sum(self.actual_alert_buffer)
        return row

    # env.py:197-226
    def _get_reward(self, action, coef_index=None):
        li = self.location_index
        ci = self.coef_index if coef_index is None else coef_index
        row = self._get_obs()
        s = 0  # Python sum() starts from int 0 and adds left to right in key order
        for j, c in enumerate(self._bcols):
            x = 1.0 if c < 0 else row[c]
            s = s + x * float(self.d.wb[j, ci, li])
        baseline = _expit(s)
        s = 0
        for j, c in enumerate(self._ecols):
            x = 1.0 if c < 0 else row[c]
            s = s + x * float(self.d.we[j, ci, li])
        effectiveness = _expit(s) * (row[self._i_hq] > 0.5)
        reward = float(-1000 / 152 * baseline * (1 - effectiveness * action))
        if action == 1 and self.at_budget:  # dead branch: action is the *actual* action (Q5)
            reward = -1
        return reward

    def _get_info(self):
        return {
            "episode_index": self.ep_index,
            "remaining_budget": self.remaining_budget,
            "at_budget": self.at_budget,
            "feature_names": self.feat_names,
            "location": self.location,
            "location_index": self.location_index,
        }

    # env.py:238-262
    def step(self, action: int):
        self.attempted_alert_buffer.append(action)
        self.at_budget = sum(self.actual_alert_buffer) == self.budget
        actual_action = 0 if (action == 1 and self.at_budget) else action
        self.actual_alert_buffer.append(actual_action)
        if actual_action == 1:
            self.remaining_budget -= 1
        if self.eval_mode:  # _deprecated/env.py:332-342
            posterior_indices = np.arange(self.n_samples)
            reward = float(np.mean([self._get_reward(actual_action, int(i)) for i in posterior_indices]))
        else:
            reward = self._get_reward(actual_action)
        done = self.t >= self.n_days - 1
        if not done:
            self.observation = self._get_obs()
            self.t += 1
            self.alert_streak = self.alert_streak + 1 if actual_action else 0
        return self.observation, reward, done, False, self._get_info()


def numpy_parity_reset_tuple(d: RefData, seed: int, location: str | None, augment: bool,
                             sticky_budget: int | None, budget_kw: int | None, sample_budget: bool,
                             sample_budget_type: str):
    """The draws of env.py:145-177 replayed without an env object. Returns
    (weather_fips, coef_col, year, coef_index, budget, info_location)."""
    rng = np.random.default_rng(seed)
    if location is None:
        location = str(rng.choice(d.fips_list))
    if augment:
        locs = [x for x in similar_counties(location, d.conf_fips, d.conf_zone) if x in d.fips_list]
        li = int(rng.choice(range(len(locs))))
        info_loc = locs[li]
    else:
        li = d.fips_list.index(location)
        info_loc = location
    year = int(rng.choice(d.valid_years))
    ep = d.episodes[(location, year)]
    ci = int(rng.integers(0, d.n_samples))
    b = sticky_budget
    if b is None:
        b = int(ep[0, d.columns.index("remaining_budget")]) if budget_kw is None else budget_kw
    if sample_budget:
        if sample_budget_type == "less_than":
            b = int(rng.integers(0, b + 1))
        elif sample_budget_type == "centered":
            b = int(rng.integers(0.5 * b, 1.5 * b + 1))
    return location, li, year, ci, b, info_loc


# --------------------------------------------------------------------------------------
# Vectorised oracle (float64, same arithmetic and summation order as OracleEnv)
# --------------------------------------------------------------------------------------
class VectorOracle:
    """N independent envs stepped together on dense tables.

    X    f64 [S_w, Y, T, C]   episode rows, C = len(columns) (significance coded)
    wb/we f32 [K, n_samples, S]
    Episode tuple per env: county_w (row of X), year_i, coef_col, sample, budget, n_days.
    """

    def __init__(self, d: RefData, fips_weather: list[str], years: list[int], fixes=(), reward_mode="sampled"):
        # reward_mode="posterior_mean": the legacy eval mode (see OracleEnv.__init__), mean over all posterior draws
        assert reward_mode in ("sampled", "posterior_mean")
        self.reward_mode = reward_mode
        # `fixes`: the build's opt-in corrections of reference quirks (include/w2a.h W2A_FIX_*: "alert_2wks",
        # "lag", "penalty", "obs"); they have no reference counterpart -- the empty default is the pinned,
        # reference-faithful behaviour
        self.fixes = set(fixes)
        self.d = d
        self.fips_weather, self.years = list(fips_weather), [int(y) for y in years]
        T = max(v.shape[0] for v in d.episodes.values())
        C = len(d.columns)
        self.X = np.zeros((len(fips_weather), len(years), T, C), np.float64)
        self.n_days_tab = np.zeros((len(fips_weather), len(years)), np.int64)
        for ci, f in enumerate(fips_weather):
            for yi, y in enumerate(self.years):
                ep = d.episodes.get((f, y))
                if ep is not None:
                    self.X[ci, yi, : ep.shape[0]] = ep
                    self.n_days_tab[ci, yi] = ep.shape[0]
        self.bcols = d.key_columns(d.baseline_keys, "baseline_")
        self.ecols = d.key_columns(d.effectiveness_keys, "effectiveness_")
        self.i_lag1 = d.columns.index("alert_lag1")
        self.i_streak = d.columns.index("alert_streak")
        self.i_rem = d.columns.index("remaining_budget")
        self.i_hq = d.columns.index("heat_qi")
        self.i_h2w = d.columns.index("alerts_2wks") if "alerts_2wks" in d.columns else -1
        self.C = C

    def default_budget(self, county_w, year_i):
        return self.X[county_w, year_i, 0, self.i_rem].astype(np.int64)

    def reset(self, county_w, year_i, coef_col, sample, budget):
        n = len(county_w)
        self.county_w = np.asarray(county_w, np.int64)
        self.year_i = np.asarray(year_i, np.int64)
        self.coef_col = np.asarray(coef_col, np.int64)
        self.sample = np.asarray(sample, np.int64)
        self.budget = np.asarray(budget, np.int64)
        self.n_days = self.n_days_tab[self.county_w, self.year_i]
        self.t = np.zeros(n, np.int64)
        self.used = np.zeros(n, np.int64)
        self.streak = np.zeros(n, np.int64)
        self.hist = np.zeros((n, 14), np.int64)  # last 14 actual actions, newest last
        self.last_actual = np.zeros(n, np.int64)
        self.at_budget = np.zeros(n, np.bool_)
        self.obs = self._get_obs()
        return self.obs.copy()

    def _get_obs(self, lag_src=None):
        n = len(self.t)
        row = np.empty((n, self.C + 1), np.float64)
        row[:, :-1] = self.X[self.county_w, self.year_i, self.t]
        row[:, self.i_lag1] = np.where(self.t > 0, self.last_actual if lag_src is None else lag_src, 0)
        row[:, -1] = self.hist.sum(axis=1)
        if "alert_2wks" in self.fixes and self.i_h2w >= 0:
            row[:, self.i_h2w] = row[:, -1]
        row[:, self.i_streak] = self.streak
        row[:, self.i_rem] = self.budget - self.used
        return row

    def step(self, action):
        action = np.asarray(action, np.int64)
        self.at_budget = self.used == self.budget
        actual = np.where((action == 1) & self.at_budget, 0, action)
        yesterday = self.last_actual
        self.hist = np.concatenate([self.hist[:, 1:], actual[:, None]], axis=1)
        self.last_actual = actual
        self.used = self.used + actual
        row = self._get_obs(yesterday if "lag" in self.fixes else None)
        def reward_for(sample):
            zb = np.zeros(len(action), np.float64)
            for j, c in enumerate(self.bcols):
                x = 1.0 if c < 0 else row[:, c]
                zb = zb + x * self.d.wb[j, sample, self.coef_col].astype(np.float64)
            ze = np.zeros(len(action), np.float64)
            for j, c in enumerate(self.ecols):
                x = 1.0 if c < 0 else row[:, c]
                ze = ze + x * self.d.we[j, sample, self.coef_col].astype(np.float64)
            baseline = _expit(zb)
            eff = _expit(ze) * (row[:, self.i_hq] > 0.5)
            return -1000 / 152 * baseline * (1 - eff * actual)

        if self.reward_mode == "posterior_mean":  # _deprecated/env.py:332-342: np.mean over every posterior index
            reward = np.mean([reward_for(np.full(len(action), i)) for i in range(self.d.n_samples)], axis=0)
        else:
            reward = reward_for(self.sample)
        if "penalty" in self.fixes:
            reward = np.where((action == 1) & self.at_budget, -1.0, reward)
        done = self.t >= self.n_days - 1
        nd = ~done
        if "obs" not in self.fixes:
            self.obs[nd] = row[nd]  # terminal step returns the stale observation (Q6)
        self.t = np.where(nd, self.t + 1, self.t)
        self.streak = np.where(nd, np.where(actual == 1, self.streak + 1, 0), self.streak)
        if "obs" in self.fixes:  # corrected: the next day's row with the advanced state; last row when done
            nxt = self._get_obs()
            self.obs[nd] = nxt[nd]
            self.obs[done] = row[done]
        return self.obs.copy(), reward, done, actual


def oracle_rollout(V: "VectorOracle", policy: dict, n_steps: int, seed_stream=None):
    """Policy loop `a = policy(obs); step(a)` on the vector oracle (the pattern of env.py:265-277), for
    checking w2a_rollout. policy: dict(kind, p, col, threshold, lag, require_budget, table); `seed_stream(i, t)`
    returns the uniform [0,1) draw of env i on day t for the Bernoulli policy. Returns per-env
    (ret, alerts, attempts_over_budget, alert_days bool [n, T])."""
    n = len(V.t)
    T = V.X.shape[2]
    ret = np.zeros(n)
    alerts = np.zeros(n, np.int64)
    over = np.zeros(n, np.int64)
    days = np.zeros((n, T), bool)
    finished = (V.t >= V.n_days - 1) & getattr(V, "_finished", np.zeros(n, bool))
    V._finished = finished.copy()
    for _ in range(n_steps):
        live = ~V._finished
        if not live.any():
            break
        act = _policy_actions(V, policy, seed_stream)
        tday = V.t.copy()
        atb = V.used == V.budget
        _, r, done, actual = V.step(act)
        ret += np.where(live, r, 0.0)
        alerts += np.where(live, actual, 0)
        over += np.where(live & (act == 1) & atb, 1, 0)
        days[np.arange(n)[live & (actual == 1)], tday[live & (actual == 1)]] = True
        V._finished = V._finished | (live & done)
    return ret, alerts, over, days


# --------------------------------------------------------------------------------------
# Restatement of the build's counter-based device RNG (no reference counterpart)
# --------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1
DRAW_COUNTY, DRAW_SIMILAR, DRAW_YEAR, DRAW_SAMPLE, DRAW_BUDGET = 0, 1, 2, 3, 4


def _mix64(z: int) -> int:
    z &= _M64
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & _M64
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & _M64
    z ^= z >> 31
    return z


def devrng_stream(seed: int, env_gid: int, episode_no: int) -> int:
    h = _mix64(seed + 0x9E3779B97F4A7C15 * (env_gid + 1))
    return _mix64(h ^ ((episode_no * 0xBF58476D1CE4E5B9 + 0x94D049BB133111EB) & _M64))


def devrng_bounded(stream: int, slot: int, n: int) -> int:
    """uniform in [0, n): high 32 bits of the slot's word, multiply-shift."""
    u = _mix64(stream + (slot + 1) * 0x9E3779B97F4A7C15) >> 32
    return (u * n) >> 32


def devrng_policy_uniform(policy_seed: int, env_gid: int, episode_no: int, day: int) -> float:
    """The Bernoulli policy's uniform draw of k_rollout for (env, episode, day), as float32 in [0, 1)."""
    st = devrng_stream(policy_seed ^ 0xA5A5A5A55A5A5A5A, env_gid, episode_no)
    u = _mix64(st + (day + 1) * 0x9E3779B97F4A7C15) >> 32
    return float(np.float32(u) * np.float32(2.3283064365386963e-10))


def _mix64_np(z):
    """_mix64 on uint64 arrays (wrap-around arithmetic)."""
    z = z ^ (z >> np.uint64(30))
    z = z * np.uint64(0xBF58476D1CE4E5B9)
    z = z ^ (z >> np.uint64(27))
    z = z * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def devrng_policy_uniform_vec(policy_seed: int, env_gid, episode_no, day) -> np.ndarray:
    """devrng_policy_uniform for arrays of (env id, episode number, day): float32 uniforms in [0, 1)."""
    with np.errstate(over="ignore"):
        gid = np.asarray(env_gid, np.uint64)
        ep = np.asarray(episode_no, np.uint64)
        seed = np.uint64((policy_seed ^ 0xA5A5A5A55A5A5A5A) & _M64)
        h = _mix64_np(seed + np.uint64(0x9E3779B97F4A7C15) * (gid + np.uint64(1)))
        st = _mix64_np(h ^ (ep * np.uint64(0xBF58476D1CE4E5B9) + np.uint64(0x94D049BB133111EB)))
        u = _mix64_np(st + (np.asarray(day, np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(32)
    return u.astype(np.float32) * np.float32(2.3283064365386963e-10)


def devrng_reset_tuple(seed, env_gid, episode_no, S, n_years, n_samples, fips_to_weather, sim_ptr, sim_cnt,
                       augment, default_budget_fn, sticky_budget, budget_kw, sample_mode):
    """Episode tuple the device reset kernel must produce for one env.
    sample_mode: 0 none, 1 less_than, 2 centered. sticky_budget < 0 means unset."""
    st = devrng_stream(seed, env_gid, episode_no)
    county = devrng_bounded(st, DRAW_COUNTY, S)
    coef_col = devrng_bounded(st, DRAW_SIMILAR, int(sim_cnt[county])) if augment else county
    year_i = devrng_bounded(st, DRAW_YEAR, n_years)
    sample = devrng_bounded(st, DRAW_SAMPLE, n_samples)
    cw = int(fips_to_weather[county])
    b = sticky_budget
    if b < 0:
        b = default_budget_fn(cw, year_i) if budget_kw < 0 else budget_kw
    base = b
    if sample_mode == 1:
        b = devrng_bounded(st, DRAW_BUDGET, base + 1)
    elif sample_mode == 2:
        lo = int(0.5 * base)
        hi = int(1.5 * base + 1)
        b = lo + devrng_bounded(st, DRAW_BUDGET, hi - lo)
    return cw, coef_col, year_i, sample, b


# --------------------------------------------------------------------------------------
# Restatement of the reference's SB3 logging callbacks (src/weather2alert/callbacks.py)
# --------------------------------------------------------------------------------------
# The callbacks poll env attributes after every step. They were written against the legacy env and read
# attributes the current env.py no longer has ("stale"); the restatement runs them against the
# CURRENT env's semantics with this mapping, each taken from where the legacy env defined it:
#   env.attempted_alert_buffer  -> same name in env.py:239
#   env.allowed_alert_buffer    -> env.actual_alert_buffer (env.py:248; legacy _deprecated/env.py:329)
#   env.penalize                -> "an alert was attempted at budget on this step" (_deprecated/env.py:324-328)
#   env.cum_reward              -> the running episode return (_deprecated/env.py:343)
#   env.t, env.n_days           -> same names (env.py:157,165,259: t stops at n_days - 1)
#   other_data["y"], ["budget"] -> the episode's year and the env's budget (FinalEvalCallback, callbacks.py:129-130)
# One callback window = one whole episode per env with no reset inside it (evaluation rollouts).
class _EnvView:
    """What the callbacks read from one env (names as in callbacks.py)."""

    def __init__(self, n_days, year, budget):
        self.n_days, self.year, self.budget = int(n_days), int(year), int(budget)
        self.t = 0
        self.attempted_alert_buffer, self.allowed_alert_buffer = [], []
        self.penalize = False
        self.cum_reward = 0.0

    def after_step(self, attempted, actual, at_budget, reward, t_after):
        self.attempted_alert_buffer.append(int(attempted))
        self.allowed_alert_buffer.append(int(actual))
        self.penalize = bool(attempted == 1 and at_budget)
        self.cum_reward += float(reward)
        self.t = int(t_after)


class AlertLoggingOracle:
    """callbacks.py:5-87 (AlertLoggingCallback), line by line, over a list of _EnvView."""

    def __init__(self):  # :8-16
        self.when_alerted, self.streaks = [], []
        self.current_streak = None
        self.last_alert = None
        self.num_over_budget = self.num_alerts = self.num_steps = 0

    def on_step(self, envs, finished=None):  # :18-59; finished[i]: env i's episode is over, it is not polled
        n_envs = len(envs)
        if self.current_streak is None:
            self.last_alert = np.zeros(n_envs, dtype=int)
            self.current_streak = np.zeros(n_envs, dtype=int)
            self.rolled_rewards = np.zeros(n_envs, dtype=float)
            self.a_50 = np.full(n_envs, np.nan)
            self.a_80 = np.full(n_envs, np.nan)
            self.a_100 = np.full(n_envs, np.nan)
        for i, env in enumerate(envs):
            if finished is not None and finished[i]:
                continue
            self.num_steps += 1
            if env.penalize:
                self.num_over_budget += 1
            if env.attempted_alert_buffer:
                prev_alert = self.last_alert[i]
                this_alert = env.attempted_alert_buffer[-1]
                if this_alert:  # alert issued
                    self.when_alerted.append(env.t)
                    self.num_alerts += 1
                    self.current_streak[i] += 1
                elif prev_alert:  # end streak
                    self.streaks.append(self.current_streak[i])
                    self.current_streak[i] = 0
                self.last_alert[i] = this_alert
            if env.t == env.n_days - 2:
                self.rolled_rewards[i] += env.cum_reward
                s = sum(env.allowed_alert_buffer)
                if s > 0:
                    fracs = np.cumsum(env.allowed_alert_buffer) / s
                    for k in range(0, len(fracs)):
                        if np.isnan(self.a_100[i]) and fracs[k] == 1:
                            self.a_100[i] = k
                        if np.isnan(self.a_80[i]) and fracs[k] >= 0.8:
                            self.a_80[i] = k
                        if np.isnan(self.a_50[i]) and fracs[k] >= 0.5:
                            self.a_50[i] = k

    def on_rollout_end(self):  # :61-77
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)  # nanmean of an all-NaN slice
            return {
                "training_rewards": float(np.mean(self.rolled_rewards)),
                "over_budget_freq": self.num_over_budget / self.num_steps,
                "alerts_freq": self.num_alerts / self.num_steps,
                "average_t_alerts": float(np.mean(self.when_alerted)) if self.when_alerted else 0,
                "stdev_t_alerts": float(np.std(self.when_alerted)) if self.when_alerted else 0,
                "average_streak": float(np.mean(self.streaks)) if self.streaks else 0,
                "stdev_streak": float(np.std(self.streaks)) if self.streaks else 0,
                "alert_t_50%": float(np.nanmean(self.a_50)),
                "alert_t_80%": float(np.nanmean(self.a_80)),
                "alert_t_100%": float(np.nanmean(self.a_100)),
            }


CSV_FIELDS = ["year", "alert_budget", "sum_alerts", "reward", "average_t_alerts", "stdev_t_alerts", "average_streak",
              "stdev_streak", "alerts"]  # callbacks.py:136-146, in the order DictWriter gets them (:153-155)


class FinalEvalOracle:
    """callbacks.py:90-157 (FinalEvalCallback) for one eval env and one episode: the row it appends to its CSV."""

    def __init__(self):  # :104-115
        self.year = self.budget = self.sum_alerts = self.reward = 0
        self.alerts, self.when_alerted, self.streaks = [], [], []
        self.current_streak = self.last_alert = 0

    def on_step(self, env):  # :116-133
        prev_alert = self.last_alert
        this_alert = env.allowed_alert_buffer[-1]
        if this_alert:  # alert issued
            self.when_alerted.append(env.t)
            self.current_streak += 1
        elif prev_alert:  # end streak
            self.streaks.append(self.current_streak)
            self.current_streak = 0
        self.last_alert = this_alert
        if env.t == env.n_days - 2:
            self.year = env.year
            self.budget = env.budget
            self.alerts = env.allowed_alert_buffer  # the list object: it keeps growing until the episode ends
            self.sum_alerts = sum(self.alerts)
            self.reward = env.cum_reward

    def row(self):  # :134-146
        return {
            "year": self.year,
            "alert_budget": self.budget,
            "sum_alerts": self.sum_alerts,
            "reward": self.reward,
            "average_t_alerts": float(np.mean(self.when_alerted)) if self.when_alerted else 0,
            "stdev_t_alerts": float(np.std(self.when_alerted)) if self.when_alerted else 0,
            "average_streak": float(np.mean(self.streaks)) if self.streaks else 0,
            "stdev_streak": float(np.std(self.streaks)) if self.streaks else 0,
            "alerts": list(self.alerts),
        }


def oracle_rollout_with_callbacks(V: "VectorOracle", policy: dict, seed_stream=None):
    """One whole episode per env under `policy` (oracle_rollout's policy loop) with both callbacks attached.
    Returns (AlertLoggingOracle summary, list of FinalEvalOracle rows, per-env returns)."""
    n = len(V.t)
    views = [_EnvView(V.n_days[i], V.years[int(V.year_i[i])], V.budget[i]) for i in range(n)]
    log = AlertLoggingOracle()
    finals = [FinalEvalOracle() for _ in range(n)]
    V._finished = np.zeros(n, bool)
    T = V.X.shape[2]
    for _ in range(T):
        live = ~V._finished
        if not live.any():
            break
        tday = V.t.copy()
        atb = V.used == V.budget
        acts = _policy_actions(V, policy, seed_stream)
        _, r, done, actual = V.step(acts)
        for i in range(n):
            if live[i]:
                views[i].after_step(acts[i], actual[i], atb[i], r[i], V.t[i])
                finals[i].on_step(views[i])
        log.on_step(views, finished=~live)
        V._finished = V._finished | (live & done)
        del tday
    return log.on_rollout_end(), [f.row() for f in finals], np.asarray([v.cum_reward for v in views])


def _policy_actions(V, policy, seed_stream):
    n = len(V.t)
    rem = V.budget - V.used
    kind = policy["kind"]
    if kind == "never":
        act = np.zeros(n, np.int64)
    elif kind == "always":
        act = np.ones(n, np.int64)
    elif kind == "bernoulli":
        if hasattr(seed_stream, "vec"):  # vectorised form: seed_stream.vec(day array) -> uniforms of every env
            u = np.asarray(seed_stream.vec(V.t))
        else:
            u = np.asarray([seed_stream(i, int(V.t[i])) for i in range(n)])
        act = (u.astype(np.float32) < np.float32(policy["p"])).astype(np.int64)
    elif kind == "threshold":
        tt = np.where((V.t > 0) & (policy.get("lag", 1) == 1), V.t - 1, V.t)
        feat = V.X[V.county_w, V.year_i, tt, policy["col"]]
        act = (feat.astype(np.float32) > np.float32(policy["threshold"])).astype(np.int64)
    else:
        R = policy["table"].shape[1]
        act = policy["table"][V.t, np.clip(rem, 0, R - 1)].astype(np.int64)
    if policy.get("require_budget"):
        act = np.where(rem <= 0, 0, act)
    return act
