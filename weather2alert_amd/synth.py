"""Schema-faithful synthetic data and posterior weights.

The real HeatAlertsRL data/weights live on the HF hub (reference
``src/weather2alert/env.py:39-67``) and are not reachable offline, so tests and the
benchmark run on synthetic tables that follow the reference ETL's schema and feature
formulas (``data-processing/merge_state_actions.py:121-287``) and the trainer's output
format (``reward-training/train.py:117-137``: ``baseline_<feat>``/``effectiveness_<feat>``
f32 ``[n_samples, 1, S]`` plus ``config.yaml`` with ``fips_list``).

Every table value is rounded to float32 before it is stored (as float64 in parquet,
like the real files), so the float64 reference/oracle and the float32 device tables see
bit-identical inputs.
"""
from __future__ import annotations

import datetime as _dt
import os
from dataclasses import dataclass, field

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(__file__), "data")

# column order of exogenous_states.parquet (merge_state_actions.py:228-248)
EXO_COLS = [
    "heat_qi",
    "heat_qi_above_25",
    "heat_qi_above_75",
    "hi_max",
    "hi_max_above_25",
    "hi_max_above_75",
    "hi_max*heat_qi",
    "hi_max_above_25*heat_qi",
    "hi_max_above_75*heat_qi",
    "heat_qi_3d",
    "excess_heat_3d",
    "excess_heat_3d*heat_qi",
    "heat_qi_7d",
    "excess_heat_7d",
    "excess_heat_7d*heat_qi",
    "weekend",
    "holiday",
    "dos",
    "bspline_dos_0",
    "bspline_dos_1",
    "bspline_dos_2",
]
# column order of endogenous_states_actions.parquet (merge_state_actions.py:264-272)
ENDO_COLS = [
    "alert",
    "alerts_2wks",
    "alert_lag1",
    "alert_streak",
    "remaining_budget",
    "issued_in_advance",
    "significance",
]
EXO_INT_COLS = ("weekend", "holiday", "dos")
SIGNIFICANCE_VALUES = ("A", "W", "Y")  # NWS VTEC significance letters

# sign constraints of weights/linear/config.yaml:4-14
NEGATIVE_BASELINE = ("alert_lag1", "alerts_2wks")
POSITIVE_EFFECTIVENESS = (
    "excess_heat_3d",
    "excess_heat_7d",
    "heat_qi_above_25",
    "heat_qi_above_75",
    "hi_max_above_25",
    "hi_max_above_75",
)


def load_fips_list(name: str = "linear") -> list[str]:
    """County list of a reference weight set (order matters: it is the coefficient column)."""
    with open(os.path.join(DATA_DIR, f"fips_{name}.txt")) as f:
        return [ln.strip() for ln in f if ln.strip()]


def load_ba_zones() -> dict[str, str]:
    """fips -> DoE Building-America climate zone (data/raw/DoE_climate_zones.csv)."""
    out = {}
    with open(os.path.join(DATA_DIR, "ba_zones.csv")) as f:
        next(f)
        for ln in f:
            fips, zone = ln.rstrip("\n").split(",", 1)
            out[fips] = zone
    return out


def _f32(a):
    return np.asarray(a, dtype=np.float32)


def _summer_dates(year: int, n_days: int) -> list[_dt.date]:
    d0 = _dt.date(year, 5, 1)
    return [d0 + _dt.timedelta(days=i) for i in range(n_days)]


def _us_summer_holidays(year: int) -> set[_dt.date]:
    out = set()
    d = _dt.date(year, 5, 31)  # Memorial Day: last Monday of May
    while d.weekday() != 0:
        d -= _dt.timedelta(days=1)
    out.add(d)
    j4 = _dt.date(year, 7, 4)
    out.add(j4)
    if j4.weekday() == 5:
        out.add(j4 - _dt.timedelta(days=1))
    elif j4.weekday() == 6:
        out.add(j4 + _dt.timedelta(days=1))
    d = _dt.date(year, 9, 1)  # Labor Day: first Monday of September
    while d.weekday() != 0:
        d += _dt.timedelta(days=1)
    out.add(d)
    return out


def _rolling_mean(x: np.ndarray, w: int) -> np.ndarray:
    """pandas ``rolling(w, min_periods=1).mean()`` along the last axis."""
    c = np.cumsum(x, axis=-1, dtype=np.float64)
    out = c.copy()
    out[..., w:] = c[..., w:] - c[..., :-w]
    n = np.minimum(np.arange(1, x.shape[-1] + 1), w)
    return out / n


def _rolling_sum(x: np.ndarray, w: int) -> np.ndarray:
    c = np.cumsum(x, axis=-1, dtype=np.float64)
    out = c.copy()
    out[..., w:] = c[..., w:] - c[..., :-w]
    return out


@dataclass
class SynthData:
    """Dense synthetic data set in the reference's logical schema."""

    fips_weather: list[str]  # counties present in the state tables
    years: list[int]
    n_days: int
    exo: np.ndarray  # f32 [S_w, Y, T, 21]  (EXO_COLS order)
    alert: np.ndarray  # bool [S_w, Y, T]
    alerts_2wks: np.ndarray  # f32
    alert_lag1: np.ndarray  # i64
    alert_streak: np.ndarray  # i64
    remaining_budget: np.ndarray  # i64
    issued_in_advance: np.ndarray  # f32
    significance: np.ndarray  # i8 code, 0 = None, k = SIGNIFICANCE_VALUES[k-1]
    fips_list: list[str]  # weight columns (config.yaml fips_list)
    weights: dict[str, np.ndarray]  # name -> f32 [n_samples, 1, S]
    confounder_fips: list[str]
    confounder_zone: list[str]
    meta: dict = field(default_factory=dict)

    @property
    def n_samples(self) -> int:
        return int(self.weights["baseline_bias"].shape[0])


def make_state_arrays(S_w: int, years: list[int], n_days: int, rng: np.random.Generator,
                      alert_rate: float = 0.05, round_f32: bool = True):
    """Dense state tables following merge_state_actions.py:121-210 on synthetic weather. round_f32=False leaves the
    ranks, rolling means, products and standardised splines in float64, as the reference's ETL produces them (the
    real HF parquet is float64 and not float32-representable)."""
    Y, T = len(years), n_days
    L = Y * T
    _f32 = globals()["_f32"] if round_f32 else (lambda a: np.asarray(a, dtype=np.float64))  # noqa: N806
    hi_max = _f32(rng.uniform(0.5, 1.2, size=(S_w, L)))
    # percentile rank per county over its whole series (groupby fips, rank(pct=True))
    order = np.argsort(np.argsort(hi_max, axis=1, kind="stable"), axis=1)
    heat_qi = _f32((order + 1) / float(L))
    hq = heat_qi.astype(np.float64)
    hm = hi_max.astype(np.float64)
    hq25 = _f32((hq > 0.25) * hq)
    hq75 = _f32((hq > 0.75) * hq)
    hm25 = _f32((hm > 25) * hm)  # hi_max is 0.01*F, so never > 25 (reference quirk kept)
    hm75 = _f32((hm > 75) * hm)
    hq3 = _f32(_rolling_mean(hq, 3))
    hq7 = _f32(_rolling_mean(hq, 7))
    ex3 = _f32(np.clip(hq - hq3.astype(np.float64), 0, None))
    ex7 = _f32(np.clip(hq - hq7.astype(np.float64), 0, None))
    dos = np.tile(np.arange(T), Y)
    weekend = np.zeros(L, dtype=np.int64)
    holiday = np.zeros(L, dtype=np.int64)
    for yi, y in enumerate(years):
        dates = _summer_dates(y, T)
        hd = _us_summer_holidays(y)
        for t, d in enumerate(dates):
            weekend[yi * T + t] = int(d.weekday() in (5, 6))
            holiday[yi * T + t] = int(d in hd)
    # patsy bs(dos/M, df=3, degree=3, lower_bound=0, upper_bound=M+1) - 1: cubic Bernstein
    # terms 1..3 on u = (dos/M)/(M+1), then standardised (merge_state_actions.py:199-210)
    M = T - 1
    u = (dos / M) / (M + 1.0)
    bs = np.stack([3 * u * (1 - u) ** 2, 3 * u**2 * (1 - u), u**3], axis=1)
    bs = (bs - bs.mean(axis=0)) / bs.std(axis=0, ddof=1)
    exo = np.empty((S_w, L, len(EXO_COLS)), dtype=np.float32 if round_f32 else np.float64)
    cols = {
        "heat_qi": heat_qi,
        "heat_qi_above_25": hq25,
        "heat_qi_above_75": hq75,
        "hi_max": hi_max,
        "hi_max_above_25": hm25,
        "hi_max_above_75": hm75,
        "hi_max*heat_qi": _f32(hq * hm),
        "hi_max_above_25*heat_qi": _f32(hq25.astype(np.float64) * hm),
        "hi_max_above_75*heat_qi": _f32(hq75.astype(np.float64) * hm),
        "heat_qi_3d": hq3,
        "excess_heat_3d": ex3,
        "excess_heat_3d*heat_qi": _f32(ex3.astype(np.float64) * hq),
        "heat_qi_7d": hq7,
        "excess_heat_7d": ex7,
        "excess_heat_7d*heat_qi": _f32(ex7.astype(np.float64) * hq),
        "weekend": np.broadcast_to(_f32(weekend), (S_w, L)),
        "holiday": np.broadcast_to(_f32(holiday), (S_w, L)),
        "dos": np.broadcast_to(_f32(dos), (S_w, L)),
        "bspline_dos_0": np.broadcast_to(_f32(bs[:, 0]), (S_w, L)),
        "bspline_dos_1": np.broadcast_to(_f32(bs[:, 1]), (S_w, L)),
        "bspline_dos_2": np.broadcast_to(_f32(bs[:, 2]), (S_w, L)),
    }
    for j, c in enumerate(EXO_COLS):
        exo[:, :, j] = cols[c]

    alert = rng.random((S_w, L)) < alert_rate
    a64 = alert.astype(np.float64)
    alerts_2wks = _f32(_rolling_sum(a64, 14))
    alert_lag1 = np.zeros((S_w, L), dtype=np.int64)
    alert_lag1[:, 1:] = alert[:, :-1]
    streak = np.zeros((S_w, L), dtype=np.int64)
    run = np.zeros(S_w, dtype=np.int64)
    for i in range(L):
        run = np.where(alert[:, i], run + 1, 0)
        streak[:, i] = run
    a3 = alert.reshape(S_w, Y, T)
    budget = a3.sum(axis=2, keepdims=True)
    remaining = (budget - np.cumsum(a3, axis=2)).astype(np.int64)
    issued = _f32(np.where(alert, rng.integers(0, 4, size=(S_w, L)), 0))
    signif = np.where(alert, rng.integers(1, len(SIGNIFICANCE_VALUES) + 1, size=(S_w, L)), 0)
    sh = (S_w, Y, T)
    return dict(
        exo=exo.reshape(S_w, Y, T, len(EXO_COLS)),
        alert=alert.reshape(sh),
        alerts_2wks=alerts_2wks.reshape(sh),
        alert_lag1=alert_lag1.reshape(sh),
        alert_streak=streak.reshape(sh),
        remaining_budget=remaining,
        issued_in_advance=issued.reshape(sh),
        significance=signif.astype(np.int8).reshape(sh),
    )


def feature_names() -> list[str]:
    """The 27 reward features: merged columns minus date/fips/year/significance
    (reward-training/modules.py:265,345)."""
    return EXO_COLS + [c for c in ENDO_COLS if c != "significance"]


def make_weights(S: int, n_samples: int, rng: np.random.Generator, scale: dict[str, float] | None = None,
                 sigma: float = 0.3) -> dict[str, np.ndarray]:
    """Posterior-sample tensors f32 [n_samples,1,S] with the linear config's sign
    constraints. ``scale[name]`` divides that feature's coefficients so that each term of
    the logit is O(sigma) (keeps sigmoids off saturation, the sensitive regime for parity)."""
    scale = scale or {}
    out: dict[str, np.ndarray] = {}
    for head in ("baseline", "effectiveness"):
        for name in feature_names():
            z = rng.normal(0.0, sigma, size=(n_samples, 1, S))
            if head == "baseline" and name in NEGATIVE_BASELINE:
                w = -np.exp(z - 1.5)
            elif head == "effectiveness" and name in POSITIVE_EFFECTIVENESS:
                w = np.exp(z - 1.5)
            else:
                w = z
            out[f"{head}_{name}"] = _f32(w / scale.get(name, 1.0))
        loc = -1.0 if head == "baseline" else 0.5
        out[f"{head}_bias"] = _f32(rng.normal(loc, 0.5, size=(n_samples, 1, S)))
    return out


DEFAULT_SCALE = {"dos": 152.0, "remaining_budget": 10.0, "alerts_2wks": 3.0, "alert_streak": 5.0,
                 "issued_in_advance": 3.0, "bspline_dos_0": 2.0, "bspline_dos_1": 2.0, "bspline_dos_2": 3.0}


def make_synth(
    weights_name: str = "linear",
    n_counties_weather: int | None = None,
    n_fips: int | None = None,
    years: list[int] | None = None,
    n_days: int = 153,
    n_samples: int = 100,
    seed: int = 0,
    weight_scale: dict[str, float] | None = DEFAULT_SCALE,
    weight_sigma: float = 0.3,
    extra_confounder_fips: int = 0,
    round_f32: bool = True,
) -> SynthData:
    """Synthetic data set shaped like the reference's.

    ``n_fips`` truncates the weight county list (always keeping '06037' when it is in the
    list), ``n_counties_weather`` the set of counties that have state tables (defaults to all
    weight counties), ``extra_confounder_fips`` adds counties that appear in the confounders
    table but not in ``fips_list`` (exercises the filter at env.py:116).
    """
    rng = np.random.default_rng(seed)
    years = list(range(2006, 2017)) if years is None else list(years)
    full = load_fips_list(weights_name)
    zones = load_ba_zones()
    if n_fips is None or n_fips >= len(full):
        fips_list = full
    else:
        pick = sorted(rng.choice(len(full), size=n_fips, replace=False).tolist())
        if "06037" in full and full.index("06037") not in pick:
            pick[0] = full.index("06037")
            pick = sorted(pick)
        fips_list = [full[i] for i in pick]
    if n_counties_weather is None or n_counties_weather >= len(fips_list):
        fips_weather = list(fips_list)
    else:
        pick = sorted(rng.choice(len(fips_list), size=n_counties_weather, replace=False).tolist())
        if "06037" in fips_list and fips_list.index("06037") not in pick:
            pick[0] = fips_list.index("06037")
            pick = sorted(pick)
        fips_weather = [fips_list[i] for i in pick]
    arrays = make_state_arrays(len(fips_weather), years, n_days, rng, round_f32=round_f32)
    weights = make_weights(len(fips_list), n_samples, rng, weight_scale, weight_sigma)
    # confounders: every weight county + a few outsiders, shuffled (its row order defines the
    # order of the "similar counties" list, datautils.py:124)
    conf = list(fips_list)
    if extra_confounder_fips:
        outsiders = [f for f in sorted(zones) if f not in set(fips_list)]
        idx = rng.choice(len(outsiders), size=extra_confounder_fips, replace=False)
        conf += [outsiders[i] for i in idx]
    perm = rng.permutation(len(conf))
    conf = [conf[i] for i in perm]
    return SynthData(
        fips_weather=fips_weather,
        years=years,
        n_days=n_days,
        fips_list=fips_list,
        weights=weights,
        confounder_fips=conf,
        confounder_zone=[zones.get(f, "Cold") for f in conf],
        meta={"seed": seed, "weights_name": weights_name, "exo_cols": list(EXO_COLS),
              "endo_cols": list(ENDO_COLS), "sig_categories": sorted(SIGNIFICANCE_VALUES)},
        **arrays,
    )


def write_reference_files(data: SynthData, root: str, weights: str = "linear", split: str = "65k",
                          style: str = "plain") -> dict:
    """Write ``data`` in the on-disk layout the reference downloads from the HF hub
    (env.py:40-47,60-67 with ``local_dir=root``):

        root/data/<split>/{exogenous_states,endogenous_states_actions,confounders}.parquet
        root/<weights>/{posterior_samples.safetensors,config.yaml}

    style="plain": frames built in memory, default RangeIndex. style="pandas_etl": what pandas leaves behind when the
    reference's ETL writes the 65k split (data-processing/merge_state_actions.py:249-287): the frames are FILTERED views
    of the all-counties frame (`df[df.fips.isin(confounders_65k.fips)].to_parquet(...)`), so their integer index has gaps
    and travels in the file as `__index_level_0__`; `alert` is bool (:118), `weekend` / `holiday` / `dos` /
    `alert_lag1` / `alert_streak` / `remaining_budget` are int64 (:151,:158,:175,:183,:191,:196), `significance` is an
    object column whose missing values are float NaN from the left merge (:116), stored as nulls; the yaml `fips_list` is
    written unquoted where yaml allows it (weights/linear/config.yaml:16-28: `- 01073` stays a string only because of the
    leading zero + 8/9 or the quotes yaml adds). The real files themselves are unreachable offline; this pins what
    their FORMAT does to the table compiler."""
    import pandas as pd
    import yaml
    from safetensors.numpy import save_file

    S_w, Y, T = data.alert.shape
    ddir = os.path.join(root, "data", split)
    wdir = os.path.join(root, weights)
    os.makedirs(ddir, exist_ok=True)
    os.makedirs(wdir, exist_ok=True)
    fips_col = np.repeat(np.asarray(data.fips_weather, dtype=object), Y * T)
    dates = []
    for y in data.years:
        dates += [d.strftime("%Y-%m-%d") for d in _summer_dates(y, T)]
    date_col = np.tile(np.asarray(dates, dtype=object), S_w)
    exo = {}
    flat = data.exo.reshape(S_w * Y * T, -1)
    for j, c in enumerate(EXO_COLS):
        v = flat[:, j].astype(np.float64)
        exo[c] = v.astype(np.int64) if c in EXO_INT_COLS else v
    exo["fips"] = fips_col
    exo["date"] = date_col
    etl = style == "pandas_etl"
    if style not in ("plain", "pandas_etl"):
        raise ValueError(f"style {style!r}")

    def save(frame: "pd.DataFrame", path: str):
        if not etl:
            frame.to_parquet(path)
            return
        # the all-counties frame holds counties the 65k confounders do not: two made-up ones interleaved here, then
        # filtered away exactly like the ETL does -- the written frame keeps the surviving rows' original index
        n0 = Y * T
        extra = frame.iloc[:n0].copy()
        parts = []
        for k, fake in enumerate(("99001", "99003")):
            e = extra.copy()
            e["fips"] = fake
            parts.append(e)
        cut = (len(frame) // (2 * n0)) * n0
        full = pd.concat([parts[0], frame.iloc[:cut], parts[1], frame.iloc[cut:]], ignore_index=True)
        kept = full[full.fips.isin(set(data.fips_weather))]
        assert len(kept) == len(frame) and not kept.index.equals(pd.RangeIndex(len(kept)))
        kept.to_parquet(path)

    save(pd.DataFrame(exo), os.path.join(ddir, "exogenous_states.parquet"))
    sig = np.asarray([np.nan if etl else None] + list(SIGNIFICANCE_VALUES), dtype=object)[data.significance.reshape(-1)]
    endo = {
        "fips": fips_col,
        "date": date_col,
        "alert": data.alert.reshape(-1).astype(bool) if etl else data.alert.reshape(-1),
        "alerts_2wks": data.alerts_2wks.reshape(-1).astype(np.float64),
        "alert_lag1": data.alert_lag1.reshape(-1).astype(np.int64) if etl else data.alert_lag1.reshape(-1),
        "alert_streak": data.alert_streak.reshape(-1).astype(np.int64) if etl else data.alert_streak.reshape(-1),
        "remaining_budget": data.remaining_budget.reshape(-1).astype(np.int64) if etl else data.remaining_budget.reshape(-1),
        "issued_in_advance": data.issued_in_advance.reshape(-1).astype(np.float64),
        "significance": sig,
    }
    save(pd.DataFrame(endo), os.path.join(ddir, "endogenous_states_actions.parquet"))
    nconf = len(data.confounder_fips)
    crng = np.random.default_rng(12345)
    pd.DataFrame(
        {
            "fips": np.asarray(data.confounder_fips, dtype=object),
            "total_pop": crng.integers(65001, 5_000_000, size=nconf),
            "ba_zone": np.asarray(data.confounder_zone, dtype=object),
            "log_pop_density": crng.normal(5.0, 1.0, size=nconf),
        }
    ).to_parquet(os.path.join(ddir, "confounders.parquet"), index=False)
    save_file({k: np.ascontiguousarray(v) for k, v in data.weights.items()},
              os.path.join(wdir, "posterior_samples.safetensors"))
    with open(os.path.join(wdir, "config.yaml"), "w") as f:
        yaml.dump({"name": weights, "num_samples": data.n_samples, "fips_list": list(data.fips_list)}, f)
    return {"data_dir": root, "weights": weights, "split": split}
