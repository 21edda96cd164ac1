"""Table compiler: the reference's parquet / safetensors / yaml artefacts -> dense HBM tables.

Replaces what ``HeatAlertEnv.__init__`` keeps as pandas/torch objects (reference
``src/weather2alert/env.py:39-105``) by the dense arrays of ``include/w2a.h: w2a_tables``:

    X   f32 [T][S_w*Y][32]      day-major feature rows (one 128-B line per (county, year, day))
    W   f32 [S*n_samples][2][32] baseline / effectiveness coefficient rows in the same slots
    n_days, B0  i32 [S_w*Y]      episode length, default budget (env.py:157,169)
    fips_to_weather, sim_cnt i32 [S]   weight column -> weather county; |similar ∩ fips_list|

The slot layout is data driven: column names come from the files, coefficient slots from
the weight key names exactly as ``env.py:77-82,208,215`` derive them; nothing about the
feature list is hard-coded except the three endogenous fields the env overrides at run time
(``env.py:190-193``).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

ROWF = 32
SLOT_LAG1, SLOT_STREAK, SLOT_REM, SLOT_A2W = 24, 25, 26, 27
SLOT_EXTRA, SLOT_BIAS, SLOT_GATE = 28, 29, 30
RUNTIME_COLS = {"alert_lag1": SLOT_LAG1, "alert_streak": SLOT_STREAK, "remaining_budget": SLOT_REM}
AGENT_2WKS = "alert_2wks"  # the key env.py:191 inserts (not the data column 'alerts_2wks')

WESTERN_STATE_FIPS = frozenset(  # datautils.py:3-17 mapped through FIPS2STATE (:42-100)
    ["03", "04", "06", "08", "16", "30", "35", "32", "41", "53", "38", "46", "31", "20"])


class SchemaError(ValueError):
    pass


def climate_group(fips: str, ba_zone: str) -> str:
    """Zone label used by get_similar_counties (datautils.py:109-120)."""
    if fips[:2] in WESTERN_STATE_FIPS:
        return "Cold-West"
    return "Cold-East" if ba_zone == "Cold" else ba_zone


def resolve_artifact(data_dir: str | None, subfolder: str, filename: str, repo_id: str, repo_type: str) -> str:
    """Local path of one reference artefact, laid out as hf_hub_download(local_dir=data_dir)
    leaves it (env.py:41-47,61-67). Falls back to the hub only if the file is absent."""
    if data_dir is not None:
        p = os.path.join(data_dir, subfolder, filename)
        if os.path.exists(p):
            return p
    try:
        from huggingface_hub import hf_hub_download
    except ImportError as e:  # pragma: no cover
        raise FileNotFoundError(f"{filename} not found under {data_dir!r} and huggingface_hub is unavailable") from e
    return hf_hub_download(repo_id=repo_id, repo_type=repo_type, subfolder=subfolder, filename=filename,
                           local_dir=data_dir)


@dataclass
class CompiledTables:
    columns: list[str]
    fips_weather: list[str]
    years: list[int]
    T: int
    X: np.ndarray
    n_days: np.ndarray
    B0: np.ndarray
    W: np.ndarray
    fips_list: list[str]
    n_samples: int
    fips_to_weather: np.ndarray
    sim_cnt: np.ndarray
    sim_ptr: np.ndarray
    sim_idx: np.ndarray  # fips_list index of each similar county, -1 if outside fips_list (never stored)
    obs_slot: list[int]
    slot_of: dict[str, int]
    baseline_keys: list[str]
    effectiveness_keys: list[str]
    sig_categories: list[str] = field(default_factory=list)
    f32_exact: bool = True  # every table value was exactly representable in float32
    conf_fips: list[str] = field(default_factory=list)   # confounders rows (file order) and their climate group
    conf_group: list[str] = field(default_factory=list)  # (datautils.py:109-120), for counties outside fips_list

    @property
    def feature_names(self) -> list[str]:
        return list(self.columns) + [AGENT_2WKS]

    @property
    def n_obs(self) -> int:
        return len(self.columns) + 1

    @property
    def S(self) -> int:
        return len(self.fips_list)

    @property
    def S_w(self) -> int:
        return len(self.fips_weather)

    @property
    def Y(self) -> int:
        return len(self.years)

    def fips_index(self, fips: str) -> int:
        """fips_list.index(fips) with a cached lookup; ValueError like list.index (env.py:121)."""
        pos = self.__dict__.get("_fips_pos")
        if pos is None:
            pos = self.__dict__["_fips_pos"] = {f: i for i, f in enumerate(self.fips_list)}
        try:
            return pos[fips]
        except KeyError:
            raise ValueError(f"{fips!r} is not in list") from None

    def similar_list(self, county: int) -> np.ndarray:
        return self.sim_idx[self.sim_ptr[county]: self.sim_ptr[county + 1]]

    def similar_for_fips(self, fips: str) -> np.ndarray:
        """fips_list indices of get_similar_counties(fips) ∩ fips_list in confounders order for ANY county of the
        confounders table, inside fips_list or not (env.py:115-116 never looks the requested county up in
        fips_list). KeyError when the county has no confounders row (datautils.py:123)."""
        pos = self.__dict__.get("_fips_pos") or {f: i for i, f in enumerate(self.fips_list)}
        if fips in pos and self.sim_cnt[pos[fips]] > 0:
            return self.similar_list(pos[fips])
        groups = self.__dict__.get("_conf_groups")
        if groups is None:
            first, members = {}, {}
            for f, g in zip(self.conf_fips, self.conf_group):
                first.setdefault(f, g)
                if f in pos:
                    members.setdefault(g, []).append(pos[f])
            groups = self.__dict__["_conf_groups"] = (first, {g: np.asarray(v, np.int32) for g, v in members.items()})
        first, members = groups
        if fips not in first:
            raise KeyError(fips)
        return members.get(first[fips], np.zeros(0, np.int32))

    def nbytes(self) -> int:
        return int(self.X.nbytes + self.W.nbytes + self.n_days.nbytes + self.B0.nbytes)

    _ARRAYS = ("X", "n_days", "B0", "W", "fips_to_weather", "sim_cnt", "sim_ptr", "sim_idx")

    def save_npz(self, path: str) -> None:
        """Cache the compiled tables (skips pandas/pyarrow/safetensors at start-up)."""
        import json

        meta = {k: getattr(self, k) for k in ("columns", "fips_weather", "years", "T", "fips_list", "n_samples",
                                              "obs_slot", "slot_of", "baseline_keys", "effectiveness_keys",
                                              "sig_categories", "f32_exact", "conf_fips", "conf_group")}
        np.savez_compressed(path, meta_json=np.asarray(json.dumps(meta)),
                            **{k: getattr(self, k) for k in self._ARRAYS})

    @classmethod
    def load_npz(cls, path: str) -> "CompiledTables":
        import json

        z = np.load(path)
        meta = json.loads(str(z["meta_json"]))
        return cls(**meta, **{k: z[k] for k in cls._ARRAYS})


# ------------------------------------------------------------------------------------------
def _assign_slots(columns: list[str], weight_names: set[str]) -> tuple[dict[str, int], list[int]]:
    for c in RUNTIME_COLS:
        if c not in columns:
            raise SchemaError(f"state tables lack the endogenous column {c!r} (env.py:190-193)")
    if "heat_qi" not in columns:
        raise SchemaError("state tables lack 'heat_qi' (effectiveness gate, env.py:218)")
    table_cols = [c for c in columns if c not in RUNTIME_COLS]
    feats = [c for c in table_cols if c in weight_names]
    rest = [c for c in table_cols if c not in weight_names]
    ordered = feats + rest
    if len(ordered) > 25:
        raise SchemaError(f"{len(ordered)} table-sourced columns; the 32-float row holds at most 25")
    slot_of = dict(RUNTIME_COLS)
    slot_of[AGENT_2WKS] = SLOT_A2W
    for i, c in enumerate(ordered):
        slot_of[c] = i if i < 24 else SLOT_EXTRA
    slot_of["bias"] = SLOT_BIAS
    obs_slot = [slot_of[c] for c in columns] + [SLOT_A2W]
    return slot_of, obs_slot


def _weight_rows(post: dict[str, np.ndarray], slot_of: dict[str, int], S: int):
    b_keys = [k for k in post if k.startswith("baseline")]  # env.py:77-79
    e_keys = [k for k in post if k.startswith("effectiveness")]  # env.py:80-82
    if "baseline_bias" not in post:
        raise SchemaError("posterior samples lack 'baseline_bias' (env.py:85)")
    n_samples = int(post["baseline_bias"].shape[0])
    W = np.zeros((S, n_samples, 2, ROWF), np.float32)
    for head, (keys, prefix) in enumerate(((b_keys, "baseline_"), (e_keys, "effectiveness_"))):
        for k in keys:
            name = k.replace(prefix, "")  # env.py:208,215
            if name not in slot_of:
                raise SchemaError(f"coefficient {k!r} has no matching column (reference: KeyError at env.py:208)")
            v = np.asarray(post[k], np.float32)
            if v.shape != (n_samples, 1, S):
                raise SchemaError(f"{k}: shape {v.shape}, expected {(n_samples, 1, S)}")
            W[:, :, head, slot_of[name]] += v[:, 0, :].T
    return W.reshape(S * n_samples, 2, ROWF), n_samples, b_keys, e_keys


def _similar_csr(fips_list: list[str], conf_fips: list[str], conf_zone: list[str]):
    """CSR of get_similar_counties(c) ∩ fips_list for every weight county c, keeping the
    confounders' row order (datautils.py:124, env.py:116)."""
    pos = {f: i for i, f in enumerate(fips_list)}
    groups: dict[str, list[int]] = {}
    zone_of = {}
    for f, z in zip(conf_fips, conf_zone):
        g = climate_group(f, z)
        if f not in zone_of:  # .loc on a duplicated index would misbehave; first row wins
            zone_of[f] = g
        if f in pos:
            groups.setdefault(g, []).append(pos[f])
    S = len(fips_list)
    ptr = np.zeros(S + 1, np.int64)
    chunks = []
    for i, f in enumerate(fips_list):
        lst = groups.get(zone_of[f], []) if f in zone_of else []
        chunks.append(np.asarray(lst, np.int32))
        ptr[i + 1] = ptr[i] + len(lst)
    idx = np.concatenate(chunks) if chunks else np.zeros(0, np.int32)
    cnt = np.diff(ptr).astype(np.int32)
    return cnt, ptr, idx


def _finish(columns, fips_weather, years, T, X, n_days, B0, post, fips_list, conf_fips, conf_zone, sig_categories,
            f32_exact, slot_of, obs_slot) -> CompiledTables:
    W, n_samples, b_keys, e_keys = _weight_rows(post, slot_of, len(fips_list))
    wpos = {f: i for i, f in enumerate(fips_weather)}
    f2w = np.asarray([wpos.get(f, -1) for f in fips_list], np.int32)
    cnt, ptr, idx = _similar_csr(fips_list, conf_fips, conf_zone)
    return CompiledTables(
        columns=list(columns), fips_weather=list(fips_weather), years=[int(y) for y in years], T=int(T),
        X=X, n_days=n_days.astype(np.int32), B0=B0.astype(np.int32), W=W, fips_list=list(fips_list),
        n_samples=n_samples, fips_to_weather=f2w, sim_cnt=cnt, sim_ptr=ptr, sim_idx=idx, obs_slot=obs_slot,
        slot_of=slot_of, baseline_keys=b_keys, effectiveness_keys=e_keys, sig_categories=list(sig_categories),
        f32_exact=bool(f32_exact), conf_fips=[str(f) for f in conf_fips],
        conf_group=[climate_group(str(f), str(z)) for f, z in zip(conf_fips, conf_zone)])


def compile_from_files(data_dir: str | None, weights: str = "nn_full_medicare_all", split: str = "65k",
                       years: list | None = None) -> CompiledTables:
    """Load the reference artefacts the way env.py:39-85 does and compile them."""
    import pandas as pd
    import yaml
    from safetensors import safe_open

    paths = {}
    for file in ["confounders", "exogenous_states", "endogenous_states_actions"]:
        paths[file] = resolve_artifact(data_dir, "data/" + split, file + ".parquet",
                                       "mauriciogtec/HeatAlertsRL-Data", "dataset")
    for file in ["posterior_samples.safetensors", "config.yaml"]:
        paths[file] = resolve_artifact(data_dir, weights, file, "mauriciogtec/HeatAlertsRL-Models", "model")
    merged = pd.merge(pd.read_parquet(paths["exogenous_states"]),
                      pd.read_parquet(paths["endogenous_states_actions"]), on=["fips", "date"])
    if not pd.api.types.is_string_dtype(merged["date"]) and not merged["date"].dtype == object:
        # the reference slices the string (env.py:54); a file that carries real timestamps is accepted as well
        merged["date"] = pd.to_datetime(merged["date"]).dt.strftime("%Y-%m-%d")
    merged["year"] = merged.date.str[:4].astype(int)
    conf = pd.read_parquet(paths["confounders"])
    post = {}
    with safe_open(paths["posterior_samples.safetensors"], framework="np") as f:
        for k in f.keys():
            post[k] = f.get_tensor(k)
    with open(paths["config.yaml"]) as fh:
        cfg = yaml.safe_load(fh)
    if "fips_list" not in cfg:
        raise SchemaError(f"{paths['config.yaml']} has no 'fips_list' (reference: KeyError at env.py:75)")
    fips_list = [str(x) for x in cfg["fips_list"]]

    columns = [c for c in merged.columns if c not in ("fips", "date", "year")]
    names = {k.replace("baseline_", "").replace("effectiveness_", "") for k in post}
    slot_of, obs_slot = _assign_slots(columns, names)
    valid_years = [int(y) for y in pd.unique(merged["year"])] if years is None else [int(y) for y in years]
    fips_weather = [str(f) for f in pd.unique(merged["fips"])]
    cats = []
    num = pd.DataFrame(index=merged.index)
    for c in columns:
        col = merged[c]
        # text columns (object dtype with today's pandas, a string dtype with pandas >= 3 / pyarrow-backed frames)
        if not (pd.api.types.is_numeric_dtype(col) or pd.api.types.is_bool_dtype(col)):
            cc = sorted(x for x in col.dropna().unique())
            if c == "significance":
                cats = [str(x) for x in cc]
            lut = {v: float(i + 1) for i, v in enumerate(cc)}
            num[c] = col.map(lambda v: 0.0 if (v is None or v is pd.NA or v != v) else lut[v]).astype(np.float64)
        else:
            num[c] = col.astype(np.float64)
    vals64 = num[columns].values
    vals32 = vals64.astype(np.float32)
    f32_exact = bool(np.array_equal(vals32.astype(np.float64), vals64, equal_nan=True))
    ci = merged["fips"].map({f: i for i, f in enumerate(fips_weather)}).values
    ymap = {y: i for i, y in enumerate(valid_years)}
    yi = merged["year"].map(lambda y: ymap.get(int(y), -1)).values
    day = merged.groupby(["fips", "year"], sort=False).cumcount().values  # file order within an episode
    keep = yi >= 0
    S_w, Y = len(fips_weather), len(valid_years)
    T = int(day[keep].max()) + 1 if keep.any() else 0
    if T == 0:
        raise SchemaError("no rows for the requested years")
    X = np.zeros((T, S_w * Y, ROWF), np.float32)
    r = (ci[keep] * Y + yi[keep]).astype(np.int64)
    d = day[keep]
    for j, c in enumerate(columns):
        if c in RUNTIME_COLS:
            continue
        X[d, r, slot_of[c]] = vals32[keep, j]
    X[d, r, SLOT_BIAS] = 1.0
    # the gate heat_qi > 0.5 (env.py:218) is decided on the file's float64 values, so a value that float32
    # rounding would move across 0.5 cannot flip it; the kernels test this 0/1 flag with "> 0.5f"
    X[d, r, SLOT_GATE] = (vals64[keep, columns.index("heat_qi")] > 0.5).astype(np.float32)
    n_days = np.zeros(S_w * Y, np.int64)
    np.add.at(n_days, r, 1)
    B0 = np.zeros(S_w * Y, np.int64)
    first = d == 0
    B0[r[first]] = vals64[keep, columns.index("remaining_budget")][first].astype(np.int64)
    return _finish(columns, fips_weather, valid_years, T, X, n_days, B0, post, fips_list,
                   [str(x) for x in conf["fips"]], [str(x) for x in conf["ba_zone"]], cats, f32_exact, slot_of,
                   obs_slot)


def compile_from_synth(d, sorted_keys: bool = True) -> CompiledTables:
    """Same tables straight from a dense ``synth.SynthData`` (no parquet round trip)."""
    exo_cols, endo_cols = d.meta["exo_cols"], d.meta["endo_cols"]
    columns = list(exo_cols) + list(endo_cols)
    keys = sorted(d.weights) if sorted_keys else list(d.weights)  # safetensors lists keys sorted
    post = {k: d.weights[k] for k in keys}
    names = {k.replace("baseline_", "").replace("effectiveness_", "") for k in post}
    slot_of, obs_slot = _assign_slots(columns, names)
    S_w, Y, T = d.alert.shape
    X = np.zeros((T, S_w * Y, ROWF), np.float32)
    Xv = X.reshape(T, S_w, Y, ROWF)
    for j, c in enumerate(exo_cols):
        Xv[..., slot_of[c]] = np.moveaxis(d.exo[..., j], 2, 0)
    for c in endo_cols:
        if c in RUNTIME_COLS:
            continue
        Xv[..., slot_of[c]] = np.moveaxis(np.asarray(getattr(d, c), np.float32), 2, 0)
    Xv[..., SLOT_BIAS] = 1.0
    # 0/1 gate flag (env.py:218), decided on the data's own (possibly float64) value, not on its float32 copy
    Xv[..., SLOT_GATE] = np.moveaxis(d.exo[..., list(exo_cols).index("heat_qi")] > 0.5, 2, 0).astype(np.float32)
    f32_exact = bool(np.array_equal(d.exo.astype(np.float32).astype(np.float64), d.exo.astype(np.float64)) and all(
        np.array_equal(np.asarray(getattr(d, c), np.float32).astype(np.float64), np.asarray(getattr(d, c), np.float64))
        for c in endo_cols if c not in RUNTIME_COLS))
    n_days = np.full(S_w * Y, T, np.int64)
    ragged = d.meta.get("n_days_per_episode")  # optional [S_w, Y] episode lengths (0 = pair absent)
    if ragged is not None:
        n_days = np.asarray(ragged, np.int64).reshape(-1)
        dead = np.arange(T)[:, None] >= n_days[None, :]
        X[dead] = 0.0
    B0 = np.where(n_days > 0, d.remaining_budget[:, :, 0].reshape(-1), 0).astype(np.int64)
    return _finish(columns, d.fips_weather, d.years, T, X, n_days, B0, post, d.fips_list, d.confounder_fips,
                   d.confounder_zone, d.meta.get("sig_categories", []), f32_exact, slot_of, obs_slot)


# ------------------------------------------------------------------------------------------
class DeviceTables:
    """CompiledTables resident in HBM (torch owns the memory) + the w2a_tables struct."""

    def __init__(self, ct: CompiledTables, device):
        import torch

        from . import _ffi

        self.ct = ct
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DeviceTables needs a ROCm GPU device ('cuda:N'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)  # noqa: E731
        self.X = t(ct.X)
        self.W = t(ct.W)
        self.n_days = t(ct.n_days)
        self.B0 = t(ct.B0)
        self.fips_to_weather = t(ct.fips_to_weather)
        self.sim_cnt = t(ct.sim_cnt)
        s = _ffi.Tables()
        s.X, s.W = self.X.data_ptr(), self.W.data_ptr()
        s.n_days, s.B0 = self.n_days.data_ptr(), self.B0.data_ptr()
        s.fips_to_weather, s.sim_cnt = self.fips_to_weather.data_ptr(), self.sim_cnt.data_ptr()
        s.T, s.S_w, s.Y, s.S, s.n_samples, s.n_obs = ct.T, ct.S_w, ct.Y, ct.S, ct.n_samples, ct.n_obs
        for j in range(_ffi.ROW_FLOATS):
            s.obs_slot[j] = ct.obs_slot[j] if j < ct.n_obs else -1
        s.slot_heat_qi = ct.slot_of["heat_qi"]
        self.sim_ptr = t(ct.sim_ptr.astype(np.int32))
        self.sim_idx = t(ct.sim_idx.astype(np.int32) if len(ct.sim_idx) else np.zeros(1, np.int32))
        s.sim_ptr, s.sim_idx = self.sim_ptr.data_ptr(), self.sim_idx.data_ptr()
        s.slot_alerts_2wks = ct.slot_of.get("alerts_2wks", -1)
        # gate bitmap [T][ceil(R / 32)]: bit (row & 31) of word row >> 5 = the slot-30 flag of (day, row)
        R = ct.S_w * ct.Y
        gw = (R + 31) // 32
        g = np.zeros((ct.T, gw * 32), np.uint8)
        g[:, :R] = ct.X[:, :, SLOT_GATE] > 0.5
        bits = np.packbits(g.reshape(ct.T, gw, 32), axis=2, bitorder="little").view(np.uint32).reshape(ct.T, gw)
        self.gate_bits = t(bits)
        s.gate_bits, s.gate_words = self.gate_bits.data_ptr(), gw
        if os.environ.get("W2A_NO_GATE_BITS"):  # A/B: every alerting env fetches its effectiveness row (no bitmap lookup)
            s.gate_bits, s.gate_words = None, 0
        self.struct = s
