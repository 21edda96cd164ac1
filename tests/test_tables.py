"""Host logic: table compiler (slot layout, day-major rows, weight rows, similar-county CSR)
checked against the oracle's independent loader on the committed mini data set."""
import os

import numpy as np
import pytest

from oracle import heatalert_oracle as O
from weather2alert_amd import synth, tables


@pytest.fixture(scope="module")
def ct(mini_root):
    return tables.compile_from_files(mini_root, "linear")


@pytest.fixture(scope="module")
def rd(mini_root):
    return O.RefData.from_files(mini_root, "linear")


def test_schema(ct, rd):
    assert ct.columns == rd.columns and ct.n_obs == 29
    assert ct.fips_list == rd.fips_list and ct.years == rd.valid_years and ct.n_samples == rd.n_samples
    assert ct.baseline_keys == rd.baseline_keys and ct.effectiveness_keys == rd.effectiveness_keys
    assert ct.sig_categories == rd.sig_categories and ct.f32_exact
    assert sorted(ct.obs_slot) == sorted(set(ct.obs_slot)) and max(ct.obs_slot) < 32
    assert [ct.slot_of[c] for c in ("alert_lag1", "alert_streak", "remaining_budget", "alert_2wks")] == [24, 25, 26, 27]
    assert ct.slot_of["bias"] == 29


def test_rows_equal_reference_frame(ct, rd):
    rt = [c for c in ct.columns if c not in tables.RUNTIME_COLS]
    for (f, y), ep in rd.episodes.items():
        r = ct.fips_weather.index(f) * ct.Y + ct.years.index(y)
        assert ct.n_days[r] == ep.shape[0] == 153
        assert ct.B0[r] == ep[0, rd.columns.index("remaining_budget")]
        for c in rt:
            np.testing.assert_array_equal(ct.X[:153, r, ct.slot_of[c]], ep[:, rd.columns.index(c)].astype(np.float32))
        assert (ct.X[:153, r, 29] == 1).all()
        np.testing.assert_array_equal(ct.X[:153, r, 30], (ep[:, rd.columns.index("heat_qi")] > 0.5).astype(np.float32))
        assert (ct.X[:153, r, 24:28] == 0).all() and (ct.X[:153, r, 31] == 0).all()


def test_weight_rows(ct, rd):
    W = ct.W.reshape(ct.S, ct.n_samples, 2, 32)
    for head, (keys, w, prefix) in enumerate(((rd.baseline_keys, rd.wb, "baseline_"),
                                               (rd.effectiveness_keys, rd.we, "effectiveness_"))):
        used = set()
        for j, k in enumerate(keys):
            s = ct.slot_of[k.replace(prefix, "")]
            used.add(s)
            np.testing.assert_array_equal(W[:, :, head, s], w[j].T)
        rest = [s for s in range(32) if s not in used]
        assert (W[:, :, head, rest] == 0).all()
        assert 30 in rest and 27 in rest  # gate copy and the agent's alert_2wks carry no coefficient (Q1)


def test_similar_csr_matches_datautils_restatement(ct, rd):
    for i, f in enumerate(ct.fips_list):
        ref = [x for x in O.similar_counties(f, rd.conf_fips, rd.conf_zone) if x in rd.fips_list]
        assert ct.sim_cnt[i] == len(ref)
        assert [ct.fips_list[j] for j in ct.similar_list(i)] == ref


def test_file_and_dense_paths_agree(ct):
    mini = synth.make_synth("linear", n_fips=24, years=[2006, 2007, 2008], n_samples=8, seed=7,
                            extra_confounder_fips=8)
    c2 = tables.compile_from_synth(mini)
    for k in ("X", "W", "n_days", "B0", "fips_to_weather", "sim_cnt", "sim_idx"):
        np.testing.assert_array_equal(getattr(ct, k), getattr(c2, k))
    assert ct.obs_slot == c2.obs_slot and ct.fips_weather == c2.fips_weather


def test_compiled_fixture_is_current(ct, golden_dir):
    import os

    c2 = tables.CompiledTables.load_npz(os.path.join(golden_dir, "mini_compiled.npz"))
    for k in tables.CompiledTables._ARRAYS:
        np.testing.assert_array_equal(getattr(ct, k), getattr(c2, k))
    for k in ("columns", "fips_weather", "years", "T", "fips_list", "n_samples", "obs_slot", "slot_of",
              "baseline_keys", "effectiveness_keys", "sig_categories", "f32_exact"):
        assert getattr(ct, k) == getattr(c2, k), k


def test_schema_errors():
    sd = synth.make_synth("linear", n_fips=8, years=[2006], n_samples=2, seed=0)
    bad = dict(sd.weights)
    bad["baseline_not_a_column"] = bad["baseline_bias"]
    sd.weights = bad
    with pytest.raises(tables.SchemaError):
        tables.compile_from_synth(sd)


def test_full_size_similar_anchor():
    """n_similar('06037') = 111 on the 746-county list with the real climate zones."""
    full = synth.load_fips_list("linear")
    zones = synth.load_ba_zones()
    cnt, ptr, idx = tables._similar_csr(full, full, [zones[f] for f in full])
    assert full.index("06037") == 84 and cnt[84] == 111
    assert synth.load_fips_list("nn_full_medicare_all").index("06037") == 82


def test_ragged_real_artefacts(tmp_path, mini_root):
    """Real-artefact shapes the reference would meet: episodes of different length, a
    (county, year) pair missing altogether, rows in shuffled file order, years restricted by the ctor."""
    import shutil

    import pandas as pd

    root = tmp_path / "ragged"
    shutil.copytree(mini_root, root)
    ddir = root / "data" / "65k"
    exo = pd.read_parquet(ddir / "exogenous_states.parquet")
    endo = pd.read_parquet(ddir / "endogenous_states_actions.parquet")
    f0, f1, f2 = sorted(exo.fips.unique())[:3]
    drop = ((exo.fips == f0) & (exo.date >= "2006-09-11")) | ((exo.fips == f1) & (exo.date.str[:4] == "2007")) | \
           ((exo.fips == f2) & (exo.date.str[:7] == "2008-05"))
    exo = exo[~drop].sample(frac=1.0, random_state=1)  # shuffled rows: episode order = file order within a pair
    exo.to_parquet(ddir / "exogenous_states.parquet")
    ct = tables.compile_from_files(str(root), "linear")
    rd = O.RefData.from_files(str(root), "linear")
    assert ct.years == rd.valid_years and ct.fips_weather == [f for f in pd.unique(exo.fips)]
    lens = set()
    for ci, f in enumerate(ct.fips_weather):
        for yi, y in enumerate(ct.years):
            r = ci * ct.Y + yi
            ep = rd.episodes.get((f, y))
            if ep is None:
                assert ct.n_days[r] == 0
                continue
            assert ct.n_days[r] == ep.shape[0]
            lens.add(int(ct.n_days[r]))
            assert ct.B0[r] == ep[0, rd.columns.index("remaining_budget")]
            np.testing.assert_array_equal(ct.X[: ep.shape[0], r, ct.slot_of["dos"]],
                                          ep[:, rd.columns.index("dos")].astype(np.float32))
            assert (ct.X[ep.shape[0]:, r] == 0).all()
    r1 = ct.fips_weather.index(f1) * ct.Y + ct.years.index(2007)
    assert ct.n_days[r1] == 0 and lens == {153, 133, 122}
    # years=[...] restricts valid_years like the ctor argument (env.py:32,104)
    ct2 = tables.compile_from_files(str(root), "linear", years=[2008, 2006])
    assert ct2.years == [2008, 2006] and ct2.Y == 2


def _drop_feature(sd, name):
    j = sd.meta["exo_cols"].index(name)
    sd.exo = np.delete(sd.exo, j, axis=3)
    sd.meta["exo_cols"] = [c for c in sd.meta["exo_cols"] if c != name]
    sd.weights = {k: v for k, v in sd.weights.items() if k not in (f"baseline_{name}", f"effectiveness_{name}")}
    return sd


def test_slot_layout_is_data_driven():
    """Nothing about the feature list is hard-coded: a schema without 'holiday' compiles to n_obs = 28 with the
    remaining columns in file order; a schema with too many table-sourced columns is refused."""
    sd = _drop_feature(synth.make_synth("linear", n_fips=8, years=[2006], n_samples=2, seed=0), "holiday")
    ct = tables.compile_from_synth(sd)
    assert ct.n_obs == 28 and "holiday" not in ct.columns and ct.feature_names[-1] == "alert_2wks"
    assert [ct.slot_of[c] for c in ct.columns[:20]] == list(range(20))
    sd2 = synth.make_synth("linear", n_fips=8, years=[2006], n_samples=2, seed=0)
    sd2.exo = np.concatenate([sd2.exo, sd2.exo[..., :1]], axis=3)
    sd2.meta["exo_cols"] = sd2.meta["exo_cols"] + ["extra_feature"]
    with pytest.raises(tables.SchemaError):
        tables.compile_from_synth(sd2)


def test_pandas_etl_file_format_compiles_to_the_same_tables(tmp_path):
    """What pandas actually leaves on disk when the reference's ETL writes the 65k split
    (data-processing/merge_state_actions.py:249-287: filtered frames -> a gapped integer index stored as
    `__index_level_0__`; bool `alert`; int64 counters; `significance` with NaN for "no alert"; 14 years, 2006-2019 as in
    data-processing/conf/config.yaml:8-9) must compile to exactly the tables of the plain in-memory layout -- and to the
    tables compiled straight from the synthetic arrays. The real HF files cannot be fetched here; their format can."""
    import pandas as pd
    import pyarrow.parquet as pq

    sd = synth.make_synth("linear", n_fips=9, years=list(range(2006, 2020)), n_samples=3, seed=8, extra_confounder_fips=2)
    plain, etl = tmp_path / "plain", tmp_path / "etl"
    synth.write_reference_files(sd, str(plain), "linear")
    synth.write_reference_files(sd, str(etl), "linear", style="pandas_etl")
    for name in ("exogenous_states", "endogenous_states_actions"):
        cols = pq.read_schema(etl / "data" / "65k" / f"{name}.parquet").names
        assert "__index_level_0__" in cols, (name, cols)  # the filtered frame's index really travelled in the file
        assert "__index_level_0__" not in pq.read_schema(plain / "data" / "65k" / f"{name}.parquet").names
    endo = pd.read_parquet(etl / "data" / "65k" / "endogenous_states_actions.parquet")
    assert endo["alert"].dtype == bool and endo["remaining_budget"].dtype == np.int64 and endo["alert_streak"].dtype == np.int64
    assert endo["significance"].isna().any() and not endo.index.equals(pd.RangeIndex(len(endo)))
    a = tables.compile_from_files(str(plain), "linear")
    b = tables.compile_from_files(str(etl), "linear")
    c = tables.compile_from_synth(sd)
    assert a.years == b.years == list(range(2006, 2020)) and a.Y == 14
    for other in (b, c):
        assert a.columns == other.columns and a.fips_weather == other.fips_weather and a.fips_list == other.fips_list
        assert a.sig_categories == other.sig_categories and a.obs_slot == other.obs_slot and a.slot_of == other.slot_of
        for k in tables.CompiledTables._ARRAYS:
            np.testing.assert_array_equal(getattr(a, k), getattr(other, k), err_msg=k)
    # the oracle's loader (the reference's own pandas calls, env.py:49-56) reads the same episodes from both layouts
    ra, rb = O.RefData.from_files(str(plain), "linear"), O.RefData.from_files(str(etl), "linear")
    assert ra.columns == rb.columns and ra.episodes.keys() == rb.episodes.keys()
    for k in ra.episodes:
        np.testing.assert_array_equal(ra.episodes[k], rb.episodes[k])


def test_string_dtype_and_timestamp_columns_are_accepted(tmp_path):
    """Frames whose text columns are not `object` (pyarrow-backed strings: the default of newer pandas) and whose `date`
    is a real timestamp compile to the same tables."""
    import pandas as pd

    sd = synth.make_synth("linear", n_fips=5, years=[2006, 2007], n_samples=2, seed=3)
    root = tmp_path / "r"
    synth.write_reference_files(sd, str(root), "linear")
    ref = tables.compile_from_files(str(root), "linear")
    ddir = root / "data" / "65k"
    for name in ("exogenous_states", "endogenous_states_actions"):
        df = pd.read_parquet(ddir / f"{name}.parquet")
        df["fips"] = df["fips"].astype("string[pyarrow]")
        df["date"] = pd.to_datetime(df["date"])
        if "significance" in df:
            df["significance"] = df["significance"].astype("string[pyarrow]")
        df.to_parquet(ddir / f"{name}.parquet")
    got = tables.compile_from_files(str(root), "linear")
    assert got.sig_categories == ref.sig_categories and got.fips_weather == ref.fips_weather
    for k in tables.CompiledTables._ARRAYS:
        np.testing.assert_array_equal(getattr(got, k), getattr(ref, k), err_msg=k)


def test_missing_files_come_from_the_hub_like_the_reference(tmp_path, monkeypatch):
    """resolve_artifact's fallback (tables.py): a file absent under data_dir is fetched with the very arguments the
    reference passes to hf_hub_download (env.py:40-47: repo mauriciogtec/HeatAlertsRL-Data, repo_type "dataset",
    subfolder "data/<split>"; env.py:60-67: repo mauriciogtec/HeatAlertsRL-Models, repo_type "model",
    subfolder = weights; local_dir = data_dir). The hub itself is a stand-in (no network here) that serves from a
    directory and records its calls."""
    import shutil
    import sys
    import types

    sd = synth.make_synth("linear", n_fips=6, years=[2006], n_samples=2, seed=5)
    hub, local = tmp_path / "hub", tmp_path / "local"
    synth.write_reference_files(sd, str(hub), "linear")
    want = tables.compile_from_files(str(hub), "linear")
    calls = []

    def hf_hub_download(repo_id, filename, subfolder=None, repo_type=None, local_dir=None, **kw):
        calls.append((repo_id, repo_type, subfolder, filename, local_dir))
        dst = os.path.join(local_dir, subfolder, filename)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copy(os.path.join(str(hub), subfolder, filename), dst)
        return dst

    monkeypatch.setitem(sys.modules, "huggingface_hub", types.SimpleNamespace(hf_hub_download=hf_hub_download))
    local.mkdir()
    # one file is already there (a partial download): only the missing ones are fetched
    os.makedirs(local / "data" / "65k")
    shutil.copy(hub / "data" / "65k" / "confounders.parquet", local / "data" / "65k" / "confounders.parquet")
    got = tables.compile_from_files(str(local), "linear")
    for k in tables.CompiledTables._ARRAYS:
        np.testing.assert_array_equal(getattr(got, k), getattr(want, k), err_msg=k)
    assert sorted(calls) == sorted([
        ("mauriciogtec/HeatAlertsRL-Data", "dataset", "data/65k", "exogenous_states.parquet", str(local)),
        ("mauriciogtec/HeatAlertsRL-Data", "dataset", "data/65k", "endogenous_states_actions.parquet", str(local)),
        ("mauriciogtec/HeatAlertsRL-Models", "model", "linear", "posterior_samples.safetensors", str(local)),
        ("mauriciogtec/HeatAlertsRL-Models", "model", "linear", "config.yaml", str(local))])
    calls.clear()
    tables.compile_from_files(str(local), "linear")  # second time: everything is local, the hub is not asked
    assert calls == []
