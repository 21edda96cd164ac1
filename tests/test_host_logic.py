"""Host logic on CPU: the NumPy-parity replay of reset() (weather2alert_amd/rng.py) against the golden
reset tuples captured from the reference, and the errors it mirrors."""
import json
import os

import numpy as np
import pytest

from weather2alert_amd import rng, tables


@pytest.fixture(scope="module")
def gold(golden_dir):
    d = dict(np.load(os.path.join(golden_dir, "mini_traj.npz")))
    ct = tables.CompiledTables.load_npz(os.path.join(golden_dir, "mini_compiled.npz"))
    return d, json.loads(str(d["meta_json"])), ct


def test_replay_matches_every_golden_reset(gold):
    d, meta, ct = gold
    sticky = {}
    for i, e in enumerate(meta["episodes"]):
        key, kw = e["env_key"], e["reset"]
        sticky.setdefault(key, e["ctor"].get("budget"))
        aug = kw.get("similar_climate_counties", e["ctor"].get("similar_climate_counties", False))
        w, y_i, li, ci, b, info_loc = rng.numpy_parity_episode(
            ct, kw["seed"], kw.get("location"), bool(aug), sticky[key], kw.get("budget"),
            kw.get("sample_budget", False), kw.get("sample_budget_type", "less_than"))
        sticky[key] = b  # self.budget keeps the (sampled) value (Q9)
        assert f"{ct.fips_weather[w]}_{ct.years[y_i]}" == e["episode_index"]
        assert (li, ci, b, info_loc) == (d["location_index"][i], d["coef_index"][i], d["budget"][i], e["info_location"])


def test_replay_errors_mirror_the_reference(gold):
    _, _, ct = gold
    with pytest.raises(ValueError):
        rng.numpy_parity_episode(ct, 0, "99999", False, None, None, False, "less_than")
    import copy

    ct2 = copy.copy(ct)
    ct2.n_days = ct.n_days.copy()
    ct2.n_days[:] = 0
    with pytest.raises(KeyError):
        rng.numpy_parity_episode(ct2, 0, ct.fips_list[0], False, None, None, False, "less_than")
    ct3 = copy.copy(ct)  # a county without a confounders row: confounders.loc[fips] (datautils.py:123)
    ct3.__dict__.pop("_conf_groups", None)
    ct3.sim_cnt = np.zeros_like(ct.sim_cnt)
    ct3.conf_fips, ct3.conf_group = [], []
    with pytest.raises(KeyError):
        rng.numpy_parity_episode(ct3, 0, ct.fips_list[0], True, None, None, False, "less_than")


def test_augmentation_of_a_county_outside_fips_list():
    """env.py:115-127 with similar_climate_counties=True never looks the requested county up in fips_list: a county
    that only has weather and a confounders row still resets (coefficients of a drawn similar county, its own
    weather), while without augmentation list.index raises ValueError (env.py:121). Checked against the oracle's
    restatement of the same draws."""
    from oracle import heatalert_oracle as O
    from weather2alert_amd import synth

    sd = synth.make_synth("linear", n_fips=30, years=[2006, 2007, 2008], n_samples=7, seed=4, extra_confounder_fips=6)
    outsider = next(f for f in sd.confounder_fips if f not in sd.fips_list)
    sd.fips_weather[5] = outsider  # that county's state tables now belong to a county outside fips_list
    ct = tables.compile_from_synth(sd)
    rd = O.RefData.from_synth(sd)
    assert outsider not in ct.fips_list and ct.fips_weather[5] == outsider
    for seed in range(12):
        w, y_i, li, ci, b, info = rng.numpy_parity_episode(ct, seed, outsider, True, None, None, seed % 2 == 1,
                                                            "centered")
        loc_o, li_o, year_o, ci_o, b_o, info_o = O.numpy_parity_reset_tuple(rd, seed, outsider, True, None, None,
                                                                           seed % 2 == 1, "centered")
        assert (ct.fips_weather[w], ct.years[y_i], li, ci, b, info) == (loc_o, year_o, li_o, ci_o, b_o, info_o)
        assert w == 5 and info in ct.fips_list
    with pytest.raises(ValueError):
        rng.numpy_parity_episode(ct, 0, outsider, False, None, None, False, "less_than")


def test_corrected_augmentation_uses_the_drawn_county(gold):
    _, _, ct = gold
    loc = "06037"
    w0, _, li0, _, _, info0 = rng.numpy_parity_episode(ct, 21, loc, True, None, None, False, "less_than")
    w1, _, li1, _, _, info1 = rng.numpy_parity_episode(ct, 21, loc, True, None, None, False, "less_than", True)
    assert info0 == info1 and ct.fips_weather[w0] == loc  # faithful: weather of the requested county (Q8)
    assert ct.fips_list[li1] == info1 and ct.fips_weather[w1] == info1 and li0 < ct.sim_cnt[ct.fips_list.index(loc)]


def test_vector_env_is_a_gymnasium_vector_env_when_gymnasium_is_importable(monkeypatch):
    """HeatAlertVecEnv subclasses gymnasium.vector.VectorEnv when the package can be imported (it is absent from
    the build image, so a stand-in module is injected) and carries the VectorEnv attributes either way."""
    import importlib
    import sys
    import types

    import enum

    class FakeVectorEnv:
        metadata: dict = {}

    class FakeAutoresetMode(enum.Enum):
        NEXT_STEP = "NextStep"
        SAME_STEP = "SameStep"
        DISABLED = "Disabled"

    gym = types.ModuleType("gymnasium")
    vec = types.ModuleType("gymnasium.vector")
    vec.VectorEnv = FakeVectorEnv
    vec.AutoresetMode = FakeAutoresetMode
    gym.vector = vec
    monkeypatch.setitem(sys.modules, "gymnasium", gym)
    monkeypatch.setitem(sys.modules, "gymnasium.vector", vec)
    import weather2alert_amd.env as envmod

    try:
        m = importlib.reload(envmod)
        assert issubclass(m.HeatAlertVecEnv, FakeVectorEnv)
        cls = m.HeatAlertVecEnv
        assert cls.metadata["autoreset_mode"] is FakeAutoresetMode.SAME_STEP and cls.spec is None and cls.render_mode is None
        assert m._autoreset_metadata("disabled") is FakeAutoresetMode.DISABLED
        assert m._autoreset_metadata("next_step") is FakeAutoresetMode.NEXT_STEP
        for attr in ("np_random", "np_random_seed", "unwrapped"):
            assert isinstance(getattr(cls, attr), property)
        for meth in ("reset", "step", "close"):
            assert callable(getattr(cls, meth))
    finally:
        monkeypatch.delitem(sys.modules, "gymnasium")
        monkeypatch.delitem(sys.modules, "gymnasium.vector")
        importlib.reload(envmod)
    assert envmod.HeatAlertVecEnv.__mro__[1] is object
    assert envmod.HeatAlertVecEnv.metadata["autoreset_mode"] == "same_step" and envmod._autoreset_metadata("disabled") == "disabled"


def test_kernel_options_record_and_overrides():
    """The switches that select HOW things are computed live in one record; each is still accepted by name."""
    from weather2alert_amd import KernelOptions

    k = KernelOptions()
    assert (k.step_kernel, k.write_obs, k.rollout_order, k.rollout_mfma, k.pm_kernel) == ("auto", True, True, True, "matrix_i8")
    k2 = k.with_overrides(step_kernel="wide", pm_kernel="vector")
    assert k2.step_kernel == "wide" and k2.pm_kernel == "vector" and k.step_kernel == "auto"  # frozen: a new record
    with pytest.raises(ValueError, match="removed"):
        KernelOptions(reward_path="table")
    with pytest.raises(ValueError):
        k.with_overrides(step_kernel="fastest")
    with pytest.raises(TypeError):
        k.with_overrides(step_kernal="wide")


def test_library_is_rebuilt_on_content_not_on_file_times(tmp_path, monkeypatch):
    """needs_build() compares a hash of the sources with the sidecar build_lib() wrote: touching a source (a fresh
    checkout, a copy) does not ask for a rebuild, changing one does; a library without a sidecar falls back to file
    times."""
    import os
    import time

    from weather2alert_amd import build as b

    if not os.path.exists(b.LIB) or not os.path.exists(b.HASH_FILE):
        pytest.skip("library not built in this tree")
    assert not b.needs_build()
    src = b._dep_files()[0]
    st = os.stat(src)
    try:
        os.utime(src, (time.time() + 3600, time.time() + 3600))  # looks newer than the library
        assert not b.needs_build()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    monkeypatch.setattr(b, "build_hash", lambda: "0" * 64)  # as if a source had changed
    assert b.needs_build()
    monkeypatch.undo()
    monkeypatch.setenv("W2A_CXXFLAGS", "-DLANES=8")  # other compiler flags = another library
    assert b.needs_build()
