"""Capture golden vectors from the *unmodified* reference HeatAlertEnv.

Runs ONLY in the build container (needs /root/reference); its outputs are committed:

  tests/golden/mini/...            small synthetic data set in the reference's on-disk
                                   format (made by weather2alert_amd.synth, seed 7)
  tests/golden/mini_traj.npz       trajectories of the reference env on that data set
  tests/golden/full_anchor.npz     reset tuples + self-contained episodes of the reference env
                                   on a full-size (S=746 weight columns) synthetic data set
                                   that is NOT committed (regenerated from its seed)

How the reference is run: ``import weather2alert.env`` from
/root/reference/src with (i) a throw-away stand-in for the missing ``gymnasium`` package
(Env / spaces.Box / spaces.Discrete, written to a temp dir), and (ii) the module-level name
``hf_hub_download`` re-pointed at local files. The env code itself is untouched; nothing of
it is copied here.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from weather2alert_amd import synth  # noqa: E402

REF_SRC = "/root/reference/src"
T = 153


def import_reference(shim_dir: str):
    os.makedirs(os.path.join(shim_dir, "gymnasium"), exist_ok=True)
    with open(os.path.join(shim_dir, "gymnasium", "__init__.py"), "w") as f:
        f.write(textwrap.dedent("""
            import numpy as _np
            class Env:
                def __init__(self): pass
            class _Space:
                pass
            class _Box(_Space):
                def __init__(self, low, high, shape, dtype):
                    self.low, self.high, self.shape, self.dtype = low, high, shape, dtype
            class _Discrete(_Space):
                def __init__(self, n): self.n = n
                def sample(self): return int(_np.random.randint(self.n))
            class spaces:
                Box = _Box
                Discrete = _Discrete
        """))
    sys.path.insert(0, shim_dir)
    sys.path.insert(0, REF_SRC)
    import weather2alert.env as refenv

    return refenv


def patch_hub(refenv, root: str):
    def local(repo_id=None, repo_type=None, subfolder=None, filename=None, local_dir=None, **kw):
        p = os.path.join(root, subfolder, filename)
        assert os.path.exists(p), p
        return p

    refenv.hf_hub_download = local


def sig_code(v, cats):
    if v is None or (isinstance(v, float) and np.isnan(v)):
        return 0.0
    return float(cats.index(v) + 1)


def numeric_obs(values, names, cats):
    out = np.empty(len(values), dtype=np.float64)
    for i, (v, n) in enumerate(zip(values, names)):
        out[i] = sig_code(v, cats) if n == "significance" else float(v)
    return out


def run_episode(env, cats, reset_kwargs, actions, rec):
    obs, info = env.reset(**reset_kwargs)
    names = info["feature_names"]
    ep = {
        "info_location": info["location"],
        "location_index": int(info["location_index"]),
        "episode_index": info["episode_index"],
        "coef_index": int(env.coef_index),
        "budget": int(env.budget),
        "n_days": int(env.n_days),
        "obs0": numeric_obs(obs, names, cats),
        "reset_remaining_budget": int(info["remaining_budget"]),
    }
    n = len(actions)
    A = np.zeros(n, np.int64)
    R = np.zeros(n, np.float64)
    D = np.zeros(n, np.bool_)
    O = np.zeros((n, len(names)), np.float64)
    RB = np.zeros(n, np.int64)
    AB = np.zeros(n, np.bool_)
    ST = np.zeros(n, np.int64)
    TT = np.zeros(n, np.int64)
    for i, a in enumerate(actions):
        obs, r, done, trunc, info = env.step(int(a))
        assert trunc is False
        R[i], D[i] = r, done
        O[i] = numeric_obs(obs, names, cats)
        RB[i], AB[i] = info["remaining_budget"], info["at_budget"]
        A[i] = env.actual_alert_buffer[-1]
        ST[i], TT[i] = env.alert_streak, env.t
        if done:
            assert i == n - 1 or n < env.n_days
    ep.update(actions=np.asarray(actions, np.int64), actual=A, reward=R, done=D, obs=O,
              remaining_budget=RB, at_budget=AB, streak_after=ST, t_after=TT)
    ep["feature_names"] = list(names)
    rec.append(ep)
    return ep


def scenarios():
    """(env_key, ctor_kwargs, reset_kwargs, action_spec) tuples. Episodes that share env_key
    run on ONE env object in order (exercises the sticky budget)."""
    S = []
    for s in range(6):
        S.append((f"plain{s}", {}, dict(location="06037", seed=s), ("bern", 0.3, 1000 + s)))
    for s in range(10, 16):
        S.append((f"randloc{s}", {}, dict(seed=s), ("bern", 0.2, 1000 + s)))
    for s in range(20, 26):
        S.append((f"aug{s}", {}, dict(location="06037", similar_climate_counties=True, seed=s),
                  ("bern", 0.3, 1000 + s)))
    for s in range(30, 34):
        S.append((f"augrand{s}", dict(similar_climate_counties=True), dict(seed=s), ("bern", 0.25, 1000 + s)))
    for b in (0, 1, 5):
        S.append((f"budget{b}", {}, dict(location="06037", seed=40 + b, budget=b), ("ones",)))
        S.append((f"budget{b}r", {}, dict(seed=50 + b, budget=b), ("bern", 0.5, 2000 + b)))
    S.append(("zeros", {}, dict(location="06037", seed=60), ("zeros",)))
    S.append(("ones", {}, dict(location="06037", seed=61), ("ones",)))
    for s in range(70, 74):
        S.append((f"sb_less{s}", {}, dict(seed=s, budget=6, sample_budget=True), ("bern", 0.4, 3000 + s)))
        S.append((f"sb_cent{s}", {}, dict(seed=s, budget=6, sample_budget=True, sample_budget_type="centered"),
                  ("bern", 0.4, 3100 + s)))
    for s in range(80, 83):
        S.append((f"sb_tbl{s}", {}, dict(seed=s, sample_budget=True, sample_budget_type="centered"),
                  ("bern", 0.3, 3200 + s)))
    # sticky budget sequences on one env
    for s in range(90, 95):
        S.append(("sticky_plain", {}, dict(seed=s), ("bern", 0.3, 4000 + s)))
    for s in range(100, 105):
        S.append(("sticky_sample", dict(budget=9), dict(seed=s, sample_budget=True), ("bern", 0.5, 4100 + s)))
    for s in range(110, 114):
        S.append(("sticky_kw", {}, dict(seed=s, budget=2 + (s % 3)), ("bern", 0.5, 4200 + s)))
    for s in range(120, 123):
        S.append(("ctor_budget3", dict(budget=3), dict(location="06037", seed=s), ("bern", 0.5, 4300 + s)))
    return S


def make_actions(spec, n):
    if spec[0] == "zeros":
        return np.zeros(n, np.int64)
    if spec[0] == "ones":
        return np.ones(n, np.int64)
    _, p, seed = spec
    return (np.random.default_rng(seed).random(n) < p).astype(np.int64)


def pack(rec, path, extra=None):
    keys_arr = ["obs0", "actions", "actual", "reward", "done", "obs", "remaining_budget", "at_budget",
                "streak_after", "t_after"]
    out = {k: np.stack([e[k] for e in rec]) for k in keys_arr}
    for k in ["location_index", "coef_index", "budget", "n_days", "reset_remaining_budget"]:
        out[k] = np.asarray([e[k] for e in rec], np.int64)
    meta = [{k: e[k] for k in ["env_key", "ctor", "reset", "info_location", "episode_index"]} for e in rec]
    out["meta_json"] = np.asarray(json.dumps({"episodes": meta, "feature_names": rec[0]["feature_names"],
                                              **(extra or {})}))
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if hasattr(v, "shape")})


def main():
    import numpy
    import pandas
    import scipy

    versions = {"numpy": numpy.__version__, "pandas": pandas.__version__, "scipy": scipy.__version__}
    tmp = tempfile.mkdtemp(prefix="w2a_golden_")
    refenv = import_reference(os.path.join(tmp, "shim"))

    # ---------------- mini data set (committed) ----------------
    mini_root = os.path.join(HERE, "mini")
    mini = synth.make_synth("linear", n_fips=24, years=[2006, 2007, 2008], n_samples=8, seed=7,
                            extra_confounder_fips=8)
    loc = synth.write_reference_files(mini, mini_root, weights="linear", split="65k")
    patch_hub(refenv, mini_root)
    cats = sorted(synth.SIGNIFICANCE_VALUES)
    rec = []
    envs = {}
    for key, ctor, reset, aspec in scenarios():
        if key not in envs:
            envs[key] = refenv.HeatAlertEnv(weights="linear", data_dir=mini_root, **ctor)
        env = envs[key]
        ep = run_episode(env, cats, reset, make_actions(aspec, T), rec)
        ep.update(env_key=key, ctor=ctor, reset=reset)
    e0 = next(iter(envs.values()))
    extra = {
        "versions": versions,
        "significance_categories": cats,
        "fips_list": e0.fips_list,
        "valid_years": [int(y) for y in e0.valid_years],
        "n_samples": int(e0.n_samples),
        "baseline_keys": list(e0.baseline_coefs.keys()),
        "effectiveness_keys": list(e0.effectiveness_coefs.keys()),
        "declared_obs_shape": list(e0.observation_space.shape),
    }
    pack(rec, os.path.join(HERE, "mini_traj.npz"), extra)
    # the same data set compiled to dense tables: the GPU box has no parquet engine, so the
    # -m gpu tests load this instead of the parquet files (tests/test_tables.py proves they agree)
    from weather2alert_amd import tables as _tables

    _tables.compile_from_files(mini_root, "linear").save_npz(os.path.join(HERE, "mini_compiled.npz"))

    # ---------------- full-size anchors (tables not committed) ----------------
    full_root = os.path.join(tmp, "full")
    full = synth.make_synth("linear", n_counties_weather=41, years=list(range(2006, 2017)), n_samples=100,
                            seed=0, extra_confounder_fips=60)
    synth.write_reference_files(full, full_root, weights="linear", split="65k")
    patch_hub(refenv, full_root)
    env = refenv.HeatAlertEnv(weights="linear", data_dir=full_root)
    rec = []
    eps_tables, eps_w = [], []
    sc = [(dict(location="06037", seed=s), ("bern", 0.3, 5000 + s)) for s in (0, 1, 123)]
    sc += [(dict(location="06037", similar_climate_counties=True, seed=s), ("bern", 0.3, 5000 + s)) for s in (5, 6)]
    sc += [(dict(location=full.fips_weather[3], seed=9), ("bern", 0.3, 5009))]
    for reset, aspec in sc:
        env.budget = None  # fresh budget per anchor episode (each behaves like a new env)
        ep = run_episode(env, cats, reset, make_actions(aspec, T), rec)
        ep.update(env_key="full", ctor={}, reset=reset)
        names = ep["feature_names"]
        tab = np.zeros((T, len(names) - 1), np.float64)
        for j, n in enumerate(names[:-1]):
            col = env.ep[n].values
            tab[:, j] = [sig_code(v, cats) for v in col] if n == "significance" else col.astype(np.float64)
        eps_tables.append(tab)
        li, ci = ep["location_index"], ep["coef_index"]
        wb = np.asarray([v[ci, 0, li].item() for v in env.baseline_coefs.values()], np.float32)
        we = np.asarray([v[ci, 0, li].item() for v in env.effectiveness_coefs.values()], np.float32)
        eps_w.append(np.stack([wb, we]))
    extra = {
        "versions": versions,
        "significance_categories": cats,
        "n_fips": len(env.fips_list),
        "index_06037": env.fips_list.index("06037"),
        "valid_years": [int(y) for y in env.valid_years],
        "n_samples": int(env.n_samples),
        "baseline_keys": list(env.baseline_coefs.keys()),
        "effectiveness_keys": list(env.effectiveness_coefs.keys()),
        "n_similar_06037": len([x for x in refenv.get_similar_counties("06037", env.confounders)
                                if x in env.fips_list]),
        "synth": {"weights_name": "linear", "n_counties_weather": 41, "seed": 0, "extra_confounder_fips": 60},
    }
    path = os.path.join(HERE, "full_anchor.npz")
    pack(rec, path, extra)
    d = dict(np.load(path))
    d["episode_table"] = np.stack(eps_tables)
    d["episode_weights"] = np.stack(eps_w)
    np.savez_compressed(path, **d)
    print("anchors:", [(e["reset"], e["episode_index"], e["coef_index"], e["location_index"], e["info_location"])
                       for e in rec])


if __name__ == "__main__":
    main()
