"""Round-4 capture from the *unmodified* reference HeatAlertEnv: call SEQUENCES beyond "reset, then one episode".

Runs ONLY in the build container (needs /root/reference); output committed: tests/golden/mini_sequences.npz.
Same import recipe as make_golden.py (a throw-away `gymnasium` stand-in on sys.path, `hf_hub_download` re-pointed at
the committed tests/golden/mini data set; the env code is untouched, nothing of it is copied).

What is recorded: per sequence one reference env object and 60-120 operations drawn at random --
  reset(**kwargs)   every kwarg of env.py:133-141, in the MIDDLE of episodes too, with seed=None now and then (the
                    reference then draws its seed from the GLOBAL NumPy generator, env.py:143-144: the capture seeds
                    that generator right before the call and records the value);
  step(action)      also after `done` (the reference keeps answering: env.py:238-262 has no guard) --
and after every operation the observation (numeric), reward, done, the info dict's entries and the env attributes
t / alert_streak / budget / coef_index / n_days / remaining_budget / at_budget. These pin the interleaving semantics
that oracle.OracleEnv restates and that the GPU sequence tests rely on (tests/test_oracle_golden_r4.py)."""
from __future__ import annotations

import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import ROOT, import_reference, numeric_obs, patch_hub  # noqa: E402

sys.path.insert(0, ROOT)
from weather2alert_amd import synth  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="w2a_golden_r4_")
    refenv = import_reference(os.path.join(tmp, "shim"))
    mini_root = os.path.join(HERE, "mini")
    patch_hub(refenv, mini_root)
    cats = sorted(synth.SIGNIFICANCE_VALUES)
    import yaml

    fips_list = [str(x) for x in yaml.safe_load(open(os.path.join(mini_root, "linear", "config.yaml")))["fips_list"]]
    seqs = []
    for s in range(14):
        rng = np.random.default_rng([2024, s])
        ctor = dict(similar_climate_counties=bool(rng.random() < 0.4))
        if rng.random() < 0.4:
            ctor["budget"] = int(rng.integers(0, 6))
        env = refenv.HeatAlertEnv(weights="linear", **ctor)
        ops = []
        fresh = True
        for _ in range(int(rng.integers(60, 121))):
            rec = {}
            if fresh or rng.random() < 0.06:
                kw = {}
                if rng.random() < 0.5:
                    kw["location"] = str(rng.choice(fips_list))
                if rng.random() < 0.5:
                    kw["similar_climate_counties"] = bool(rng.random() < 0.5)
                if rng.random() < 0.8:
                    kw["seed"] = int(rng.integers(0, 10000))
                if rng.random() < 0.5:
                    kw["budget"] = int(rng.integers(0, 8))
                if rng.random() < 0.35:
                    kw["sample_budget"] = True
                    kw["sample_budget_type"] = str(rng.choice(["less_than", "centered"]))
                g = int(rng.integers(0, 1 << 30))
                np.random.seed(g)
                obs, info = env.reset(**kw)
                rec.update(op="reset", kwargs=kw, global_seed=g, reward=None, done=False)
                fresh = False
            else:
                # short stretches near the end of an episode are made likely: jump close to the last day now and then
                a = int(rng.random() < 0.45)
                obs, r, done, trunc, info = env.step(a)
                assert trunc is False
                rec.update(op="step", action=a, reward=float(r), done=bool(done))
            names = info["feature_names"]
            rec.update(obs=[float(x) for x in numeric_obs(obs, names, cats)],
                       info={k: (info[k] if isinstance(info[k], str) else int(info[k]) if k != "at_budget" else bool(info[k]))
                             for k in ("episode_index", "remaining_budget", "at_budget", "location", "location_index")},
                       attrs=dict(t=int(env.t), alert_streak=int(env.alert_streak), budget=int(env.budget),
                                  coef_index=int(env.coef_index), n_days=int(env.n_days),
                                  remaining_budget=int(env.remaining_budget), at_budget=bool(env.at_budget)))
            ops.append(rec)
            if rec["op"] == "step" and rec["done"] and rng.random() < 0.5:
                pass  # keep stepping the finished episode (the reference allows it)
            elif rec["op"] == "step" and rec["done"] is False and env.t < env.n_days - 12 and rng.random() < 0.15:
                # fast-forward (recorded like any other steps) so that ends of episodes are reached inside a sequence
                for _ in range(env.n_days - 8 - env.t):
                    a = int(rng.random() < 0.3)
                    obs, r, done, trunc, info = env.step(a)
                    ops.append(dict(op="step", action=a, reward=float(r), done=bool(done),
                                    obs=[float(x) for x in numeric_obs(obs, names, cats)],
                                    info={k: (info[k] if isinstance(info[k], str) else int(info[k]) if k != "at_budget" else bool(info[k]))
                                          for k in ("episode_index", "remaining_budget", "at_budget", "location", "location_index")},
                                    attrs=dict(t=int(env.t), alert_streak=int(env.alert_streak), budget=int(env.budget),
                                               coef_index=int(env.coef_index), n_days=int(env.n_days),
                                               remaining_budget=int(env.remaining_budget), at_budget=bool(env.at_budget))))
        seqs.append(dict(ctor=ctor, ops=ops))
        print(f"sequence {s}: {len(ops)} operations, {sum(o['op'] == 'reset' for o in ops)} resets, "
              f"{sum(o['op'] == 'step' and o['done'] for o in ops)} steps returning done")
    import numpy
    import pandas
    import scipy

    # flat arrays (compress well) + a small JSON of what is not numeric
    flat = [o for q in seqs for o in q["ops"]]
    arr = dict(
        obs=np.asarray([o["obs"] for o in flat], np.float64),
        reward=np.asarray([np.nan if o["reward"] is None else o["reward"] for o in flat], np.float64),
        done=np.asarray([o["done"] for o in flat], np.bool_),
        action=np.asarray([o.get("action", -1) for o in flat], np.int8),
        info_int=np.asarray([[o["info"]["remaining_budget"], int(o["info"]["at_budget"]), o["info"]["location_index"]]
                             for o in flat], np.int32),
        attrs=np.asarray([[o["attrs"][k] for k in ("t", "alert_streak", "budget", "coef_index", "n_days", "remaining_budget")] +
                          [int(o["attrs"]["at_budget"])] for o in flat], np.int32))
    meta = dict(versions={"numpy": numpy.__version__, "pandas": pandas.__version__, "scipy": scipy.__version__},
                data="tests/golden/mini (weights 'linear', split '65k')",
                attr_names=["t", "alert_streak", "budget", "coef_index", "n_days", "remaining_budget", "at_budget"],
                info_int_names=["remaining_budget", "at_budget", "location_index"],
                sequences=[dict(ctor=q["ctor"], n_ops=len(q["ops"]),
                                resets=[dict(at=i, kwargs=o["kwargs"], global_seed=o["global_seed"])
                                        for i, o in enumerate(q["ops"]) if o["op"] == "reset"]) for q in seqs],
                # the string entries of info change only at resets: recorded per reset
                info_str=[[dict(at=i, episode_index=o["info"]["episode_index"], location=o["info"]["location"])
                           for i, o in enumerate(q["ops"]) if o["op"] == "reset"] for q in seqs])
    for q in seqs:  # (they really are constant between resets)
        cur = None
        for o in q["ops"]:
            if o["op"] == "reset":
                cur = (o["info"]["episode_index"], o["info"]["location"])
            assert (o["info"]["episode_index"], o["info"]["location"]) == cur
    path = os.path.join(HERE, "mini_sequences.npz")
    np.savez_compressed(path, meta_json=np.asarray(json.dumps(meta)), **arr)
    print("wrote", path, os.path.getsize(path), "bytes;", len(flat), "operations")


if __name__ == "__main__":
    main()
