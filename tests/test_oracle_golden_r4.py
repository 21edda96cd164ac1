"""Pin the scalar oracle to call SEQUENCES captured from the unmodified reference env (tests/golden/make_golden_r4.py):
resets in the middle of episodes with every kwarg, seed=None through the global NumPy generator, and steps after `done`
(449 of them) -- the interleaving semantics of env.py:133-184,238-262 that the GPU sequence tests take from the oracle.
9 361 operations on 14 env objects; float64 rewards and observations bit-exact, integers and strings exact."""
import json
import os

import numpy as np
import pytest

from oracle import heatalert_oracle as O


@pytest.fixture(scope="module")
def seqs(golden_dir, mini_root):
    d = dict(np.load(os.path.join(golden_dir, "mini_sequences.npz")))
    return d, json.loads(str(d["meta_json"])), O.RefData.from_files(mini_root, weights="linear", split="65k")


def iter_ops(d, meta):
    """(sequence index, position, reset dict or None, flat index) over all operations."""
    k = 0
    for si, q in enumerate(meta["sequences"]):
        resets = {r["at"]: r for r in q["resets"]}
        strs = {r["at"]: r for r in meta["info_str"][si]}
        for i in range(q["n_ops"]):
            yield si, i, resets.get(i), strs.get(i), k
            k += 1


def test_scalar_oracle_reproduces_reference_call_sequences(seqs):
    d, meta, data = seqs
    A = meta["attr_names"]
    envs = {}
    n_after_done = n_mid_resets = 0
    cur_str = None
    for si, i, rs, st, k in iter_ops(d, meta):
        if si not in envs:
            envs[si] = O.OracleEnv(data, **meta["sequences"][si]["ctor"])
        env = envs[si]
        if rs is not None:
            n_mid_resets += i > 0 and env.t < env.n_days - 1
            np.random.seed(rs["global_seed"])  # seed=None draws from the global generator (env.py:143-144)
            obs, info = env.reset(**rs["kwargs"])
            cur_str = (st["episode_index"], st["location"])
            assert np.isnan(d["reward"][k]) and not d["done"][k]
        else:
            n_after_done += env.t >= env.n_days - 1 and len(env.actual_alert_buffer) >= env.n_days
            obs, r, done, trunc, info = env.step(int(d["action"][k]))
            assert r == d["reward"][k], (si, i, r, d["reward"][k])
            assert done == d["done"][k] and trunc is False
        np.testing.assert_array_equal(obs, d["obs"][k], err_msg=f"sequence {si} op {i}")
        assert (info["remaining_budget"], int(info["at_budget"]), info["location_index"]) == tuple(d["info_int"][k]), (si, i)
        assert (info["episode_index"], info["location"]) == cur_str, (si, i)
        got = [env.t, env.alert_streak, env.budget, env.coef_index, env.n_days, env.remaining_budget, int(env.at_budget)]
        assert got == list(d["attrs"][k]), (si, i, dict(zip(A, got)), dict(zip(A, d["attrs"][k])))
    assert k + 1 == len(d["obs"]) == 9361 and n_after_done > 300 and n_mid_resets > 40


def test_vector_oracle_reproduces_reference_call_sequences(seqs):
    """The vectorised oracle (what the GPU parity tests and the call-sequence model compute with) on the same recording,
    one env at a time: episode tuples from what the reference reported at each reset; rewards, observations and done --
    also for steps after `done` -- bit-exact against the reference."""
    d, meta, data = seqs
    fw = sorted({k[0] for k in data.episodes})
    V = O.VectorOracle(data, fw, data.valid_years)
    n_after = 0
    for si, i, rs, st, k in iter_ops(d, meta):
        if rs is not None:
            f, y = st["episode_index"].split("_")
            t, streak, budget, coef_index, n_days, rem, atb = (int(x) for x in d["attrs"][k])
            obs = V.reset([fw.index(f)], [data.valid_years.index(int(y))], [int(d["info_int"][k][2])], [coef_index], [budget])
            assert V.n_days[0] == n_days
        else:
            n_after += bool(V.t[0] >= V.n_days[0] - 1 and V.used[0] + 0 >= 0 and d["attrs"][k - 1][0] == V.n_days[0] - 1 and d["done"][k - 1])
            obs, r, done, actual = V.step(np.asarray([int(d["action"][k])]))
            assert r[0] == d["reward"][k] and bool(done[0]) == bool(d["done"][k]), (si, i)
            assert [int(V.t[0]), int(V.streak[0]), int(V.budget[0] - V.used[0]), int(V.at_budget[0])] == \
                [int(d["attrs"][k][0]), int(d["attrs"][k][1]), int(d["attrs"][k][5]), int(d["attrs"][k][6])], (si, i)
        np.testing.assert_array_equal(obs[0], d["obs"][k], err_msg=f"sequence {si} op {i}")
    assert n_after > 300
