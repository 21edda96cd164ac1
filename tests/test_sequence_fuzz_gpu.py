"""Model-based call-sequence fuzz of the libw2a handle: 1000 random sequences of 30-80 operations each
-- resets (device RNG / injected tuples, masked / unmasked), steps in every kernel form and autoreset mode, partial and
whole rollouts, state(), checkpoints, w2a_invalidate / w2a_set_budget_bound, episode_order="sorted", the posterior-mean
reward with each kernel, hipGraph capture + replays -- mirrored on oracle/sequence_model.HandleModel; outputs compared
after every operation, w2a_query against what the sequence implies (tools/sequence_fuzz.py is the long form).
The reference allows reset/step to interleave arbitrarily (env.py:133-184,238-262); round 3 added seven validity flags
to the handle whose interleavings had one scripted test."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

N_SEQUENCES = 1000  # ~0.1 s each on MI355X (even ones read the state after every operation)
MASTER_SEED = 2024


def _load():
    spec = importlib.util.spec_from_file_location("sequence_fuzz", os.path.join(os.path.dirname(__file__), "..", "tools",
                                                                               "sequence_fuzz.py"))
    sf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sf)
    return sf


@pytest.mark.timeout(900)
def test_call_sequence_fuzz_against_the_model():
    assert torch.cuda.is_available()
    sf = _load()
    dev = torch.device("cuda:0")
    tot: dict = {}
    for i in range(N_SEQUENCES):
        s = sf.run_sequence(i, MASTER_SEED, dev)  # raises SequenceFailure with the operation log on a violation
        for k, v in s.items():
            tot[k] = max(tot.get(k, 0.0), v) if k == "worst" else tot.get(k, 0) + v
    print(f"sequence fuzz: {N_SEQUENCES} sequences, {tot}")
    assert tot["worst"] <= 1e-5
    # the sweep must actually have visited what it is for
    assert tot["ops"] >= 60 * N_SEQUENCES and tot["steps"] > 40 * N_SEQUENCES and tot["resets"] > 8 * N_SEQUENCES
    assert tot["rollouts"] > 4 * N_SEQUENCES and tot["packed_steps"] > N_SEQUENCES // 2 and tot["mfma_rollouts"] > N_SEQUENCES // 4
    assert tot["graphs"] > N_SEQUENCES // 4 and tot["ckpt"] > N_SEQUENCES and tot["after_done"] > 5 * N_SEQUENCES
    assert tot["autoresets"] > 100 * N_SEQUENCES


def test_policy_loop_over_a_ragged_batch_raises_no_status_bit():
    """Finding 1 of the fuzz (round 4, sequence 82 of seed 2024). reward_mode='posterior_mean' with the fp64 matrix kernel
    has no one-launch rollout, so rollout() runs the per-day calls w2a_policy_actions / w2a_posterior_mean_reward /
    w2a_step(REWARD_GIVEN | SKIP_FINISHED). After a masked reset the envs finish on different days; the step skips the
    finished ones silently, but the reward pre-pass (k_pm_prep) raised W2A_ST_STEP_AFTER_DONE for them: the status word
    came back 4 after a rollout in which nothing was wrong."""
    import numpy as np

    from weather2alert_amd import HeatAlertVecEnv, synth, tables

    sd = synth.make_synth("linear", n_fips=12, years=[2006, 2007, 2008], n_samples=6, n_days=16, seed=0, extra_confounder_fips=2)
    ct = tables.compile_from_synth(sd)
    n = 127
    for pmk in ("matrix", "vector", "matrix_i8"):
        env = HeatAlertVecEnv(n, tables=ct, device="cuda:0", autoreset="disabled", reward_mode="posterior_mean", pm_kernel=pmk,
                              step_kernel="wide")
        env.pm_rollout_kernel = pmk == "matrix_i8"  # the two others through the per-day calls
        env.reset(seed=1)
        for _ in range(9):
            env.step(torch.ones(n, dtype=torch.int32, device="cuda:0"))
        env.reset(seed=2, options={"mask": np.arange(n) % 11 == 0})
        out = env.rollout(dict(kind="bernoulli", p=0.3, seed=5), alert_mask=True)
        assert bool(out["done"].all()) and env.check_status() == 0, pmk
        env.close()
