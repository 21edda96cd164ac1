"""Model-based call-sequence fuzz of the libw2a handle: 240 random sequences of 30-80 operations each
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

N_SEQUENCES = 240
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
    assert tot["ops"] >= 30 * N_SEQUENCES and tot["steps"] > 5000 and tot["resets"] > 1000 and tot["rollouts"] > 800
    assert tot["packed_steps"] > 500 and tot["mfma_rollouts"] > 50 and tot["graphs"] > 20 and tot["ckpt"] > 100
    assert tot["after_done"] > 50 and tot["autoresets"] > 1000


@pytest.mark.parametrize("seq", [])
def test_call_sequence_regressions(seq):
    """Sequences that once failed (findings of the fuzz), replayed by number: filled in as findings are fixed."""
    sf = _load()
    sf.run_sequence(seq, MASTER_SEED, torch.device("cuda:0"))
