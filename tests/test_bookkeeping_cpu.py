"""The handle's host-side bookkeeping (weather2alert_amd/csrc/w2a_bookkeeping.h -- the header libw2a.so itself compiles)
on the CPU, under AddressSanitizer + UBSan, against a recording stub that knows which form of the per-env state is really
current, which day the envs are really on and which budgets they really hold (tests/bookkeeping_check.cpp): random call
sequences of every entry point that touches a validity flag (the round-2 advisor found two stale-flag
bugs of this kind by reading, the GPU sequence fuzz covers the same ground end to end).

Second half: MUTANTS of the header -- one rule broken each (a flag not cleared, a day not checked, a bound not kept) --
must every one be caught by the same program. That is what says the harness would have seen such a bug."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "weather2alert_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "bookkeeping_check.cpp")
FLAGS = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Werror"]

# (name, text in w2a_bookkeeping.h, replacement): each breaks exactly one rule
MUTANTS = [
    ("reset keeps a stale column grouping (round 2's stale perm_valid)",
     "  if (!observe_only) b.perm_valid = 0;  // new episode tuples: the column grouping is stale\n", "\n"),
    ("sort keeps a stale column grouping",
     "  b.perm_valid = 0;  // every env index now holds another episode: the column grouping is stale\n", "\n"),
    ("in-kernel autoreset keeps the column grouping",
     "    b.perm_valid = 0;  // tile list go stale", "    //"),
    ("modified canonical words leave the mirror marked current",
     "  b.pk_valid = 0;\n  if (!keeps_lockstep) b.uni_t = -1;", "  if (!keeps_lockstep) b.uni_t = -1;"),
    ("reads of the canonical words never unpack", "  if (b.canon_valid) return;\n  d.unpack_state", "  return;\n  d.unpack_state"),
    ("packed step leaves the canonical words marked current", "      b.canon_valid = 0;\n", "\n"),
    ("lock-step day survives the terminal step", "b.uni_t + 1 < b.uni_nd) ? b.uni_t + 1 : -1;", "true) ? b.uni_t + 1 : -1;"),
    ("masked reset claims lock step", "if (!masked && b.uni_nd > 0 && !b.graph_captured) b.uni_t = 0;",
     "if (b.uni_nd > 0 && !b.graph_captured) b.uni_t = 0;"),
    ("ragged tables claim lock step", "if (!masked && b.uni_nd > 0 && !b.graph_captured) b.uni_t = 0;",
     "if (!masked && !b.graph_captured) b.uni_t = 0;"),
    ("invalidate keeps the lock-step day", "b.pk_valid = 0; b.canon_valid = 1; b.uni_t = -1; b.perm_valid = 0;",
     "b.pk_valid = 0; b.canon_valid = 1; b.perm_valid = 0;"),
    ("invalidate keeps the mirror", "b.pk_valid = 0; b.canon_valid = 1; b.uni_t = -1; b.perm_valid = 0;",
     "b.canon_valid = 1; b.uni_t = -1; b.perm_valid = 0;"),
    ("a captured handle goes back to the packed form", "b.budget_bound <= W2A_BK_PACKED_MAX_BUDGET && !b.graph_captured;",
     "b.budget_bound <= W2A_BK_PACKED_MAX_BUDGET;"),
    ("a captured handle keeps claiming the day after a reset", "if (!masked && b.uni_nd > 0 && !b.graph_captured) b.uni_t = 0;",
     "if (!masked && b.uni_nd > 0) b.uni_t = 0;"),
    ("rollout to the end keeps the lock-step day", "(day >= 0 && day + n_steps < b.uni_nd && !b.graph_captured)",
     "(day >= 0 && !b.graph_captured)"),
    ("a sticky centred budget counts as bounded", "  if (centered && sticky) { b.budget_bound = b.budget_bound_known = W2A_BK_UNKNOWN; return; }\n", "\n"),
    ("a sticky random walk is forgotten by the next statement (the r3 rule)",
     "  if (centered && sticky) { b.budget_bound = b.budget_bound_known = W2A_BK_UNKNOWN; return; }",
     "  if (centered && sticky) { if (b.budget_bound != W2A_BK_UNKNOWN) b.budget_bound_known = b.budget_bound; b.budget_bound = W2A_BK_UNKNOWN; return; }"),
    ("a stated bound forgets earlier sticky budgets", "    b.budget_bound = bound > prev ? bound : prev;", "    b.budget_bound = bound;"),
    ("budgets in device memory count as known", "  if (cand < 0) { b.budget_bound = W2A_BK_UNKNOWN; return; }\n", "  if (cand < 0) return;\n"),
    ("a statement after a restore forgets the autoreset parameters",
     "  if (b.has_auto) bk_note_budgets(b, b.auto_cand, b.auto_centered != 0, b.auto_sticky != 0);\n", "\n"),
    ("the packed form ignores the budget bound", "b.uni_t >= 0 &&\n                        b.budget_bound <= W2A_BK_PACKED_MAX_BUDGET &&", "b.uni_t >= 0 &&"),
    ("the packed form ignores the table limits", "!unpacked_flag && b.pk_static_ok && b.uni_t >= 0", "!unpacked_flag && b.uni_t >= 0"),
    ("a new visiting order keeps the old tile list", "static inline void bk_order_set(W2aBook &b) { b.has_order = 1; b.rm_valid = 0; }",
     "static inline void bk_order_set(W2aBook &b) { b.has_order = 1; }"),
    ("reset keeps the matrix-core rollout's tile list", "  b.rm_valid = 0;  // new episodes: the feature-row tile list of the matrix-core rollout is stale\n", "\n"),
    ("the day restored on unpacking is the day after the terminal step", "b.pk_t = uni_next >= 0 ? uni_next : b.uni_t;", "b.pk_t = b.uni_t + 1;"),
    ("a REWARD_GIVEN step runs packed", "const bool packed = !given && !autoreset", "const bool packed = !autoreset"),
    ("an autoreset step runs packed", "const bool packed = !given && !autoreset &&", "const bool packed = !given &&"),
    ("a rollout does not bring the canonical words up to date", "  const int32_t day = b.uni_t;\n  bk_ensure_canonical(b, d);", "  const int32_t day = b.uni_t;"),
    ("a rollout leaves the mirror marked current", "  bk_ensure_canonical(b, d);\n  bk_canonical_modified(b, true);\n  b.uni_t = (day >= 0",
     "  bk_ensure_canonical(b, d);\n  b.uni_t = (day >= 0"),
]


def _build(inc_dir, out):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    r = subprocess.run([gxx, *FLAGS, f"-I{inc_dir}", SRC, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def test_bookkeeping_sequences_under_sanitizers(tmp_path):
    exe = _build(CSRC, str(tmp_path / "bkcheck"))
    for seed in (1, 2, 3):
        r = subprocess.run([exe, "4000", "160", str(seed)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (seed, r.stdout[-3000:], r.stderr[-3000:])
        assert "no violation" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr


def test_every_mutant_of_the_bookkeeping_is_caught(tmp_path):
    hdr = open(os.path.join(CSRC, "w2a_bookkeeping.h")).read()
    missed = []
    for k, (name, old, new) in enumerate(MUTANTS):
        assert hdr.count(old) == 1, f"mutant {k} ({name}): its anchor text occurs {hdr.count(old)} times in the header"
        d = tmp_path / f"m{k}"
        d.mkdir()
        (d / "w2a_bookkeeping.h").write_text(hdr.replace(old, new))
        exe = _build(str(d), str(d / "bkcheck"))
        for seed in range(1, 9):  # most mutants die within the first few hundred sequences; the budget ones need rarer
            r = subprocess.run([exe, "4000", "200", str(seed)], capture_output=True, text=True, timeout=600)  # set-ups
            if r.returncode != 0:
                assert "VIOLATION" in r.stdout, (name, r.stdout[-500:], r.stderr[-1500:])
                break
        else:
            missed.append(name)
    assert not missed, f"mutants the harness did not catch: {missed}"


if __name__ == "__main__":
    sys.exit(pytest.main([__file__, "-q"]))
