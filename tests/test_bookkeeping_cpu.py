"""The handle's host-side bookkeeping (weather2alert_amd/csrc/w2a_bookkeeping.h -- the header libw2a.so itself compiles)
on the CPU, under AddressSanitizer + UBSan, against a recording stub that knows which form of the per-env state is really
current and which day the envs are really on (tests/bookkeeping_check.cpp): random call
sequences of every entry point that touches a validity flag (the round-2 advisor found two stale-flag
bugs of this kind by reading, the GPU sequence fuzz covers the same ground end to end).

Second half: the same program walks the ABSTRACT state space of the header breadth first to closure (nothing sampled).
Third: MUTANTS of the header -- one rule broken each (a flag not cleared, a day not checked) -- must
every one be caught by the same program. That is what says the harness would have seen such a bug.

Round 6: the header no longer knows anything about budgets (the packed kernel serves any budget, csrc/w2a_common.hip.h
pk_budget16), so the "budgets" walk, its invariant and ten mutants are gone; the walk of what is left grew from two envs and
two-day episodes to three and three."""
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
     "    b.perm_valid = 0;  // list and the row counts go stale\n", "\n"),
    ("modified canonical words leave the mirror marked current",
     "  b.pk_valid = 0;\n  if (!keeps_lockstep) { b.lock = 0; b.uni_t = -1; }", "  if (!keeps_lockstep) { b.lock = 0; b.uni_t = -1; }"),
    ("reads of the canonical words never unpack", "  if (b.canon_valid) return;\n  d.unpack_state", "  return;\n  d.unpack_state"),
    ("packed step leaves the canonical words marked current", "    b.canon_valid = 0;\n    b.uni_t = uni_next;\n", "    b.uni_t = uni_next;\n"),
    ("lock-step day survives the terminal step", "b.uni_t + 1 < b.uni_nd) ? b.uni_t + 1 : -1;", "true) ? b.uni_t + 1 : -1;"),
    ("masked reset claims lock step", "if (!masked && b.uni_nd > 0) b.lock = 1;", "if (b.uni_nd > 0) b.lock = 1;"),
    ("ragged tables claim lock step", "if (!masked && b.uni_nd > 0) b.lock = 1;", "if (!masked) b.lock = 1;"),
    ("masked reset claims the day", "if (!masked && b.uni_nd > 0 && !bk_any_graph(b)) b.uni_t = 0;",
     "if (b.uni_nd > 0 && !bk_any_graph(b)) b.uni_t = 0;"),
    ("invalidate keeps lock step", "b.pk_valid = 0; b.canon_valid = 1; b.lock = 0; b.uni_t = -1; b.perm_valid = 0;",
     "b.pk_valid = 0; b.canon_valid = 1; b.uni_t = -1; b.perm_valid = 0;"),
    ("invalidate keeps the mirror", "b.pk_valid = 0; b.canon_valid = 1; b.lock = 0; b.uni_t = -1; b.perm_valid = 0;",
     "b.canon_valid = 1; b.lock = 0; b.uni_t = -1; b.perm_valid = 0;"),
    ("a handle with a recorded canonical step goes back to the packed form",
     "return bk_packed_eligible(b) && b.lock && !b.graph_canon; }", "return bk_packed_eligible(b) && b.lock; }"),
    ("a handle with a recorded step keeps claiming the day after a reset", "if (!masked && b.uni_nd > 0 && !bk_any_graph(b)) b.uni_t = 0;",
     "if (!masked && b.uni_nd > 0) b.uni_t = 0;"),
    ("rollout to the end keeps the lock-step day", "(day >= 0 && day + n_steps < b.uni_nd && !bk_any_graph(b))",
     "(day >= 0 && !bk_any_graph(b))"),
    ("the packed form ignores the table limits", "  return b.pk_static_ok && b.uni_nd > 0;", "  return b.uni_nd > 0;"),
    ("a new visiting order keeps the old tile list", "static inline void bk_order_set(W2aBook &b) { b.has_order = 1; b.rm_valid = 0; }",
     "static inline void bk_order_set(W2aBook &b) { b.has_order = 1; }"),
    ("reset keeps the matrix-core rollout's tile list", "  b.rm_valid = 0;  // new episodes: the feature-row tile list of the matrix-core rollout is stale\n", "\n"),
    ("a REWARD_GIVEN step runs packed", "bool packed = wide_wanted && !given && !unpacked_flag", "bool packed = wide_wanted && !unpacked_flag"),
    ("a rollout does not bring the canonical words up to date", "  const int32_t day = b.uni_t;\n  bk_ensure_canonical(b, d);", "  const int32_t day = b.uni_t;"),
    ("a rollout leaves the mirror marked current", "  bk_ensure_canonical(b, d);\n  bk_canonical_modified(b, true);\n  b.uni_t = (day >= 0",
     "  bk_ensure_canonical(b, d);\n  b.uni_t = (day >= 0"),
    ("the matrix-core rollout runs without lock step", "if (b.rm_valid && b.has_order && !fixes && lock && mfma_built)",
     "if (b.rm_valid && b.has_order && !fixes && mfma_built)"),
    # ---- round 5: recorded graphs on either form, the poisoned mirror, row counts from k_reset
    ("a capture records the conversion into the packed form", "    if (packed && !b.pk_valid) packed = false;\n", "\n"),
    ("a capture records the conversion back to the canonical words", "    if (!packed && !b.canon_valid) { p.kernel = -1; return p; }\n", "\n"),
    ("after a recorded packed step, a read leaves the canonical words marked current",
     "  if (b.pk_valid) { b.canon_valid = 0; return; }  // a replay may step the mirror at any time from here on", "  if (b.pk_valid) return;"),
    ("after a recorded packed step, nothing keeps the mirror current", "  if (!b.graph_packed) return;\n", "  return;\n"),
    ("a mirror that cannot be kept current is not poisoned", "  } else if (!b.poisoned) {\n    d.poison_mirror();\n    b.poisoned = 1;\n  }", "  }"),
    ("the mirror is re-packed although the batch cannot be packed", "  if (bk_can_pack(b)) {  // the canonical words were modified", "  if (true) {  //"),
    ("a recorded autoreset step is not remembered", "    if (autoreset) b.graph_autoreset = 1;\n", "\n"),
    ("an in-kernel autoreset keeps the row counts", "    b.hist_valid = 0;\n  }\n  if (capturing) {", "  }\n  if (capturing) {"),
    ("a relabelling keeps the per-env ranks", "  b.hist_valid = 0;  // ... and so are the per-env ranks inside the feature rows\n", "\n"),
    ("a masked reset leaves the row counts valid", "b.hist_valid = (!masked && b.has_order_ws && !b.graph_autoreset) ? 1 : 0;",
     "b.hist_valid = (b.has_order_ws && !b.graph_autoreset) ? 1 : 0;"),
    ("a restore keeps the row counts", "  b.hist_valid = 0;\n  b.poisoned = 0;", "  b.poisoned = 0;"),
    ("a restored buffer is taken to be still poisoned (round 5, GPU fuzz seed 505 sequence 357)",
     "  b.poisoned = 0;  // the caller's copy covered the mirror's day words too", "  //"),
    ("the fused sorted reset claims row counts its k_reset pass never took", "  bk_reset(b, d, false, false);\n  b.hist_valid = 0;\n",
     "  bk_reset(b, d, false, false);\n"),
    ("another order workspace inherits the row counts", "  b.has_order_ws = 1; b.hist_valid = 0;\n", "  b.has_order_ws = 1;\n"),
    ("a failed full reset claims the canonical words it never wrote",
     "  const bool unpacked = (masked || observe_only) && !before.canon_valid;", "  const bool unpacked = !before.canon_valid;"),
    # (round 5's "a failed launch after a conversion forgets that the mirror was rewritten" -- bk_step_rollback without its
    # `b.poisoned = 0` -- is no longer a mutant: a step could only convert INTO the packed form on a poisoned handle after
    # w2a_set_budget_bound had made the batch packable again without running bk_end_call, the advisor's round-5 finding;
    # with that entry point gone, poisoned implies "cannot be packed" at every call boundary and the walk shows the line
    # unreachable. It stays in the header as written.)
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


def test_bookkeeping_state_space_walked_to_closure_under_sanitizers(tmp_path):
    """Not sampled: a breadth-first walk of the abstract state space of the header -- every operation
    with every parameter and every outcome of its internal choices from every reachable state, until no new state
    appears; the invariants of the random driver after every transition. Prints how many states there are."""
    import re
    import time

    exe = _build(CSRC, str(tmp_path / "bkcheck"))
    t0 = time.time()
    r = subprocess.run([exe, "--bfs"], capture_output=True, text=True, timeout=900)
    dt = time.time() - t0
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert "walked to closure" in r.stdout and "no violation" in r.stdout, r.stdout[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
    m = re.search(r"(\d+) reachable abstract states, (\d+) transitions", r.stdout)
    states, edges = int(m.group(1)), int(m.group(2))
    assert states > 50_000 and edges > 3_000_000, r.stdout[-600:]  # a walk that explores nothing proves nothing
    print(r.stdout.strip().splitlines()[-1], f"({dt:.0f} s under ASan + UBSan)")


def test_every_mutant_of_the_bookkeeping_is_caught(tmp_path):
    """Each mutant must be caught by the random driver or, failing that, by the exhaustive walk (mutants are built without
    the sanitizers: what is tested here is the harness, and three dozen sanitizer builds would dominate the CPU suite)."""
    hdr = open(os.path.join(CSRC, "w2a_bookkeeping.h")).read()
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    missed, by_walk = [], []
    for k, (name, old, new) in enumerate(MUTANTS):
        assert hdr.count(old) == 1, f"mutant {k} ({name}): its anchor text occurs {hdr.count(old)} times in the header"
        d = tmp_path / f"m{k}"
        d.mkdir()
        (d / "w2a_bookkeeping.h").write_text(hdr.replace(old, new))
        exe = str(d / "bkcheck")
        r = subprocess.run([gxx, "-std=c++17", "-O2", "-Wall", f"-I{d}", SRC, "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stderr[-3000:])
        caught = False
        for seed in range(1, 4):  # most mutants die within the first few hundred sequences
            r = subprocess.run([exe, "3000", "200", str(seed)], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                assert "VIOLATION" in r.stdout, (name, r.stdout[-500:], r.stderr[-1500:])
                caught = True
                break
        if not caught:  # the walk sees every reachable state: what survives it is not observable at all
            r = subprocess.run([exe, "--bfs"], capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                assert "VIOLATION" in r.stdout, (name, r.stdout[-500:], r.stderr[-1500:])
                caught = True
                by_walk.append(name)
        if not caught:
            missed.append(name)
    print(f"{len(MUTANTS)} mutants, {len(by_walk)} of them caught only by the exhaustive walk: {by_walk}")
    assert not missed, f"mutants the harness did not catch: {missed}"


if __name__ == "__main__":
    sys.exit(pytest.main([__file__, "-q"]))
