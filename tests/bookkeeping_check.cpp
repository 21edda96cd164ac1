// bookkeeping_check.cpp -- the handle's host-side bookkeeping (weather2alert_amd/csrc/w2a_bookkeeping.h, the very header
// libw2a.so compiles) on the CPU against a recording stub that KNOWS what is really the case: a handful of simulated envs
// (day, finished, episode length, budget, sticky budget, episode id), for each of the two forms of the per-env step state
// which version of the contents it holds, what the mirror's day word says, and which kinds of step were recorded into
// hipGraphs (a replay runs a recorded kernel with NO bookkeeping at all). Built by tests/test_bookkeeping_cpu.py with
// g++ -fsanitize=address,undefined; the same test also builds mutants of the header (one rule broken each) and requires
// this program to catch every one of them.
//
// Two drivers over the same operations:
//   random      bookkeeping_check <sequences> <ops per sequence> <seed>      six envs, free parameters
//   exhaustive  bookkeeping_check --bfs | --bfs-full [max depth, 0 = closure] breadth-first walk of the ABSTRACT state space
//               to closure, two envs, episodes of two days (one in the budgets walk), as two walks: "forms" = every flag of W2aBook about the two
//               forms of the state, lock step, recorded graphs and the validity of column grouping / visiting order / tile
//               list / row counts x what is really current x the day structure of the batch x the graphs recorded (budgets
//               small, known or out of sight); "budgets" = the budget knowledge of the handle x the budget classes (current,
//               sticky) of the envs x reset / autoreset parameters, on the invariant every use of the 16-bit mirror rests
//               on: the handle's bound is never below a budget the buffer holds. Every operation with every parameter
//               and every outcome of its internal choices from every reachable state; prints the number of reachable
//               states. Nothing is sampled. --bfs-full adds recorded autoreset steps and their replays (which keep drawing
//               budgets with the parameters they were recorded with) to the budgets walk: 7.8 M states, four minutes
//               without sanitizers -- run once per change of the budget rules (profiles/r05/bookkeeping_bfs_full.log),
//               not in the suite.
//
// The entry points below restate, call for call, what csrc/w2a_kernels.hip / w2a_step_dispatch.hip.h do around their
// kernel launches (each names the function it follows); a launch becomes "reads form X" / "writes form X".
//
// Violations reported:
//   stale read        a kernel reads a form of the state that does not hold the latest contents
//   false lock step   the handle claims lock step / a day the envs are not on, or packs a batch that is not on one day
//   packed budgets    the handle's budget bound is below a budget the buffer holds (current or sticky), or the 16-bit
//                     packed form is used while some env's budget exceeds 65535
//   stale grouping    the posterior-mean reward / matrix-core rollout / order placement runs on a grouping, tile list
//                     or row counts of other episodes
//   no valid form     neither form is marked current
//   unsafe replay     at the end of an API call, a recorded graph's kernel would step a form that is not current
//                     (and, for the packed form, not poisoned either)
//
// exit 0 = no violation
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <deque>
#include <string>
#include <unordered_set>
#include <vector>

#include "w2a_bookkeeping.h"

#ifndef NE
#define NE 6  // simulated envs of the random driver
#endif
static const int MAXE = NE > 2 ? NE : 2;
static const int64_t POISON = -7;

// Source of every choice an operation makes by itself (which envs a mask selects, which budget is drawn, ...): random
// numbers in the random driver; in the exhaustive driver a script that is advanced like an odometer until every
// combination of outcomes has been seen (ranges above four values are represented by their two ends).
struct Chooser {
  bool scripted = false;
  uint64_t s = 1;
  std::vector<int> script, arity;
  size_t pos = 0;
  uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  int64_t below(int64_t n) {
    if (n <= 1) return 0;
    if (!scripted) return (int64_t)(next() % (uint64_t)n);
    const int k = n <= 4 ? (int)n : 2;
    if (pos >= script.size()) { script.push_back(0); arity.push_back(k); }
    const int c = script[pos++];
    return n <= 4 ? c : (c ? n - 1 : 0);
  }
  bool coin(int pct) {
    if (!scripted) return below(100) < pct;
    (void)pct;
    return below(2) == 1;
  }
  void rewind() { pos = 0; }
  bool advance() {  // next combination of outcomes; false when all have been seen
    script.resize(pos); arity.resize(pos);
    while (!script.empty()) {
      if (++script.back() < arity.back()) return true;
      script.pop_back(); arity.pop_back();
    }
    return false;
  }
};

struct Cfg { int64_t budget_kw; int mode; int sticky; };  // autoreset / reset parameters (w2a_set_autoreset)
enum { G_PACKED = 1, G_CANON = 2, G_CANON_AUTO_SAME = 4, G_CANON_AUTO_NEXT = 8, G_PACKED_AUTO_SAME = 16, G_PACKED_AUTO_NEXT = 32 };
static const int G_ANY_PACKED = G_PACKED | G_PACKED_AUTO_SAME | G_PACKED_AUTO_NEXT;

struct World {
  int ne = NE;
  bool bfs = false;      // exhaustive driver: budgets are snapped to class representatives so that the space is finite
  // ---- truth
  int32_t day[MAXE], nd[MAXE];
  bool fin[MAXE];
  int64_t budget[MAXE], sticky[MAXE];
  long episode[MAXE];
  long clock = 0, latest = 0, canon_id = 0, packed_id = -1;
  int64_t pk_day_val = -1;  // what the mirror's day words hold (POISON: poisoned)
  long epoch = 0;        // changes whenever any env index gets another episode
  long perm_for = -1, order_exists = 0, rm_for = -1, rm_order_gen = -1, order_gen = 0, hist_for = -1;
  int32_t uni_nd;        // table property
  int32_t b0_max;
  bool static_ok;
  bool has_autoreset = false;
  Cfg acfg{-1, 0, 1};
  int graphs = 0;        // kinds of recorded step (G_*)
  Cfg gcfg[6];           // the autoreset parameters each kind was recorded with (kernel arguments: a replay keeps them)
  int64_t unstated = 0;  // largest budget handed over in device memory that no w2a_set_budget_bound has covered yet
  // ---- the handle
  W2aBook bk;
  std::vector<std::string> trace;  // random driver: what was called (with the launches in between)
  uint16_t path[48];               // exhaustive driver: indices of the operations that led here
  int path_len = 0;
  Chooser *rng;
};
struct Op;
static std::string describe(const Op &o);
static const std::vector<Op> *g_ops = nullptr;

static void print_path(const World &w);
static void fail(World &w, const char *what) {
  printf("VIOLATION: %s\n", what);
  if (w.bfs) print_path(w);
  size_t from = w.trace.size() > 40 ? w.trace.size() - 40 : 0;
  for (size_t i = from; i < w.trace.size(); ++i) printf("  %s\n", w.trace[i].c_str());
  exit(1);
}
static inline void note(World &w, const char *what) { if (!w.bfs) w.trace.push_back(what); }
#define REQUIRE(w, cond, what) do { if (!(cond)) fail(w, what); } while (0)

static bool truly_uniform(const World &w) {  // every env on the same day of an episode of the same length, same `finished`
  for (int i = 0; i < w.ne; ++i)
    if (w.fin[i] != w.fin[0] || w.day[i] != w.day[0] || w.nd[i] != w.nd[0]) return false;
  return true;
}
static bool truly_lockstep(const World &w, int32_t *day) {  // ... and nobody finished: what a known day means
  if (!truly_uniform(w) || w.fin[0]) return false;
  *day = w.day[0];
  return true;
}

struct StubDev {
  World &w;
  void pack_state() {
    REQUIRE(w, w.canon_id == w.latest, "stale read: k_pack_state reads canonical words that are not current");
    REQUIRE(w, truly_uniform(w) && w.nd[0] == w.bk.uni_nd, "false lock step: k_pack_state packs a batch that is not on one day");
    for (int i = 0; i < w.ne; ++i) REQUIRE(w, w.budget[i] <= 65535, "packed budgets: budget above 65535 packed into the 16-bit mirror");
    REQUIRE(w, w.static_ok, "packed form used although the tables forbid it");
    w.packed_id = w.latest;
    w.pk_day_val = w.day[0];
    note(w, "    [k_pack_state]");
  }
  void unpack_state(int32_t n_days) {
    REQUIRE(w, w.packed_id == w.latest, "stale read: k_unpack_state reads a mirror that is not current");
    REQUIRE(w, w.pk_day_val != POISON, "stale read: k_unpack_state reads a poisoned mirror");
    for (int i = 0; i < w.ne; ++i) {
      REQUIRE(w, w.day[i] == w.pk_day_val, "false lock step: k_unpack_state restores a day the envs are not on");
      REQUIRE(w, w.nd[i] == n_days, "false lock step: k_unpack_state restores a wrong episode length");
    }
    w.canon_id = w.latest;
    note(w, "    [k_unpack_state]");
  }
  void poison_mirror() {
    w.pk_day_val = POISON;
    note(w, "    [k_poison_mirror]");
  }
};

static void read_canon(World &w, const char *who) {
  if (w.canon_id != w.latest) fail(w, (std::string("stale read: ") + who + " reads canonical words that are not current").c_str());
}
static void write_canon(World &w) { w.canon_id = w.latest = ++w.clock; }
// ragged tables in the exhaustive driver: env i always draws a (county, year) of length 2 + (i & 1)
static int32_t table_len(World &w, int i) { return w.uni_nd > 0 ? w.uni_nd : (int32_t)(w.bfs ? 2 + (i & 1) : 3 + w.rng->below(6)); }
static int64_t table_b0(World &w) { return w.bfs ? (int64_t)w.b0_max : w.rng->below((int64_t)w.b0_max + 1); }
// exhaustive driver: a budget is represented by the SMALLEST value of its class (what the code compares budgets with is
// 65535 and sums of reset arguments; rounding down never puts a budget above a bound it was under): 0, [1, 20] -> 5 (no
// smaller positive value occurs), (20, 65535] -> 21, above -> 65536. A centred draw takes 60000 to 90000 = the class
// above: the way a sticky random walk leaves the 16-bit range is in the space.
static int64_t snap(const World &w, int64_t b) {
  if (!w.bfs || b <= 0) return b;
  return b <= 20 ? 5 : (b <= 65535 ? 21 : 65536);
}

// draw_episode (csrc/w2a_common.hip.h) for env i: budget rules of env.py:167-178
static void new_episode(World &w, int i, const Cfg &c) {
  int64_t b = (c.sticky && w.sticky[i] >= 0) ? w.sticky[i] : (c.budget_kw < 0 ? table_b0(w) : c.budget_kw);
  if (b < 0) b = 0;
  if (c.mode == 1) b = w.bfs ? b : w.rng->below(b + 1);  // (exhaustive: the largest draw, the case that matters for a bound)
  else if (c.mode == 2) {
    int64_t lo = b / 2, hi = (int64_t)(1.5 * (double)b + 1.0);
    b = w.bfs ? hi - 1 : lo + w.rng->below(hi - lo);
  }
  b = snap(w, b);
  w.budget[i] = b;
  w.sticky[i] = c.sticky ? b : -1;
  w.day[i] = 0; w.fin[i] = false; w.nd[i] = table_len(w, i); w.episode[i]++;
}

static void check_invariants(World &w) {
  REQUIRE(w, w.bk.pk_valid || w.bk.canon_valid, "no valid form: neither the canonical words nor the mirror is marked current");
  if (w.bk.canon_valid) REQUIRE(w, w.canon_id == w.latest, "stale flag: canonical words marked current, but they are not");
  if (w.bk.pk_valid) {
    REQUIRE(w, w.packed_id == w.latest, "stale flag: mirror marked current, but it is not");
    REQUIRE(w, truly_uniform(w) && w.pk_day_val == w.day[0], "false lock step: the mirror's day word is not the day the envs are on");
  }
  if (w.bk.uni_t >= 0) {
    int32_t d = -1;
    REQUIRE(w, truly_lockstep(w, &d) && d == w.bk.uni_t, "false lock step: W2A_Q_LOCKSTEP_DAY is not the day the envs are on");
    REQUIRE(w, w.bk.lock, "false lock step: a day is claimed without lock step");
  }
  if (w.bk.lock) REQUIRE(w, truly_uniform(w) && w.nd[0] == w.bk.uni_nd, "false lock step: W2A_Q_LOCKSTEP although the envs are not on one day");
  if (w.bk.perm_valid) REQUIRE(w, w.perm_for == w.epoch, "stale grouping: perm_valid although env indices hold other episodes");
  if (w.bk.rm_valid) REQUIRE(w, w.rm_for == w.epoch && w.rm_order_gen == w.order_gen,
                             "stale grouping: rm_valid although the tile list belongs to other episodes / another order");
  if (w.bk.hist_valid) REQUIRE(w, w.hist_for == w.epoch, "stale grouping: hist_valid although the row counts belong to other episodes");
  REQUIRE(w, !!w.bk.poisoned == (w.pk_day_val == POISON), "the handle's idea of the poison differs from the mirror's day word");
  if (w.bk.budget_bound != W2A_BK_UNKNOWN)  // what every use of the 16-bit mirror rests on
    for (int i = 0; i < w.ne; ++i)
      REQUIRE(w, w.budget[i] <= w.bk.budget_bound && w.sticky[i] <= w.bk.budget_bound,
              "packed budgets: the handle's budget bound is below a budget (current or sticky) the state buffer holds");
  // a replay may come between any two API calls: what a recorded kernel would step must be current (or poisoned)
  if (w.graphs & G_ANY_PACKED)
    REQUIRE(w, w.pk_day_val == POISON || (w.packed_id == w.latest && truly_uniform(w) && w.pk_day_val == w.day[0]),
            "unsafe replay: a recorded packed step would step a mirror that is neither current nor poisoned");
  if (w.graphs & (G_CANON | G_CANON_AUTO_SAME | G_CANON_AUTO_NEXT))
    REQUIRE(w, w.canon_id == w.latest, "unsafe replay: a recorded canonical step would step canonical words that are not current");
}

// ---------------------------------------------------------------- entry points (bookkeeping side of w2a_kernels.hip)
static void end_call(World &w) { StubDev d{w}; bk_end_call(w.bk, d); }

static void api_reset_device(World &w, const Cfg &c, bool masked, unsigned sel, bool launch_fails = false) {  // w2a_reset_device_rng + launch_reset
  bk_note_budgets(w.bk, c.budget_kw >= 0 ? c.budget_kw : w.bk.b0_max, c.mode == 2, c.sticky != 0);
  StubDev d{w};
  const W2aBook before = w.bk;
  bk_reset(w.bk, d, false, masked);
  if (launch_fails) {  // (the budget note above stays: conservative)
    bk_reset_rollback(w.bk, before, false, masked);
    note(w, "    (the launch failed: rolled back)");
    end_call(w);
    return;
  }
  if (masked) read_canon(w, "k_reset (masked)");
  for (int i = 0; i < w.ne; ++i)
    if (!masked || ((sel >> i) & 1u)) new_episode(w, i, c);
  w.epoch = ++w.clock;
  if (w.bk.hist_valid) {  // launch_reset: k_reset also counts rows / ranks envs -- those it selects
    REQUIRE(w, !masked, "stale grouping: a masked k_reset left row counts of the selected envs only");
    w.hist_for = w.epoch;
  }
  write_canon(w);
  end_call(w);
}
static void api_reset_tuples(World &w, bool with_budgets, int64_t bmax, bool masked, unsigned sel, bool tell_bound) {  // w2a_reset
  bk_note_budgets(w.bk, with_budgets ? -1 : w.bk.b0_max, false, false);
  StubDev d{w};
  bk_reset(w.bk, d, false, masked);
  if (masked) read_canon(w, "k_reset (masked)");
  int64_t seen = 0;
  for (int i = 0; i < w.ne; ++i)
    if (!masked || ((sel >> i) & 1u)) {
      w.budget[i] = snap(w, with_budgets ? (w.bfs ? bmax : w.rng->below(bmax + 1)) : table_b0(w));
      if (w.budget[i] > seen) seen = w.budget[i];
      w.day[i] = 0; w.fin[i] = false; w.nd[i] = table_len(w, i); w.episode[i]++;  // the sticky budget stays (cold.z)
    }
  w.epoch = ++w.clock;
  if (w.bk.hist_valid) {
    REQUIRE(w, !masked, "stale grouping: a masked k_reset left row counts of the selected envs only");
    w.hist_for = w.epoch;
  }
  write_canon(w);
  end_call(w);
  if (with_budgets && seen > w.unstated) w.unstated = seen;
  if (tell_bound) {  // HeatAlertVecEnv._reset_tuples: w2a_set_budget_bound(max of the budgets handed over -- by contract of
    seen = w.unstated > seen ? w.unstated : seen;  // every hand-over since the bound was last known, include/w2a.h)
    w.unstated = 0;
    bk_set_budget_bound(w.bk, seen);
    if (!w.bfs) w.trace.push_back("    w2a_set_budget_bound(" + std::to_string(seen) + ")");
  }
}
static void api_observe(World &w, bool launch_fails = false) {  // w2a_observe
  StubDev d{w};
  const W2aBook before = w.bk;
  bk_reset(w.bk, d, true, false);
  if (launch_fails) bk_reset_rollback(w.bk, before, true, false);
  else read_canon(w, "k_reset (observe)");
  end_call(w);
}
static void api_set_autoreset(World &w, const Cfg &c) {  // w2a_set_autoreset
  w.acfg = c; w.has_autoreset = true;
  bk_set_autoreset(w.bk, c.budget_kw >= 0 ? c.budget_kw : w.bk.b0_max, c.mode == 2, c.sticky != 0);
}
static void advance(World &w, int i) {  // one day of env.py:256-260
  if (w.day[i] + 1 >= w.nd[i]) w.fin[i] = true; else w.day[i]++;
}
static void new_episode(World &w, int i, const Cfg &c);
// k_step64<..., PACKED[, AUTORESET]>: eager or replayed. In lock step the envs finish, and restart, together.
static int kind_index(int kind) { int i = 0; while ((1 << i) != kind) ++i; return i; }
static void packed_kernel(World &w, bool autoreset, bool next_step, const char *who, const Cfg *cfg = nullptr) {
  const Cfg &ac = cfg ? *cfg : w.acfg;
  if (w.pk_day_val == POISON) { note(w, "    (poisoned mirror: W2A_ST_STALE_GRAPH, nothing stepped)"); return; }
  REQUIRE(w, w.packed_id == w.latest, (std::string("stale read: ") + who + " reads a mirror that is not current").c_str());
  REQUIRE(w, truly_uniform(w) && w.pk_day_val == w.day[0] && w.nd[0] == w.bk.uni_nd,
          "false lock step: the packed step kernel finds a day / length in the mirror the envs are not on");
  for (int i = 0; i < w.ne; ++i) REQUIRE(w, w.budget[i] <= 65535, "packed budgets: budget above 65535 in the 16-bit mirror");
  REQUIRE(w, w.static_ok, "packed form used although the tables forbid it");
  bool changed = false;
  for (int i = 0; i < w.ne; ++i) {
    if (autoreset && next_step && w.fin[i]) { new_episode(w, i, ac); changed = true; continue; }
    advance(w, i);
    if (autoreset && !next_step && w.fin[i]) { new_episode(w, i, ac); changed = true; }
  }
  if (changed) {  // the epilogue packs the new episodes' words itself
    w.epoch = ++w.clock;
    for (int i = 0; i < w.ne; ++i) REQUIRE(w, w.budget[i] <= 65535, "packed budgets: an in-kernel autoreset packs a budget above 65535");
  }
  w.pk_day_val = w.day[0];  // the owning wave writes the tile's day word back (0 after a restart)
  w.packed_id = w.latest = ++w.clock;
}
static void canon_kernel(World &w, bool autoreset, bool next_step, const char *who, const Cfg *cfg = nullptr) {  // k_step / k_step64 on the canonical words
  const Cfg &ac = cfg ? *cfg : w.acfg;
  read_canon(w, who);
  bool changed = false;
  for (int i = 0; i < w.ne; ++i) {
    if (autoreset && next_step && w.fin[i]) { new_episode(w, i, ac); changed = true; continue; }
    advance(w, i);  // a finished env repeats its last day (env.py:256: done again)
    if (autoreset && !next_step && w.fin[i]) { new_episode(w, i, ac); changed = true; }
  }
  if (changed) w.epoch = ++w.clock;
  write_canon(w);
}
// w2a_step. launch_fails: hipLaunchKernel reports an error (conversions that bk_step asked for did run)
static void api_step(World &w, bool wide, bool autoreset, bool next_step, bool given, bool unpacked, bool capturing, bool launch_fails) {
  StubDev d{w};
  const W2aBook before = w.bk;
  const BkStepPlan p = bk_step(w.bk, d, wide, autoreset, given, unpacked, capturing);
  if (p.kernel < 0) { note(w, "    (refused: the capture would have to record a conversion)"); return; }
  if (launch_fails) {
    bk_step_rollback(w.bk, before, p);
    note(w, "    (the launch failed: rolled back)");
    end_call(w);
    return;
  }
  if (p.kernel == W2A_BK_STEP_PACKED) REQUIRE(w, !given, "the packed kernel has no REWARD_GIVEN variant");
  if (capturing) {  // recorded, not executed
    REQUIRE(w, p.converted == 0, "a conversion of the state's form was recorded into a hipGraph");
    int kind;
    if (p.kernel == W2A_BK_STEP_PACKED) kind = !autoreset ? G_PACKED : (next_step ? G_PACKED_AUTO_NEXT : G_PACKED_AUTO_SAME);
    else kind = !autoreset ? G_CANON : (next_step ? G_CANON_AUTO_NEXT : G_CANON_AUTO_SAME);
    w.graphs |= kind;
    w.gcfg[kind_index(kind)] = w.acfg;  // (one recording per kind is kept: a later one of the same kind replaces it)
    if (p.kernel == W2A_BK_STEP_PACKED) end_call(w);
    return;
  }
  if (p.kernel == W2A_BK_STEP_PACKED) {
    REQUIRE(w, w.pk_day_val != POISON, "the packed step kernel is launched on a poisoned mirror");
    packed_kernel(w, autoreset, next_step, "the packed step kernel");
  } else {
    canon_kernel(w, autoreset, next_step, "the step kernel");
  }
  end_call(w);
}
static void api_graph_replay(World &w, int kind) {  // hipGraphLaunch of a recorded step kernel: no host bookkeeping runs
  const Cfg *c = &w.gcfg[kind_index(kind)];
  if (kind & G_ANY_PACKED) packed_kernel(w, kind != G_PACKED, kind == G_PACKED_AUTO_NEXT, "a replayed packed step kernel", c);
  else canon_kernel(w, kind != G_CANON, kind == G_CANON_AUTO_NEXT, "a replayed canonical step kernel", c);
}
static void api_rollout(World &w, int32_t n_steps, bool fixes) {  // w2a_rollout
  StubDev d{w};
  const bool lock = bk_rollout_begin(w.bk, d, n_steps);
  const int k = bk_rollout_kernel(w.bk, lock, fixes, true, true);
  read_canon(w, "the rollout kernel");
  if (k == W2A_BK_ROLLOUT_MFMA) {
    REQUIRE(w, truly_uniform(w), "false lock step: the matrix-core rollout needs the batch in lock step");
    REQUIRE(w, w.rm_for == w.epoch && w.rm_order_gen == w.order_gen && !fixes,
            "stale grouping: the matrix-core rollout runs on a tile list of other episodes / another order");
  }
  for (int i = 0; i < w.ne; ++i)
    for (int s = 0; s < n_steps && !w.fin[i]; ++s) advance(w, i);
  write_canon(w);
  end_call(w);
}
static void api_get_state(World &w) {  // w2a_get_state
  StubDev d{w};
  bk_ensure_canonical(w.bk, d);
  read_canon(w, "k_get_state");
  end_call(w);
}
static void api_sort(World &w) {  // w2a_sort_episodes
  StubDev d{w};
  bk_sort(w.bk, d);
  read_canon(w, "k_permute_state");
  for (int i = 0; i + 1 < w.ne; i += 2) {  // a relabelling: whole records swap places
    std::swap(w.day[i], w.day[i + 1]); std::swap(w.nd[i], w.nd[i + 1]); std::swap(w.fin[i], w.fin[i + 1]);
    std::swap(w.budget[i], w.budget[i + 1]); std::swap(w.sticky[i], w.sticky[i + 1]); std::swap(w.episode[i], w.episode[i + 1]);
  }
  w.epoch = ++w.clock;
  write_canon(w);
  end_call(w);
}
static void api_group(World &w) {  // w2a_group_by_column (reads stepc / cold: never stale)
  w.perm_for = w.epoch;
  bk_grouped(w.bk);
}
static void api_pm_reward(World &w) {  // w2a_posterior_mean_reward
  if (!w.bk.perm_valid) { note(w, "    (refused: grouping stale)"); return; }
  REQUIRE(w, w.perm_for == w.epoch, "stale grouping: the posterior-mean reward runs on a grouping of other episodes");
  StubDev d{w};
  bk_ensure_canonical(w.bk, d);
  read_canon(w, "k_pm_prep");
  end_call(w);
}
static void api_order_attach(World &w) { bk_order_attach(w.bk); w.hist_for = -1; }  // w2a_rollout_order_attach (a new workspace)
static void api_rollout_order(World &w) {  // w2a_rollout_order (reads cold: never stale)
  if (!w.bk.has_order_ws) api_order_attach(w);
  if (!w.bk.hist_valid) w.hist_for = w.epoch;  // k_order_rank counts now
  REQUIRE(w, w.hist_for == w.epoch, "stale grouping: the order is placed from row counts of other episodes");
  w.order_exists = 1; w.order_gen = ++w.clock;
  bk_order_set(w.bk);
}
static void api_rm_prepare(World &w) {  // w2a_rollout_mfma_prepare (needs an order)
  if (!w.bk.has_order) return;
  w.rm_for = w.epoch; w.rm_order_gen = w.order_gen;
  bk_rm_prepared(w.bk);
}
// the caller restored a checkpoint of the canonical part, then w2a_invalidate
static void api_invalidate(World &w, bool tell, int shape) {
  for (int i = 0; i < w.ne; ++i) {
    if (w.bfs) {  // shape bit 0: the envs on different days, bit 1: a sticky budget above 65535 somewhere
      w.day[i] = (shape & 1) ? (i & 1) : 0; w.nd[i] = table_len(w, i); w.fin[i] = false;
      w.budget[i] = 5; w.sticky[i] = ((shape & 2) && i == 0) ? 65536 : -1;  // the large budget hides as a STICKY one (r4 finding 4)
    } else {
      w.day[i] = (int32_t)w.rng->below(3); w.nd[i] = table_len(w, i); w.fin[i] = w.rng->coin(10);
      w.budget[i] = w.rng->below(w.rng->coin(20) ? 100000 : 12); w.sticky[i] = w.rng->coin(50) ? w.budget[i] : -1;
    }
    if (w.day[i] >= w.nd[i]) w.day[i] = w.nd[i] - 1;
    w.episode[i]++;
  }
  w.epoch = ++w.clock;
  write_canon(w);
  w.pk_day_val = 0;  // the caller's copy covers the whole buffer, the mirror's day words included: some old day, not the poison
  bk_invalidate(w.bk);
  {  // w2a_invalidate scans the restored buffer itself (k_budget_scan): largest budget, current and sticky
    int64_t m = 0;
    for (int i = 0; i < w.ne; ++i) { if (w.budget[i] > m) m = w.budget[i]; if (w.sticky[i] > m) m = w.sticky[i]; }
    bk_set_budget_bound(w.bk, m);
    w.unstated = 0;
  }
  end_call(w);
  if (tell) bk_set_budget_bound(w.bk, w.rng->below(6));  // a caller's (possibly smaller) statement changes nothing
}

// ---------------------------------------------------------------- operations, shared by both drivers
enum OpKind { OP_STEP, OP_RESET_DEVICE, OP_RESET_TUPLES, OP_ROLLOUT, OP_GET_STATE, OP_SORT, OP_GROUP, OP_PM_REWARD, OP_OBSERVE,
              OP_INVALIDATE, OP_REPLAY, OP_SET_AUTORESET, OP_ORDER_ATTACH, OP_SET_BOUND, N_OPKINDS };
struct Op {
  int kind = OP_STEP;
  // step
  bool wide = false, autoreset = false, next = false, given = false, unpacked = false, capturing = false, fails = false;
  // resets
  Cfg cfg{-1, 0, 1};
  bool masked = false; unsigned sel = 0; bool with_budgets = false, tell = false; int64_t bmax = 9;
  // rollout
  int32_t n_steps = 1; bool fixes = false; int prep = 0;  // prep: 0 nothing, 1 new order, 2 new order + tile list
  int shape = 0, graph = 0; int64_t bound = 0;
};
static std::string describe(const Op &o) {
  char b[240];
  switch (o.kind) {
    case OP_STEP: snprintf(b, sizeof b, "step(wide %d autoreset %d next %d given %d unpacked %d%s%s)", o.wide, o.autoreset, o.next, o.given,
                           o.unpacked, o.capturing ? " CAPTURING" : "", o.fails ? " LAUNCH FAILS" : ""); break;
    case OP_RESET_DEVICE: snprintf(b, sizeof b, "reset_device(kw %lld mode %d sticky %d masked %d sel %u%s)", (long long)o.cfg.budget_kw,
                                   o.cfg.mode, o.cfg.sticky, o.masked, o.sel, o.fails ? " LAUNCH FAILS" : ""); break;
    case OP_RESET_TUPLES: snprintf(b, sizeof b, "reset_tuples(budgets %d max %lld masked %d sel %u tell %d)", o.with_budgets,
                                   (long long)o.bmax, o.masked, o.sel, o.tell); break;
    case OP_ROLLOUT: snprintf(b, sizeof b, "rollout(%d, fixes %d, prep %d)", o.n_steps, o.fixes, o.prep); break;
    case OP_GET_STATE: return "get_state";
    case OP_SORT: return "sort";
    case OP_GROUP: return "group_by_column";
    case OP_PM_REWARD: return "posterior_mean_reward";
    case OP_OBSERVE: return o.fails ? "observe LAUNCH FAILS" : "observe";
    case OP_INVALIDATE: snprintf(b, sizeof b, "checkpoint restore (shape %d); invalidate%s", o.shape, o.tell ? "; set_budget_bound" : ""); break;
    case OP_REPLAY: snprintf(b, sizeof b, "graph replay (kind %d)", o.graph); break;
    case OP_SET_AUTORESET: snprintf(b, sizeof b, "set_autoreset(kw %lld mode %d sticky %d)", (long long)o.cfg.budget_kw, o.cfg.mode, o.cfg.sticky); break;
    case OP_ORDER_ATTACH: return "rollout_order_attach (another workspace)";
    case OP_SET_BOUND: snprintf(b, sizeof b, "set_budget_bound(%lld)", (long long)o.bound); break;
    default: return "?";
  }
  return b;
}
static void print_path(const World &w) {
  for (int i = 0; i < w.path_len; ++i) printf("  %s\n", describe((*g_ops)[w.path[i]]).c_str());
}
static void apply(World &w, const Op &o) {
  if (!w.bfs) w.trace.push_back(describe(o));
  switch (o.kind) {
    case OP_STEP:
      if (o.given) api_pm_reward(w);
      api_step(w, o.wide, o.autoreset && w.has_autoreset, o.next, o.given, o.unpacked, o.capturing, o.fails);
      break;
    case OP_RESET_DEVICE: api_reset_device(w, o.cfg, o.masked, o.sel, o.fails); if (!o.fails) api_set_autoreset(w, o.cfg); break;
    case OP_RESET_TUPLES: api_reset_tuples(w, o.with_budgets, o.bmax, o.masked, o.sel, o.tell); break;
    case OP_ROLLOUT:
      if (o.prep >= 1) { api_rollout_order(w); if (o.prep >= 2) api_rm_prepare(w); }
      api_rollout(w, o.n_steps, o.fixes);
      break;
    case OP_GET_STATE: api_get_state(w); break;
    case OP_SORT: api_sort(w); break;
    case OP_GROUP: api_group(w); break;
    case OP_PM_REWARD: api_pm_reward(w); break;
    case OP_OBSERVE: api_observe(w, o.fails); break;
    case OP_INVALIDATE: api_invalidate(w, o.tell, o.shape); break;
    case OP_REPLAY: if (w.graphs & o.graph) api_graph_replay(w, o.graph); break;
    case OP_SET_AUTORESET: api_set_autoreset(w, o.cfg); break;
    case OP_ORDER_ATTACH: api_order_attach(w); break;
    case OP_SET_BOUND: {  // a caller's statement is true by contract: -2 = exactly the largest budget the envs hold now
      int64_t m = 0;
      for (int i = 0; i < w.ne; ++i) if (w.budget[i] > m) m = w.budget[i];
      bk_set_budget_bound(w.bk, o.bound == -2 ? m : o.bound);
      if (o.bound != -1) w.unstated = 0;
      break;
    }
  }
  check_invariants(w);
}
static void init_world(World &w, Chooser *rng, int32_t uni_nd, int32_t b0_max, bool static_ok) {
  for (Cfg &c : w.gcfg) c = Cfg{-1, 0, 1};
  w.rng = rng; w.uni_nd = uni_nd; w.b0_max = b0_max; w.static_ok = static_ok;
  bk_init(w.bk, static_ok, uni_nd, b0_max);
  for (int i = 0; i < w.ne; ++i) { w.day[i] = 0; w.nd[i] = 1; w.fin[i] = true; w.budget[i] = 0; w.sticky[i] = -1; w.episode[i] = -1; }
}

// ---------------------------------------------------------------- random driver
static Cfg random_cfg(Chooser &r) {
  Cfg c;
  const int u = (int)r.below(10);
  c.budget_kw = u < 5 ? -1 : (u < 8 ? r.below(9) : 60000 + r.below(20000));
  c.mode = (int)r.below(3);
  c.sticky = r.coin(70) ? 1 : 0;
  return c;
}
static void run_sequence(uint64_t seed, int n_ops) {
  Chooser rng;
  rng.s = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  World w;
  init_world(w, &rng, rng.coin(75) ? (int32_t)(2 + rng.below(7)) : -1, rng.coin(80) ? (int32_t)(1 + rng.below(9)) : 70000, rng.coin(90));
  w.trace.push_back("sequence " + std::to_string(seed) + ": uni_nd " + std::to_string(w.uni_nd) + ", b0_max " +
                    std::to_string(w.b0_max) + ", static_ok " + std::to_string((int)w.static_ok));
  Op first; first.kind = OP_RESET_DEVICE; first.cfg = random_cfg(rng);
  apply(w, first);
  for (int op = 0; op < n_ops; ++op) {
    const int u = (int)rng.below(100);
    Op o;
    const unsigned all = (1u << w.ne) - 1u;
    if (u < 40) {
      o.kind = OP_STEP;
      o.autoreset = w.has_autoreset && rng.coin(25); o.next = rng.coin(40); o.given = !o.autoreset && rng.coin(10);
      o.wide = o.given || rng.coin(70); o.unpacked = rng.coin(10); o.fails = rng.coin(3);
    } else if (u < 48) {
      o.kind = OP_RESET_DEVICE; o.cfg = random_cfg(rng); o.masked = rng.coin(40);
      o.sel = rng.coin(20) ? all : (unsigned)rng.below(all + 1); o.fails = rng.coin(4);
    } else if (u < 54) {
      o.kind = OP_RESET_TUPLES; o.with_budgets = rng.coin(70); o.masked = rng.coin(40); o.tell = rng.coin(80);
      o.bmax = rng.coin(80) ? 9 : 90000; o.sel = (unsigned)rng.below(all + 1);
    } else if (u < 62) {
      o.kind = OP_ROLLOUT; o.n_steps = (int32_t)(1 + rng.below(9)); o.fixes = rng.coin(15);
      o.prep = rng.coin(60) ? (rng.coin(80) ? 2 : 1) : 0;
    } else if (u < 70) o.kind = OP_GET_STATE;
    else if (u < 73) o.kind = OP_SORT;
    else if (u < 78) o.kind = OP_GROUP;
    else if (u < 83) o.kind = OP_PM_REWARD;
    else if (u < 86) { o.kind = OP_OBSERVE; o.fails = rng.coin(10); }
    else if (u < 89) { o.kind = OP_INVALIDATE; o.tell = rng.coin(70); }
    else if (u < 93) {
      o.kind = OP_STEP; o.capturing = true;
      o.wide = rng.coin(70); o.autoreset = w.has_autoreset && rng.coin(40); o.next = rng.coin(40); o.unpacked = rng.coin(10);
    } else if (u < 97) { o.kind = OP_REPLAY; o.graph = 1 << rng.below(6); }
    else if (u < 99) { o.kind = OP_SET_AUTORESET; o.cfg = random_cfg(rng); }
    else o.kind = OP_ORDER_ATTACH;
    apply(w, o);
  }
}

// ---------------------------------------------------------------- exhaustive driver
// Everything that decides what any later operation does or checks, with the unbounded counters (clock, episode ids)
// replaced by the relations the checks read.
static std::string key_of(const World &w) {
  std::string k;
  k.reserve(96);
  auto put = [&k](int64_t v) {  // one byte where it fits (almost everything), nine otherwise
    if (v >= -100 && v < 100) k.push_back((char)(v + 100));
    else { k.push_back((char)255); k.append(reinterpret_cast<const char *>(&v), sizeof v); }
  };
  const W2aBook &b = w.bk;
  for (int64_t v : {(int64_t)b.pk_valid, (int64_t)b.canon_valid, (int64_t)b.lock, (int64_t)b.uni_t, b.budget_bound, b.budget_bound_known,
                    (int64_t)b.foreign, (int64_t)b.has_auto, b.auto_cand, (int64_t)b.auto_centered, (int64_t)b.auto_sticky,
                    (int64_t)b.graph_canon, (int64_t)b.graph_packed, (int64_t)b.graph_autoreset, (int64_t)b.poisoned,
                    b.graph_cand, (int64_t)b.graph_centered, (int64_t)b.graph_sticky,
                    (int64_t)b.perm_valid, (int64_t)b.has_order, (int64_t)b.rm_valid, (int64_t)b.has_order_ws, (int64_t)b.hist_valid})
    put(v);  // last_step_kernel / last_rollout_kernel: outputs only
  for (int i = 0; i < w.ne; ++i) { put(w.day[i]); put(w.nd[i]); put(w.fin[i]); put(w.budget[i]); put(w.sticky[i]); }
  put(w.canon_id == w.latest); put(w.packed_id == w.latest);
  put(w.pk_day_val == POISON ? 0 : (truly_uniform(w) && w.pk_day_val == w.day[0] ? 1 : 2));
  put(w.perm_for == w.epoch); put(w.rm_for == w.epoch && w.rm_order_gen == w.order_gen); put(w.hist_for == w.epoch);
  for (int k : {2, 3, 4, 5})  // recorded autoreset kinds: the parameters their replays draw with
    if (w.graphs & (1 << k)) { put(w.gcfg[k].budget_kw); put(w.gcfg[k].mode); put(w.gcfg[k].sticky); }
  put(w.unstated); put(w.order_exists); put(w.has_autoreset); put(w.acfg.budget_kw); put(w.acfg.mode); put(w.acfg.sticky); put(w.graphs);
  return k;
}
// walk 0 "budgets": every operation that touches budget knowledge (+ plain / autoreset steps, so that episodes end, restart
// inside the kernel and get packed). walk 1 "forms": everything about forms, lock step, graphs and grouping, with one reset
// configuration and small budgets -- known, or handed over in device memory with / without a stated bound.
static bool g_full = false;  // --bfs-full: the budgets walk also records and replays autoreset steps (7.8 M states, minutes)
static std::vector<Op> all_ops(int walk) {
  std::vector<Op> v;
  std::vector<Cfg> cfgs;
  if (walk == 0) {
    for (int64_t kw : {(int64_t)-1, (int64_t)60000})  // (a small budget_kw acts like the small table budgets; less_than draws
      for (int mode : {0, 2})                         // are bounded by the fixed budget they start from)
        for (int st = 0; st < 2; ++st) cfgs.push_back(Cfg{kw, mode, st});
  } else cfgs.push_back(Cfg{-1, 0, 1});
  if (walk == 0) {
    for (int ar = 0; ar < 3; ++ar) { Op o; o.kind = OP_STEP; o.wide = true; o.autoreset = ar > 0; o.next = ar == 2; v.push_back(o); }
    if (g_full) {  // --bfs-full: a recorded autoreset step keeps the parameters it was recorded with, its replays go on
      Op o; o.kind = OP_STEP; o.wide = true; o.autoreset = true; o.capturing = true; v.push_back(o);  // drawing budgets with
      for (int g : {(int)G_CANON_AUTO_SAME, (int)G_PACKED_AUTO_SAME}) { Op r; r.kind = OP_REPLAY; r.graph = g; v.push_back(r); }  // them
    }
  } else {
    for (int form = 0; form < 3; ++form)      // 4-lanes-per-env kernel; 64-envs-per-wave; the latter with W2A_STEP_UNPACKED
      for (int ar = 0; ar < 3; ++ar)          // none, same-step, next-step autoreset
        for (int mode = 0; mode < 3; ++mode) {  // eager, capturing, eager with a failing launch
          Op o; o.kind = OP_STEP; o.wide = form > 0; o.unpacked = form == 2; o.autoreset = ar > 0; o.next = ar == 2;
          o.capturing = mode == 1; o.fails = mode == 2;
          v.push_back(o);
        }
    Op o; o.kind = OP_STEP; o.wide = true; o.given = true; v.push_back(o); o.fails = true; v.push_back(o);
  }
  for (const Cfg &c : cfgs)
    for (int m = 0; m < 3; ++m) {  // unmasked; a mask selecting env 0; a mask selecting both (the handle cannot see what a mask selects)
      Op o; o.kind = OP_RESET_DEVICE; o.cfg = c; o.masked = m > 0; o.sel = m == 1 ? 1u : 3u;
      v.push_back(o);
      if (walk == 1) { o.fails = true; v.push_back(o); }  // ... and the same reset with a k_reset launch that fails
    }
  if (walk == 1) { Op o; o.kind = OP_OBSERVE; o.fails = true; v.push_back(o); }
  for (int wb = 0; wb < 2; ++wb)
    for (int64_t bmax : {(int64_t)9, (int64_t)90000})
      for (int m = 0; m < 3; ++m)
        for (int tell = 0; tell < 2; ++tell) {
          if ((!wb && (tell || bmax != 9)) || (walk == 1 && bmax != 9)) continue;
          Op o; o.kind = OP_RESET_TUPLES; o.with_budgets = wb; o.bmax = bmax; o.masked = m > 0; o.sel = m == 1 ? 1u : 3u; o.tell = tell;
          v.push_back(o);
        }
  if (walk == 1)
    for (int32_t n : {1, 9})
      for (int fx = 0; fx < 2; ++fx)
        for (int prep = 0; prep < 3; ++prep) { Op o; o.kind = OP_ROLLOUT; o.n_steps = n; o.fixes = fx; o.prep = prep; v.push_back(o); }
  { Op o; o.kind = OP_GET_STATE; v.push_back(o); }
  if (walk == 1)
    for (int kind : {(int)OP_SORT, (int)OP_OBSERVE, (int)OP_GROUP, (int)OP_PM_REWARD, (int)OP_ORDER_ATTACH}) { Op o; o.kind = kind; v.push_back(o); }
  for (int shape = 0; shape < (walk == 0 ? 4 : 2); ++shape)
    for (int tell = 0; tell < (walk == 0 ? 2 : 1); ++tell) { Op o; o.kind = OP_INVALIDATE; o.shape = shape; o.tell = tell; v.push_back(o); }
  if (walk == 1)
    for (int g : {(int)G_PACKED, (int)G_CANON, (int)G_CANON_AUTO_SAME, (int)G_CANON_AUTO_NEXT, (int)G_PACKED_AUTO_SAME, (int)G_PACKED_AUTO_NEXT}) {
      Op o; o.kind = OP_REPLAY; o.graph = g; v.push_back(o);
    }
  if (walk == 0)
    for (const Cfg &c : cfgs) { Op o; o.kind = OP_SET_AUTORESET; o.cfg = c; v.push_back(o); }
  for (int64_t bd : {(int64_t)-1, (int64_t)-2, (int64_t)100000}) { Op o; o.kind = OP_SET_BOUND; o.bound = bd; v.push_back(o); }
  return v;
}
static int run_bfs(int max_depth) {
  size_t total_states = 0, total_edges = 0;
  int deepest = 0;
  bool closed = true;
  for (int walk = 0; walk < 2; ++walk) {
    const std::vector<Op> ops = all_ops(walk);
    g_ops = &ops;
    for (int cfg = 0; cfg < 8; ++cfg) {  // the table properties a handle is created with
      // (the budgets walk runs one-day episodes: every step is a terminal step, every autoreset step a new draw)
      const int32_t uni_nd = (cfg & 1) ? -1 : (walk == 0 ? 1 : 2), b0_max = (cfg & 2) ? 70000 : 9;
      const bool static_ok = !(cfg & 4);
      if (walk == 1 && (cfg & 2)) continue;          // the forms walk keeps budgets small
      if (walk == 0 && (cfg & 5)) continue;          // budgets only matter to handles that can pack at all
      if (walk == 1 && (cfg & 5) == 5) continue;     // ragged tables never pack, whatever their dimensions
      Chooser ch;
      ch.scripted = true;
      World w0;
      w0.ne = 2; w0.bfs = true;
      init_world(w0, &ch, uni_nd, b0_max, static_ok);
      check_invariants(w0);
      std::unordered_set<std::string> seen;
      std::deque<std::pair<World, int>> queue;
      seen.insert(key_of(w0));
      queue.emplace_back(w0, 0);
      size_t edges = 0;
      while (!queue.empty()) {
        const World cur = queue.front().first;
        const int depth = queue.front().second;
        queue.pop_front();
        if (depth > deepest) deepest = depth;
        if ((max_depth > 0 && depth >= max_depth) || cur.path_len >= 47) { closed = false; continue; }
        for (size_t oi = 0; oi < ops.size(); ++oi) {
          const Op &o = ops[oi];
          if (o.kind == OP_REPLAY && !(cur.graphs & o.graph)) continue;
          if (o.kind == OP_STEP && o.autoreset && !cur.has_autoreset) continue;
          ch.script.clear(); ch.arity.clear();
          do {  // every outcome of the operation's own choices
            ch.rewind();
            World nx = cur;
            nx.rng = &ch;
            nx.path[nx.path_len++] = (uint16_t)oi;
            apply(nx, o);  // exits with the path on a violation
            ++edges;
            std::string k = key_of(nx);
            if (seen.insert(std::move(k)).second) queue.emplace_back(std::move(nx), depth + 1);
          } while (ch.advance());
        }
      }
      printf("  walk %-8s tables (uni_nd %2d, b0_max %5d, static_ok %d): %8zu reachable states, %10zu transitions checked\n",
             walk ? "forms" : "budgets", uni_nd, b0_max, (int)static_ok, seen.size(), edges);
      fflush(stdout);
      total_states += seen.size(); total_edges += edges;
    }
  }
  printf("bookkeeping_check --bfs: %zu reachable abstract states, %zu transitions, depth %d, %s, no violation\n", total_states,
         total_edges, deepest, closed ? "walked to closure" : "CUT at the depth limit");
  return closed || max_depth > 0 ? 0 : 2;
}

int main(int argc, char **argv) {
  if (argc > 1 && (!strcmp(argv[1], "--bfs") || !strcmp(argv[1], "--bfs-full"))) {
    g_full = !strcmp(argv[1], "--bfs-full");
    return run_bfs(argc > 2 ? atoi(argv[2]) : 0);
  }
  const int n_seq = argc > 1 ? atoi(argv[1]) : 2000;
  const int n_ops = argc > 2 ? atoi(argv[2]) : 120;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
  for (int i = 0; i < n_seq; ++i) run_sequence(seed * 1000003ull + (uint64_t)i, n_ops);
  printf("bookkeeping_check: %d sequences x %d operations, no violation\n", n_seq, n_ops);
  return 0;
}
