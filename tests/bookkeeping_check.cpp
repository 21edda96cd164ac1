// bookkeeping_check.cpp -- the handle's host-side bookkeeping (weather2alert_amd/csrc/w2a_bookkeeping.h, the very header
// libw2a.so compiles) on the CPU against a recording stub that KNOWS what is really the case: a handful of simulated envs
// (day, finished, episode length, episode id), for each of the two forms of the per-env step state
// which version of the contents it holds, what the mirror's day word says, and which kinds of step were recorded into
// hipGraphs (a replay runs a recorded kernel with NO bookkeeping at all). Budgets are not part of the model: since round 6
// the packed kernel serves any budget by itself (csrc/w2a_common.hip.h, pk_budget16) and the header knows nothing of them --
// the "budgets" walk, its invariant and its mutants went with the eleven fields they guarded. Built by tests/test_bookkeeping_cpu.py with
// g++ -fsanitize=address,undefined; the same test also builds mutants of the header (one rule broken each) and requires
// this program to catch every one of them.
//
// Two drivers over the same operations:
//   random      bookkeeping_check <sequences> <ops per sequence> <seed>      six envs, free parameters
//   exhaustive  bookkeeping_check --bfs [max depth, 0 = closure] breadth-first walk of the ABSTRACT state space to closure,
//               three envs, episodes of three days (two and three on ragged tables): every flag of W2aBook about the two forms of the state, lock step, recorded
//               graphs and the validity of column grouping / visiting order / tile list / row counts x what is really
//               current x the day structure of the batch x the graphs recorded. Every operation with every parameter
//               and every outcome of its internal choices from every reachable state; prints the number of reachable
//               states. Nothing is sampled.
//
// The entry points below restate, call for call, what csrc/w2a_kernels.hip / w2a_step_dispatch.hip.h do around their
// kernel launches (each names the function it follows); a launch becomes "reads form X" / "writes form X".
//
// Violations reported:
//   stale read        a kernel reads a form of the state that does not hold the latest contents
//   false lock step   the handle claims lock step / a day the envs are not on, or packs a batch that is not on one day
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

enum { G_PACKED = 1, G_CANON = 2, G_CANON_AUTO_SAME = 4, G_CANON_AUTO_NEXT = 8, G_PACKED_AUTO_SAME = 16, G_PACKED_AUTO_NEXT = 32 };
static const int G_ANY_PACKED = G_PACKED | G_PACKED_AUTO_SAME | G_PACKED_AUTO_NEXT;

struct World {
  int ne = NE;
  bool bfs = false;      // exhaustive driver: budgets are snapped to class representatives so that the space is finite
  // ---- truth
  int32_t day[MAXE], nd[MAXE];
  bool fin[MAXE];
  long episode[MAXE];
  long clock = 0, latest = 0, canon_id = 0, packed_id = -1;
  int64_t pk_day_val = -1;  // what the mirror's day words hold (POISON: poisoned)
  long epoch = 0;        // changes whenever any env index gets another episode
  long perm_for = -1, order_exists = 0, rm_for = -1, rm_order_gen = -1, order_gen = 0, hist_for = -1;
  int32_t uni_nd;        // table property
  bool static_ok;
  bool has_autoreset = false;
  int graphs = 0;        // kinds of recorded step (G_*)
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

// draw_episode (csrc/w2a_common.hip.h) for env i
static void new_episode(World &w, int i) {
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
  // a replay may come between any two API calls: what a recorded kernel would step must be current (or poisoned)
  if (w.graphs & G_ANY_PACKED)
    REQUIRE(w, w.pk_day_val == POISON || (w.packed_id == w.latest && truly_uniform(w) && w.pk_day_val == w.day[0]),
            "unsafe replay: a recorded packed step would step a mirror that is neither current nor poisoned");
  if (w.graphs & (G_CANON | G_CANON_AUTO_SAME | G_CANON_AUTO_NEXT))
    REQUIRE(w, w.canon_id == w.latest, "unsafe replay: a recorded canonical step would step canonical words that are not current");
}

// ---------------------------------------------------------------- entry points (bookkeeping side of w2a_kernels.hip)
static void end_call(World &w) { StubDev d{w}; bk_end_call(w.bk, d); }

// sorted: w2a_reset_device_rng_sorted (whole batch; the new episodes land in coefficient-row order, and its k_reset pass does
// not do the rank atomics of an attached order workspace)
static void api_reset_device(World &w, bool masked, unsigned sel, bool launch_fails = false, bool sorted = false) {  // w2a_reset_device_rng / w2a_reset + launch_reset
  StubDev d{w};
  const W2aBook before = w.bk;
  if (sorted) bk_reset_sorted(w.bk, d);
  else bk_reset(w.bk, d, false, masked);
  if (launch_fails) {
    bk_reset_rollback(w.bk, before, false, masked);
    note(w, "    (the launch failed: rolled back)");
    end_call(w);
    return;
  }
  if (masked) read_canon(w, "k_reset (masked)");
  for (int i = 0; i < w.ne; ++i)
    if (!masked || ((sel >> i) & 1u)) new_episode(w, i);
  w.epoch = ++w.clock;
  if (w.bk.hist_valid && !sorted) {  // launch_reset: k_reset also counts rows / ranks envs -- those it selects
    REQUIRE(w, !masked, "stale grouping: a masked k_reset left row counts of the selected envs only");
    w.hist_for = w.epoch;
  }
  write_canon(w);
  end_call(w);
}
static void api_observe(World &w, bool launch_fails = false) {  // w2a_observe
  StubDev d{w};
  const W2aBook before = w.bk;
  bk_reset(w.bk, d, true, false);
  if (launch_fails) bk_reset_rollback(w.bk, before, true, false);
  else read_canon(w, "k_reset (observe)");
  end_call(w);
}
static void api_set_autoreset(World &w) { w.has_autoreset = true; }  // w2a_set_autoreset (no bookkeeping: kernel arguments only)
static void advance(World &w, int i) {  // one day of env.py:256-260
  if (w.day[i] + 1 >= w.nd[i]) w.fin[i] = true; else w.day[i]++;
}
// k_step64<..., PACKED[, AUTORESET]>: eager or replayed. In lock step the envs finish, and restart, together.
static void packed_kernel(World &w, bool autoreset, bool next_step, const char *who) {
  if (w.pk_day_val == POISON) { note(w, "    (poisoned mirror: W2A_ST_STALE_GRAPH, nothing stepped)"); return; }
  REQUIRE(w, w.packed_id == w.latest, (std::string("stale read: ") + who + " reads a mirror that is not current").c_str());
  REQUIRE(w, truly_uniform(w) && w.pk_day_val == w.day[0] && w.nd[0] == w.bk.uni_nd,
          "false lock step: the packed step kernel finds a day / length in the mirror the envs are not on");
  REQUIRE(w, w.static_ok, "packed form used although the tables forbid it");
  bool changed = false;
  for (int i = 0; i < w.ne; ++i) {
    if (autoreset && next_step && w.fin[i]) { new_episode(w, i); changed = true; continue; }
    advance(w, i);
    if (autoreset && !next_step && w.fin[i]) { new_episode(w, i); changed = true; }
  }
  if (changed) w.epoch = ++w.clock;  // the epilogue packs the new episodes' words itself
  w.pk_day_val = w.day[0];  // the owning wave writes the tile's day word back (0 after a restart)
  w.packed_id = w.latest = ++w.clock;
}
static void canon_kernel(World &w, bool autoreset, bool next_step, const char *who) {  // k_step / k_step64 on the canonical words
  read_canon(w, who);
  bool changed = false;
  for (int i = 0; i < w.ne; ++i) {
    if (autoreset && next_step && w.fin[i]) { new_episode(w, i); changed = true; continue; }
    advance(w, i);  // a finished env repeats its last day (env.py:256: done again)
    if (autoreset && !next_step && w.fin[i]) { new_episode(w, i); changed = true; }
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
  if (kind & G_ANY_PACKED) packed_kernel(w, kind != G_PACKED, kind == G_PACKED_AUTO_NEXT, "a replayed packed step kernel");
  else canon_kernel(w, kind != G_CANON, kind == G_CANON_AUTO_NEXT, "a replayed canonical step kernel");
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
    std::swap(w.episode[i], w.episode[i + 1]);
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
static void api_invalidate(World &w, int shape) {
  for (int i = 0; i < w.ne; ++i) {
    if (w.bfs) {  // shape bit 0: the envs on different days
      w.day[i] = (shape & 1) ? (i % 3) : 0; w.nd[i] = table_len(w, i); w.fin[i] = false;
    } else {
      w.day[i] = (int32_t)w.rng->below(3); w.nd[i] = table_len(w, i); w.fin[i] = w.rng->coin(10);
    }
    if (w.day[i] >= w.nd[i]) w.day[i] = w.nd[i] - 1;
    w.episode[i]++;
  }
  w.epoch = ++w.clock;
  write_canon(w);
  w.pk_day_val = 0;  // the caller's copy covers the whole buffer, the mirror's day words included: some old day, not the poison
  bk_invalidate(w.bk);
  end_call(w);
}

// ---------------------------------------------------------------- operations, shared by both drivers
enum OpKind { OP_STEP, OP_RESET, OP_ROLLOUT, OP_GET_STATE, OP_SORT, OP_GROUP, OP_PM_REWARD, OP_OBSERVE,
              OP_INVALIDATE, OP_REPLAY, OP_SET_AUTORESET, OP_ORDER_ATTACH, N_OPKINDS };
struct Op {
  int kind = OP_STEP;
  // step
  bool wide = false, autoreset = false, next = false, given = false, unpacked = false, capturing = false, fails = false;
  // resets (w2a_reset_device_rng and w2a_reset look the same to the bookkeeping)
  bool masked = false; unsigned sel = 0; bool sorted = false;
  // rollout
  int32_t n_steps = 1; bool fixes = false; int prep = 0;  // prep: 0 nothing, 1 new order, 2 new order + tile list
  int shape = 0, graph = 0;
};
static std::string describe(const Op &o) {
  char b[240];
  switch (o.kind) {
    case OP_STEP: snprintf(b, sizeof b, "step(wide %d autoreset %d next %d given %d unpacked %d%s%s)", o.wide, o.autoreset, o.next, o.given,
                           o.unpacked, o.capturing ? " CAPTURING" : "", o.fails ? " LAUNCH FAILS" : ""); break;
    case OP_RESET: snprintf(b, sizeof b, "reset(masked %d sel %u%s%s)", o.masked, o.sel, o.sorted ? " SORTED" : "", o.fails ? " LAUNCH FAILS" : ""); break;
    case OP_ROLLOUT: snprintf(b, sizeof b, "rollout(%d, fixes %d, prep %d)", o.n_steps, o.fixes, o.prep); break;
    case OP_GET_STATE: return "get_state";
    case OP_SORT: return "sort";
    case OP_GROUP: return "group_by_column";
    case OP_PM_REWARD: return "posterior_mean_reward";
    case OP_OBSERVE: return o.fails ? "observe LAUNCH FAILS" : "observe";
    case OP_INVALIDATE: snprintf(b, sizeof b, "checkpoint restore (shape %d); invalidate", o.shape); break;
    case OP_REPLAY: snprintf(b, sizeof b, "graph replay (kind %d)", o.graph); break;
    case OP_SET_AUTORESET: return "set_autoreset";
    case OP_ORDER_ATTACH: return "rollout_order_attach (another workspace)";
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
    case OP_RESET: api_reset_device(w, o.masked, o.sel, o.fails, o.sorted); break;
    case OP_ROLLOUT:
      if (o.prep >= 1) { api_rollout_order(w); if (o.prep >= 2) api_rm_prepare(w); }
      api_rollout(w, o.n_steps, o.fixes);
      break;
    case OP_GET_STATE: api_get_state(w); break;
    case OP_SORT: api_sort(w); break;
    case OP_GROUP: api_group(w); break;
    case OP_PM_REWARD: api_pm_reward(w); break;
    case OP_OBSERVE: api_observe(w, o.fails); break;
    case OP_INVALIDATE: api_invalidate(w, o.shape); break;
    case OP_REPLAY: if (w.graphs & o.graph) api_graph_replay(w, o.graph); break;
    case OP_SET_AUTORESET: api_set_autoreset(w); break;
    case OP_ORDER_ATTACH: api_order_attach(w); break;
  }
  check_invariants(w);
}
static void init_world(World &w, Chooser *rng, int32_t uni_nd, bool static_ok) {
  w.rng = rng; w.uni_nd = uni_nd; w.static_ok = static_ok;
  bk_init(w.bk, static_ok, uni_nd);
  for (int i = 0; i < w.ne; ++i) { w.day[i] = 0; w.nd[i] = 1; w.fin[i] = true; w.episode[i] = -1; }
}

// ---------------------------------------------------------------- random driver
static void run_sequence(uint64_t seed, int n_ops) {
  Chooser rng;
  rng.s = seed * 0x9E3779B97F4A7C15ull + 0x1234567ull;
  World w;
  init_world(w, &rng, rng.coin(75) ? (int32_t)(2 + rng.below(7)) : -1, rng.coin(90));
  w.trace.push_back("sequence " + std::to_string(seed) + ": uni_nd " + std::to_string(w.uni_nd) + ", static_ok " +
                    std::to_string((int)w.static_ok));
  Op first; first.kind = OP_RESET;
  apply(w, first);
  if (rng.coin(70)) api_set_autoreset(w);
  for (int op = 0; op < n_ops; ++op) {
    const int u = (int)rng.below(100);
    Op o;
    const unsigned all = (1u << w.ne) - 1u;
    if (u < 40) {
      o.kind = OP_STEP;
      o.autoreset = w.has_autoreset && rng.coin(25); o.next = rng.coin(40); o.given = !o.autoreset && rng.coin(10);
      o.wide = o.given || rng.coin(70); o.unpacked = rng.coin(10); o.fails = rng.coin(3);
    } else if (u < 54) {
      o.kind = OP_RESET; o.masked = rng.coin(40);
      o.sel = rng.coin(20) ? all : (unsigned)rng.below(all + 1); o.fails = rng.coin(4);
      if (!o.masked && rng.coin(25)) o.sorted = true;
    } else if (u < 62) {
      o.kind = OP_ROLLOUT; o.n_steps = (int32_t)(1 + rng.below(9)); o.fixes = rng.coin(15);
      o.prep = rng.coin(60) ? (rng.coin(80) ? 2 : 1) : 0;
    } else if (u < 70) o.kind = OP_GET_STATE;
    else if (u < 73) o.kind = OP_SORT;
    else if (u < 78) o.kind = OP_GROUP;
    else if (u < 83) o.kind = OP_PM_REWARD;
    else if (u < 86) { o.kind = OP_OBSERVE; o.fails = rng.coin(10); }
    else if (u < 89) o.kind = OP_INVALIDATE;
    else if (u < 93) {
      o.kind = OP_STEP; o.capturing = true;
      o.wide = rng.coin(70); o.autoreset = w.has_autoreset && rng.coin(40); o.next = rng.coin(40); o.unpacked = rng.coin(10);
    } else if (u < 97) { o.kind = OP_REPLAY; o.graph = 1 << rng.below(6); }
    else if (u < 99) o.kind = OP_SET_AUTORESET;
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
  for (int64_t v : {(int64_t)b.pk_valid, (int64_t)b.canon_valid, (int64_t)b.lock, (int64_t)b.uni_t,
                    (int64_t)b.graph_canon, (int64_t)b.graph_packed, (int64_t)b.graph_autoreset, (int64_t)b.poisoned,
                    (int64_t)b.perm_valid, (int64_t)b.has_order, (int64_t)b.rm_valid, (int64_t)b.has_order_ws, (int64_t)b.hist_valid})
    put(v);  // last_step_kernel / last_rollout_kernel: outputs only
  for (int i = 0; i < w.ne; ++i) { put(w.day[i]); put(w.nd[i]); put(w.fin[i]); }
  put(w.canon_id == w.latest); put(w.packed_id == w.latest);
  put(w.pk_day_val == POISON ? 0 : (truly_uniform(w) && w.pk_day_val == w.day[0] ? 1 : 2));
  put(w.perm_for == w.epoch); put(w.rm_for == w.epoch && w.rm_order_gen == w.order_gen); put(w.hist_for == w.epoch);
  put(w.order_exists); put(w.has_autoreset); put(w.graphs);
  return k;
}
static std::vector<Op> all_ops() {
  std::vector<Op> v;
  for (int form = 0; form < 3; ++form)      // 4-lanes-per-env kernel; 64-envs-per-wave; the latter with W2A_STEP_UNPACKED
    for (int ar = 0; ar < 3; ++ar)          // none, same-step, next-step autoreset
      for (int mode = 0; mode < 3; ++mode) {  // eager, capturing, eager with a failing launch
        Op o; o.kind = OP_STEP; o.wide = form > 0; o.unpacked = form == 2; o.autoreset = ar > 0; o.next = ar == 2;
        o.capturing = mode == 1; o.fails = mode == 2;
        v.push_back(o);
      }
  { Op o; o.kind = OP_STEP; o.wide = true; o.given = true; v.push_back(o); o.fails = true; v.push_back(o); }
  for (int m = 0; m < 4; ++m) {  // unmasked; masks selecting env 0, envs 0 and 1, all three (the handle cannot see what a mask selects)
    Op o; o.kind = OP_RESET; o.masked = m > 0; o.sel = m == 1 ? 1u : (m == 2 ? 3u : 7u);
    v.push_back(o);
    o.fails = true; v.push_back(o);  // ... and the same reset with a k_reset launch that fails
    if (m == 0) { o.sorted = true; v.push_back(o); o.fails = false; v.push_back(o); }  // w2a_reset_device_rng_sorted, failing and not
  }
  { Op o; o.kind = OP_OBSERVE; o.fails = true; v.push_back(o); }
  for (int32_t n : {1, 2, 9})
    for (int fx = 0; fx < 2; ++fx)
      for (int prep = 0; prep < 3; ++prep) { Op o; o.kind = OP_ROLLOUT; o.n_steps = n; o.fixes = fx; o.prep = prep; v.push_back(o); }
  for (int kind : {(int)OP_GET_STATE, (int)OP_SORT, (int)OP_OBSERVE, (int)OP_GROUP, (int)OP_PM_REWARD, (int)OP_ORDER_ATTACH, (int)OP_SET_AUTORESET}) {
    Op o; o.kind = kind; v.push_back(o);
  }
  for (int shape = 0; shape < 2; ++shape) { Op o; o.kind = OP_INVALIDATE; o.shape = shape; v.push_back(o); }
  for (int g : {(int)G_PACKED, (int)G_CANON, (int)G_CANON_AUTO_SAME, (int)G_CANON_AUTO_NEXT, (int)G_PACKED_AUTO_SAME, (int)G_PACKED_AUTO_NEXT}) {
    Op o; o.kind = OP_REPLAY; o.graph = g; v.push_back(o);
  }
  return v;
}
static int run_bfs(int max_depth) {
  size_t total_states = 0, total_edges = 0;
  int deepest = 0;
  bool closed = true;
  const std::vector<Op> ops = all_ops();
  g_ops = &ops;
  for (int cfg = 0; cfg < 3; ++cfg) {  // the table properties a handle is created with: packable; ragged; dims outside the mirror's fields
    const int32_t uni_nd = cfg == 1 ? -1 : 3;
    const bool static_ok = cfg != 2;
    Chooser ch;
    ch.scripted = true;
    World w0;
    w0.ne = 3; w0.bfs = true;
    init_world(w0, &ch, uni_nd, static_ok);
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
    printf("  tables (uni_nd %2d, static_ok %d): %8zu reachable states, %10zu transitions checked\n", uni_nd, (int)static_ok,
           seen.size(), edges);
    fflush(stdout);
    total_states += seen.size(); total_edges += edges;
  }
  printf("bookkeeping_check --bfs: %zu reachable abstract states, %zu transitions, depth %d, %s, no violation\n", total_states,
         total_edges, deepest, closed ? "walked to closure" : "CUT at the depth limit");
  return closed || max_depth > 0 ? 0 : 2;
}

int main(int argc, char **argv) {
  if (argc > 1 && !strcmp(argv[1], "--bfs")) return run_bfs(argc > 2 ? atoi(argv[2]) : 0);
  const int n_seq = argc > 1 ? atoi(argv[1]) : 2000;
  const int n_ops = argc > 2 ? atoi(argv[2]) : 120;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
  for (int i = 0; i < n_seq; ++i) run_sequence(seed * 1000003ull + (uint64_t)i, n_ops);
  printf("bookkeeping_check: %d sequences x %d operations, no violation\n", n_seq, n_ops);
  return 0;
}
