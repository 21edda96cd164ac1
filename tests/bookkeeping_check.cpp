// bookkeeping_check.cpp -- the handle's host-side bookkeeping (weather2alert_amd/csrc/w2a_bookkeeping.h, the very header
// libw2a.so compiles) driven by random call sequences on the CPU against a recording stub that KNOWS what is really the
// case: a handful of simulated envs (day, finished, episode length, budget, sticky budget, episode id) and, for each of
// the two forms of the per-env step state, which version of the contents it holds. Built by tests/test_bookkeeping_cpu.py
// with g++ -fsanitize=address,undefined; the same test also builds mutants of the header (one flag rule broken each) and
// requires this program to catch every one of them.
//
// The entry points below restate, call for call, what csrc/w2a_kernels.hip / w2a_step_dispatch.hip.h do around their
// kernel launches (each names the function it follows); a launch becomes "reads form X" / "writes form X".
//
// Violations reported:
//   stale read        a kernel reads a form of the state that does not hold the latest contents
//   false lock step   the handle claims a lock-step day the envs are not on / hands a wrong day or length to a kernel
//   packed budgets    the 16-bit packed form is used while some env's budget exceeds 65535
//   stale grouping    the posterior-mean reward / matrix-core rollout runs on a grouping or tile list of other episodes
//   no valid form     neither form is marked current
//
// usage: bookkeeping_check <sequences> <ops per sequence> <seed>     exit 0 = no violation
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "w2a_bookkeeping.h"

static const int NE = 6;  // simulated envs

struct Rng {
  uint64_t s;
  uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  int64_t below(int64_t n) { return (int64_t)(next() % (uint64_t)n); }
  bool coin(int pct) { return below(100) < pct; }
};

struct Cfg { int64_t budget_kw; int mode; int sticky; };  // autoreset / reset parameters (w2a_set_autoreset)

struct World {
  // ---- truth
  int32_t day[NE], nd[NE];
  bool fin[NE];
  int64_t budget[NE], sticky[NE];
  long episode[NE];
  long clock = 0, latest = 0, canon_id = 0, packed_id = -1;
  long epoch = 0;        // changes whenever any env index gets another episode
  long perm_for = -1, order_exists = 0, rm_for = -1, rm_order_gen = -1, order_gen = 0;
  int32_t uni_nd;        // table property
  int32_t b0_max;
  bool static_ok;
  bool has_autoreset = false;
  Cfg acfg{-1, 0, 1};
  bool graph_recorded = false;
  bool graph_auto = false, graph_next = false;  // the recorded step carried W2A_STEP_AUTORESET (| W2A_STEP_NEXT_STEP)
  // ---- the handle
  W2aBook bk;
  std::vector<std::string> trace;
  Rng *rng;
};

static void fail(World &w, const char *what) {
  printf("VIOLATION: %s\n", what);
  size_t from = w.trace.size() > 40 ? w.trace.size() - 40 : 0;
  for (size_t i = from; i < w.trace.size(); ++i) printf("  %s\n", w.trace[i].c_str());
  exit(1);
}
#define REQUIRE(w, cond, what) do { if (!(cond)) fail(w, what); } while (0)

struct StubDev {
  World &w;
  void pack_state() {
    REQUIRE(w, w.canon_id == w.latest, "stale read: k_pack_state reads canonical words that are not current");
    w.packed_id = w.latest;
    w.trace.push_back("    [k_pack_state]");
  }
  void unpack_state(int32_t t, int32_t n_days) {
    REQUIRE(w, w.packed_id == w.latest, "stale read: k_unpack_state reads a mirror that is not current");
    for (int i = 0; i < NE; ++i) {
      REQUIRE(w, w.day[i] == t, "false lock step: k_unpack_state restores a day the envs are not on");
      REQUIRE(w, w.nd[i] == n_days, "false lock step: k_unpack_state restores a wrong episode length");
    }
    w.canon_id = w.latest;
    w.trace.push_back("    [k_unpack_state t=" + std::to_string(t) + "]");
  }
};

static bool truly_lockstep(const World &w, int32_t *day) {
  for (int i = 0; i < NE; ++i)
    if (w.fin[i] || w.day[i] != w.day[0] || w.nd[i] != w.nd[0]) return false;
  *day = w.day[0];
  return true;
}
static void read_canon(World &w, const char *who) {
  if (w.canon_id != w.latest) fail(w, (std::string("stale read: ") + who + " reads canonical words that are not current").c_str());
}
static void write_canon(World &w) { w.canon_id = w.latest = ++w.clock; }
static int32_t table_len(World &w) { return w.uni_nd > 0 ? w.uni_nd : (int32_t)(3 + w.rng->below(6)); }
static int64_t table_b0(World &w) { return w.rng->below((int64_t)w.b0_max + 1); }

// draw_episode (csrc/w2a_common.hip.h) for env i: budget rules of env.py:167-178
static void new_episode(World &w, int i, const Cfg &c) {
  int64_t b = (c.sticky && w.sticky[i] >= 0) ? w.sticky[i] : (c.budget_kw < 0 ? table_b0(w) : c.budget_kw);
  if (b < 0) b = 0;
  if (c.mode == 1) b = w.rng->below(b + 1);
  else if (c.mode == 2) { int64_t lo = b / 2, hi = (int64_t)(1.5 * (double)b + 1.0); b = lo + w.rng->below(hi - lo); }
  w.budget[i] = b;
  w.sticky[i] = c.sticky ? b : -1;
  w.day[i] = 0; w.fin[i] = false; w.nd[i] = table_len(w); w.episode[i]++;
}

static void check_invariants(World &w) {
  REQUIRE(w, w.bk.pk_valid || w.bk.canon_valid, "no valid form: neither the canonical words nor the mirror is marked current");
  if (w.bk.canon_valid) REQUIRE(w, w.canon_id == w.latest, "stale flag: canonical words marked current, but they are not");
  if (w.bk.pk_valid) REQUIRE(w, w.packed_id == w.latest, "stale flag: mirror marked current, but it is not");
  if (w.bk.uni_t >= 0) {
    int32_t d = -1;
    REQUIRE(w, truly_lockstep(w, &d) && d == w.bk.uni_t, "false lock step: W2A_Q_LOCKSTEP_DAY is not the day the envs are on");
  }
  if (w.bk.perm_valid) REQUIRE(w, w.perm_for == w.epoch, "stale grouping: perm_valid although env indices hold other episodes");
  if (w.bk.rm_valid) REQUIRE(w, w.rm_for == w.epoch && w.rm_order_gen == w.order_gen,
                             "stale grouping: rm_valid although the tile list belongs to other episodes / another order");
}

// ---------------------------------------------------------------- entry points (bookkeeping side of w2a_kernels.hip)
static void api_reset_device(World &w, const Cfg &c, bool masked, bool mask_all) {  // w2a_reset_device_rng + launch_reset
  bk_note_budgets(w.bk, c.budget_kw >= 0 ? c.budget_kw : w.bk.b0_max, c.mode == 2, c.sticky != 0);
  StubDev d{w};
  bk_reset(w.bk, d, false, masked);
  if (masked) read_canon(w, "k_reset (masked)");
  for (int i = 0; i < NE; ++i)
    if (!masked || mask_all || w.rng->coin(50)) new_episode(w, i, c);
  w.epoch = ++w.clock;
  write_canon(w);
}
static void api_reset_tuples(World &w, bool with_budgets, int64_t bmax, bool masked, bool tell_bound) {  // w2a_reset
  bk_note_budgets(w.bk, with_budgets ? -1 : w.bk.b0_max, false, false);
  StubDev d{w};
  bk_reset(w.bk, d, false, masked);
  if (masked) read_canon(w, "k_reset (masked)");
  int64_t seen = 0;
  for (int i = 0; i < NE; ++i)
    if (!masked || w.rng->coin(50)) {
      w.budget[i] = with_budgets ? w.rng->below(bmax + 1) : table_b0(w);
      if (w.budget[i] > seen) seen = w.budget[i];
      w.day[i] = 0; w.fin[i] = false; w.nd[i] = table_len(w); w.episode[i]++;  // the sticky budget stays (cold.z)
    }
  w.epoch = ++w.clock;
  write_canon(w);
  if (tell_bound) {  // HeatAlertVecEnv._reset_tuples: w2a_set_budget_bound(max of the budgets handed over)
    bk_set_budget_bound(w.bk, seen);
    w.trace.push_back("    w2a_set_budget_bound(" + std::to_string(seen) + ")");
  }
}
static void api_observe(World &w) {  // w2a_observe
  StubDev d{w};
  bk_reset(w.bk, d, true, false);
  read_canon(w, "k_reset (observe)");
}
static void api_set_autoreset(World &w, const Cfg &c) {  // w2a_set_autoreset
  w.acfg = c; w.has_autoreset = true;
  bk_set_autoreset(w.bk, c.budget_kw >= 0 ? c.budget_kw : w.bk.b0_max, c.mode == 2, c.sticky != 0);
}
static void advance(World &w, int i) {  // one day of env.py:256-260
  if (w.day[i] + 1 >= w.nd[i]) w.fin[i] = true; else w.day[i]++;
}
static void api_step(World &w, bool wide, bool autoreset, bool next_step, bool given, bool unpacked, bool capturing) {  // w2a_step
  StubDev d{w};
  const BkStepPlan p = bk_step(w.bk, d, wide, autoreset, given, unpacked, capturing);
  if (p.kernel < 0) { w.trace.push_back("    (refused: capture on the packed form)"); return; }
  if (capturing) {  // recorded, not executed
    w.graph_recorded = true; w.graph_auto = autoreset; w.graph_next = next_step;
    return;
  }
  if (p.kernel == W2A_BK_STEP_PACKED) {
    REQUIRE(w, w.packed_id == w.latest, "stale read: the packed step kernel reads a mirror that is not current");
    int32_t day = -1;
    REQUIRE(w, truly_lockstep(w, &day) && day == p.uni_t && w.nd[0] == p.uni_nd,
            "false lock step: the packed step kernel was handed a day / length the envs are not on");
    for (int i = 0; i < NE; ++i) REQUIRE(w, w.budget[i] <= 65535, "packed budgets: budget above 65535 in the 16-bit mirror");
    REQUIRE(w, w.static_ok && !w.graph_recorded, "packed form used although the tables / a recorded graph forbid it");
    REQUIRE(w, !given && !autoreset, "the packed kernel has no REWARD_GIVEN / AUTORESET variant");
    for (int i = 0; i < NE; ++i) advance(w, i);
    w.packed_id = w.latest = ++w.clock;
    return;
  }
  read_canon(w, "the step kernel");
  bool changed = false;
  for (int i = 0; i < NE; ++i) {
    if (autoreset && next_step && w.fin[i]) { new_episode(w, i, w.acfg); changed = true; continue; }
    advance(w, i);  // a finished env repeats its last day (env.py:256: done again)
    if (autoreset && !next_step && w.fin[i]) { new_episode(w, i, w.acfg); changed = true; }
  }
  if (changed) w.epoch = ++w.clock;
  write_canon(w);
}
static void api_graph_replay(World &w) {  // hipGraphLaunch of recorded canonical step kernels: no host bookkeeping runs
  read_canon(w, "a replayed (captured) step kernel");
  bool changed = false;
  for (int i = 0; i < NE; ++i) {
    if (w.graph_auto && w.graph_next && w.fin[i]) { new_episode(w, i, w.acfg); changed = true; continue; }
    advance(w, i);
    if (w.graph_auto && !w.graph_next && w.fin[i]) { new_episode(w, i, w.acfg); changed = true; }
  }
  if (changed) w.epoch = ++w.clock;
  write_canon(w);
}
static void api_rollout(World &w, int32_t n_steps, bool fixes) {  // w2a_rollout
  StubDev d{w};
  const int32_t start = bk_rollout_begin(w.bk, d, n_steps);
  const int k = bk_rollout_kernel(w.bk, start, fixes, true, true);
  read_canon(w, "the rollout kernel");
  if (k == W2A_BK_ROLLOUT_MFMA) {
    int32_t day = -1;
    REQUIRE(w, truly_lockstep(w, &day) && day == start, "false lock step: the matrix-core rollout needs the batch in lock step");
    REQUIRE(w, w.rm_for == w.epoch && w.rm_order_gen == w.order_gen && !fixes,
            "stale grouping: the matrix-core rollout runs on a tile list of other episodes / another order");
  }
  for (int i = 0; i < NE; ++i)
    for (int s = 0; s < n_steps && !w.fin[i]; ++s) advance(w, i);
  write_canon(w);
}
static void api_get_state(World &w) {  // w2a_get_state
  StubDev d{w};
  bk_ensure_canonical(w.bk, d);
  read_canon(w, "k_get_state");
}
static void api_sort(World &w) {  // w2a_sort_episodes
  StubDev d{w};
  bk_sort(w.bk, d);
  read_canon(w, "k_permute_state");
  for (int i = 0; i + 1 < NE; i += 2) {  // a relabelling: whole records swap places
    std::swap(w.day[i], w.day[i + 1]); std::swap(w.nd[i], w.nd[i + 1]); std::swap(w.fin[i], w.fin[i + 1]);
    std::swap(w.budget[i], w.budget[i + 1]); std::swap(w.sticky[i], w.sticky[i + 1]); std::swap(w.episode[i], w.episode[i + 1]);
  }
  w.epoch = ++w.clock;
  write_canon(w);
}
static void api_group(World &w) {  // w2a_group_by_column (reads stepc / cold: never stale)
  w.perm_for = w.epoch;
  bk_grouped(w.bk);
}
static void api_pm_reward(World &w) {  // w2a_posterior_mean_reward
  if (!w.bk.perm_valid) { w.trace.push_back("    (refused: grouping stale)"); return; }
  REQUIRE(w, w.perm_for == w.epoch, "stale grouping: the posterior-mean reward runs on a grouping of other episodes");
  StubDev d{w};
  bk_ensure_canonical(w.bk, d);
  read_canon(w, "k_pm_prep");
}
static void api_rollout_order(World &w) {  // w2a_rollout_order
  w.order_exists = 1; w.order_gen = ++w.clock;
  bk_order_set(w.bk);
}
static void api_rm_prepare(World &w) {  // w2a_rollout_mfma_prepare (needs an order)
  if (!w.bk.has_order) return;
  w.rm_for = w.epoch; w.rm_order_gen = w.order_gen;
  bk_rm_prepared(w.bk);
}
static void api_invalidate(World &w, bool tell) {  // the caller restored a checkpoint of the canonical part, then w2a_invalidate
  for (int i = 0; i < NE; ++i) {
    w.day[i] = (int32_t)w.rng->below(3); w.nd[i] = table_len(w); w.fin[i] = w.rng->coin(10);
    w.budget[i] = w.rng->below(w.rng->coin(20) ? 100000 : 12); w.sticky[i] = w.rng->coin(50) ? w.budget[i] : -1; w.episode[i]++;
  }
  w.epoch = ++w.clock;
  write_canon(w);
  bk_invalidate(w.bk);
  {  // w2a_invalidate scans the restored buffer itself (k_budget_scan): largest budget, current and sticky
    int64_t m = 0;
    for (int i = 0; i < NE; ++i) { if (w.budget[i] > m) m = w.budget[i]; if (w.sticky[i] > m) m = w.sticky[i]; }
    bk_set_budget_bound(w.bk, m);
    if (tell) bk_set_budget_bound(w.bk, w.rng->below(m + 1));  // a caller's (possibly smaller) statement changes nothing
  }
}

static Cfg random_cfg(World &w) {
  Cfg c;
  const int u = (int)w.rng->below(10);
  c.budget_kw = u < 5 ? -1 : (u < 8 ? w.rng->below(9) : 60000 + w.rng->below(20000));
  c.mode = (int)w.rng->below(3);
  c.sticky = w.rng->coin(70) ? 1 : 0;
  return c;
}

static void run_sequence(uint64_t seed, int n_ops) {
  Rng rng{seed * 0x9E3779B97F4A7C15ull + 0x1234567ull};
  World w;
  w.rng = &rng;
  w.uni_nd = rng.coin(75) ? (int32_t)(2 + rng.below(7)) : -1;
  w.b0_max = rng.coin(80) ? (int32_t)(1 + rng.below(9)) : 70000;
  w.static_ok = rng.coin(90);
  bk_init(w.bk, w.static_ok, w.uni_nd, w.b0_max);
  for (int i = 0; i < NE; ++i) { w.day[i] = 0; w.nd[i] = 1; w.fin[i] = true; w.budget[i] = 0; w.sticky[i] = -1; w.episode[i] = -1; }
  w.trace.push_back("sequence " + std::to_string(seed) + ": uni_nd " + std::to_string(w.uni_nd) + ", b0_max " +
                    std::to_string(w.b0_max) + ", static_ok " + std::to_string((int)w.static_ok));
  Cfg c0 = random_cfg(w);
  api_reset_device(w, c0, false, false);
  api_set_autoreset(w, c0);
  check_invariants(w);
  for (int op = 0; op < n_ops; ++op) {
    const int u = (int)rng.below(100);
    char buf[200];
    if (u < 42) {
      const bool autoreset = w.has_autoreset && rng.coin(25), next = rng.coin(40), given = !autoreset && rng.coin(10);
      const bool wide = given || rng.coin(70), unpacked = rng.coin(10);
      snprintf(buf, sizeof buf, "step(wide %d autoreset %d next %d given %d unpacked %d)", wide, autoreset, next, given, unpacked);
      w.trace.push_back(buf);
      if (given) api_pm_reward(w);
      api_step(w, wide, autoreset, next, given, unpacked, false);
    } else if (u < 50) {
      const Cfg c = random_cfg(w);
      const bool masked = rng.coin(40), all = rng.coin(20);
      snprintf(buf, sizeof buf, "reset_device(kw %lld mode %d sticky %d masked %d)", (long long)c.budget_kw, c.mode, c.sticky, masked);
      w.trace.push_back(buf);
      api_reset_device(w, c, masked, all);
      api_set_autoreset(w, c);
    } else if (u < 56) {
      const bool wb = rng.coin(70), masked = rng.coin(40), tell = rng.coin(80);
      const int64_t bmax = rng.coin(80) ? 9 : 90000;
      snprintf(buf, sizeof buf, "reset_tuples(budgets %d max %lld masked %d tell %d)", wb, (long long)bmax, masked, tell);
      w.trace.push_back(buf);
      api_reset_tuples(w, wb, bmax, masked, tell);
    } else if (u < 64) {
      const int32_t n = (int32_t)(1 + rng.below(9));
      const bool fixes = rng.coin(15);
      snprintf(buf, sizeof buf, "rollout(%d, fixes %d)", n, fixes);
      w.trace.push_back(buf);
      if (rng.coin(60)) { api_rollout_order(w); if (rng.coin(80)) api_rm_prepare(w); }
      api_rollout(w, n, fixes);
    } else if (u < 72) { w.trace.push_back("get_state"); api_get_state(w);
    } else if (u < 75) { w.trace.push_back("sort"); api_sort(w);
    } else if (u < 80) { w.trace.push_back("group_by_column"); api_group(w);
    } else if (u < 85) { w.trace.push_back("posterior_mean_reward"); api_pm_reward(w);
    } else if (u < 88) { w.trace.push_back("observe"); api_observe(w);
    } else if (u < 91) {
      const bool tell = rng.coin(70);
      w.trace.push_back(tell ? "checkpoint restore; invalidate; set_budget_bound" : "checkpoint restore; invalidate");
      api_invalidate(w, tell);
    } else if (u < 94) {
      w.trace.push_back("capture one step into a hipGraph");
      api_step(w, rng.coin(70), w.has_autoreset && rng.coin(50), rng.coin(40), false, rng.coin(10), true);
    } else if (u < 97) {
      if (w.graph_recorded) { w.trace.push_back("graph replay"); api_graph_replay(w); }
    } else {
      const Cfg c = random_cfg(w);
      w.trace.push_back("set_autoreset");
      api_set_autoreset(w, c);
    }
    check_invariants(w);
  }
}

int main(int argc, char **argv) {
  const int n_seq = argc > 1 ? atoi(argv[1]) : 2000;
  const int n_ops = argc > 2 ? atoi(argv[2]) : 120;
  const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
  for (int i = 0; i < n_seq; ++i) run_sequence(seed * 1000003ull + (uint64_t)i, n_ops);
  printf("bookkeeping_check: %d sequences x %d operations, no violation\n", n_seq, n_ops);
  return 0;
}
