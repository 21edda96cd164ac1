// w2a_bookkeeping.h -- host-side bookkeeping of a libw2a handle: which form of the per-env step state is current,
// what the handle knows about days and budgets, which derived structures (column grouping, visiting order, tile lists)
// still belong to the episodes the envs hold, and -- from that -- which kernel an entry point may launch.
//
// Plain C++17, no HIP: the decisions live here, the launches behind a small `Dev` interface, so that the same code
// runs (a) inside libw2a.so (w2a_kernels.hip: Dev = the HIP launches on the caller's stream) and (b) on a CPU under
// AddressSanitizer / UBSan against a recording stub that knows which form of the state is really current
// (tests/bookkeeping_check.cpp, driven by random call sequences in tests/test_bookkeeping_cpu.py). A wrong flag here
// means silently wrong rewards on the GPU, which is why this part must be testable without one.
//
// Dev must provide:   void pack_state();                              canonical words -> lock-step mirror (k_pack_state)
//                     void unpack_state(int32_t t, int32_t n_days);   mirror -> canonical words (k_unpack_state)
//
// The reference has no counterpart (its env holds one Python object per episode, env.py:162-180); what the kernels
// compute from either form of the state is env.py:238-262.
#ifndef W2A_BOOKKEEPING_H
#define W2A_BOOKKEEPING_H

#include <stdint.h>

#define W2A_BK_UNKNOWN INT64_MAX
#define W2A_BK_PACKED_MAX_BUDGET 65535  // the lock-step mirror holds budgets in 16 bits

struct W2aBook {
  // Per-env step state exists in two forms (w2a_common.hip.h, StateArrays): the canonical words and the 16-B lock-step
  // mirror. At least one of them is always current; both after a read-back of the packed form.
  int pk_valid, canon_valid;
  int pk_static_ok;      // table dims fit the mirror's bit fields (T <= 255, S < 65536, n_samples <= 1024, S_w*Y < 2^22)
  int32_t uni_t;         // the day every env is on while the batch is KNOWN to be in lock step, else -1
  int32_t uni_nd;        // the one episode length of the tables, -1 if (county, year) pairs differ in length
  int32_t pk_t;          // day to restore into the canonical words when the mirror is unpacked
  int32_t b0_max;        // largest default budget of the tables
  int64_t budget_bound;        // no env's budget exceeds this (W2A_BK_UNKNOWN: budgets came over in device memory)
  int64_t budget_bound_known;  // its last known value (w2a_set_budget_bound restores knowledge from it; W2A_BK_UNKNOWN:
                               // no bound can be known -- a sticky random walk of budgets -- until w2a_invalidate)
  int foreign;           // the state buffer was replaced by the caller (w2a_invalidate) and no bound was stated since
  int has_auto;          // w2a_set_autoreset was called: in-kernel autoresets keep drawing budgets with these arguments,
  int64_t auto_cand;     // whatever is stated about the budgets the buffer holds NOW
  int auto_centered, auto_sticky;
  int graph_captured;    // a w2a_step was recorded into a hipGraph: a replay advances days behind the host's back, so
                         // nothing that depends on the host's day count may run again -- canonical form for good, and
                         // the handle never again claims to know the day (uni_t stays -1: no packed step, no
                         // matrix-core rollout, whose tile list a replayed in-kernel autoreset would also outdate)
  int graph_autoreset;   // ... and such a recorded step carried W2A_STEP_AUTORESET: replays re-draw episodes, so a column
                         // grouping can go stale at any time -- it is never again reported valid
  int perm_valid;        // the column grouping (w2a_group_by_column) belongs to the episodes the envs hold
  int has_order;         // a visiting order exists (any permutation is correct; it may be stale = unsorted)
  int rm_valid;          // the matrix-core rollout's tile list belongs to the current episodes and order
  int last_step_kernel, last_rollout_kernel;
};

enum { W2A_BK_STEP_CLASSIC = 0, W2A_BK_STEP_WIDE = 1, W2A_BK_STEP_PACKED = 2 };
enum { W2A_BK_ROLLOUT_4LANE = 0, W2A_BK_ROLLOUT_WIDE = 1, W2A_BK_ROLLOUT_MFMA = 2 };

static inline void bk_init(W2aBook &b, bool pk_static_ok, int32_t uni_nd, int32_t b0_max) {
  b.pk_valid = 0; b.canon_valid = 1; b.pk_static_ok = pk_static_ok ? 1 : 0;
  b.uni_t = -1; b.uni_nd = uni_nd; b.pk_t = 0; b.b0_max = b0_max;
  b.budget_bound = 0; b.budget_bound_known = 0; b.graph_captured = 0; b.graph_autoreset = 0; b.foreign = 0;
  b.has_auto = 0; b.auto_cand = 0; b.auto_centered = 0; b.auto_sticky = 0;
  b.perm_valid = 0; b.has_order = 0; b.rm_valid = 0;
  b.last_step_kernel = -1; b.last_rollout_kernel = -1;
}

// something is about to READ the canonical words
template <class Dev>
static inline void bk_ensure_canonical(W2aBook &b, Dev &d) {
  if (b.canon_valid) return;
  d.unpack_state(b.pk_t, b.uni_nd);
  b.canon_valid = 1;
}
// the canonical words are about to be MODIFIED by something that does not maintain the mirror
static inline void bk_canonical_modified(W2aBook &b, bool keeps_lockstep) {
  b.pk_valid = 0;
  if (!keeps_lockstep) b.uni_t = -1;
}

// Budgets. The packed form holds budgets in 16 bits, so the handle keeps an upper bound of every budget the state buffer
// holds -- the current episodes' AND the sticky ones (cold.z) that later device-RNG resets may hand out again
// (env.py:167-170, Q9). It only ever sees reset ARGUMENTS:
//   `cand`     the largest budget a reset with these arguments can draw by itself (< 0: a caller's array in device
//              memory: the current budgets are unknown until w2a_set_budget_bound, the sticky ones are untouched);
//   centered   W2A_BUDGET_CENTERED; with `sticky` the budget is a random walk (each episode re-samples around the last
//              sampled value, also inside the kernels): no bound exists, and none can be restored by a later statement
//              about the CURRENT budgets, because the unbounded values live on as sticky budgets
//              (found by tests/bookkeeping_check.cpp: centred sticky episodes, then w2a_reset + w2a_set_budget_bound,
//              then a sticky device reset handed a budget above 65535 to the packed kernel).
static inline void bk_note_budgets(W2aBook &b, int64_t cand, bool centered, bool sticky) {
  if (centered && sticky) { b.budget_bound = b.budget_bound_known = W2A_BK_UNKNOWN; return; }
  if (b.budget_bound != W2A_BK_UNKNOWN) b.budget_bound_known = b.budget_bound;
  if (cand < 0) { b.budget_bound = W2A_BK_UNKNOWN; return; }
  if (centered) cand = cand + cand / 2 + 1;
  if (cand > b.budget_bound) b.budget_bound = cand;
}
// The caller states that no budget handed over in device memory exceeds `bound`. The first statement after
// bk_invalidate comes from the library itself (w2a_invalidate scans the restored buffer for its largest budget, sticky
// ones included) and is taken as covering everything; any other is combined with what the handle knew before the
// budgets went out of sight -- which may be "nothing can be known" (W2A_BK_UNKNOWN: sticky random walk), and then stays so.
static inline void bk_set_budget_bound(W2aBook &b, int64_t bound) {
  if (bound < 0) { b.budget_bound = W2A_BK_UNKNOWN; return; }
  if (b.foreign) {
    b.budget_bound = b.budget_bound_known = bound;
    b.foreign = 0;
  } else {
    const int64_t prev = b.budget_bound != W2A_BK_UNKNOWN ? b.budget_bound : b.budget_bound_known;
    if (prev == W2A_BK_UNKNOWN) return;
    b.budget_bound = bound > prev ? bound : prev;  // budgets of earlier episodes may live on as sticky budgets
  }
  // the statement is about the budgets in the buffer; the autoreset parameters of the handle go on handing out theirs
  if (b.has_auto) bk_note_budgets(b, b.auto_cand, b.auto_centered != 0, b.auto_sticky != 0);
}
// w2a_set_autoreset: the parameters in-kernel autoresets draw budgets with from now on
static inline void bk_set_autoreset(W2aBook &b, int64_t cand, bool centered, bool sticky) {
  b.has_auto = 1; b.auto_cand = cand; b.auto_centered = centered ? 1 : 0; b.auto_sticky = sticky ? 1 : 0;
  bk_note_budgets(b, cand, centered, sticky);
}
static inline bool bk_packed_eligible(const W2aBook &b) {
  return b.pk_static_ok && b.budget_bound <= W2A_BK_PACKED_MAX_BUDGET && b.uni_nd > 0;
}

// k_reset. observe_only: w2a_observe (first observations re-emitted, state untouched). A full reset rewrites every
// env's canonical words from `cold`, which is never stale; a masked reset and w2a_observe read the rest as well.
template <class Dev>
static inline void bk_reset(W2aBook &b, Dev &d, bool observe_only, bool masked) {
  if (!observe_only) b.perm_valid = 0;  // new episode tuples: the column grouping is stale
  if (masked || observe_only) bk_ensure_canonical(b, d);
  if (observe_only) return;
  b.rm_valid = 0;  // new episodes: the feature-row tile list of the matrix-core rollout is stale
  b.canon_valid = 1;
  bk_canonical_modified(b, false);
  // every env on day 0 of an episode of the one length there is -- knowledge that a replay of a recorded graph would
  // outdate without the host noticing, so a handle that was ever captured does not keep it
  if (!masked && b.uni_nd > 0 && !b.graph_captured) b.uni_t = 0;
}

struct BkStepPlan {
  int kernel;            // W2A_BK_STEP_*; < 0: refused (capture started on the packed form)
  int32_t uni_t, uni_nd; // kernel arguments of the packed variant
};
// w2a_step. wide_wanted: the 64-envs-per-wave kernel serves this call (batch size / W2A_STEP_WIDE / REWARD_GIVEN, and
// W2A_STEP_CLASSIC not set). capturing: the stream is recording a hipGraph.
template <class Dev>
static inline BkStepPlan bk_step(W2aBook &b, Dev &d, bool wide_wanted, bool autoreset, bool given, bool unpacked_flag,
                                 bool capturing) {
  BkStepPlan p;
  p.kernel = W2A_BK_STEP_CLASSIC; p.uni_t = -1; p.uni_nd = b.uni_nd;
  if (autoreset) {  // envs that finish draw new episodes inside the kernel: the column grouping and the feature-row
    b.perm_valid = 0;  // tile list go stale (the latter was only ever used in lock step, which such a step ends too;
    b.rm_valid = 0;    // dropped here as well so that "valid" always means what it says)
  }
  // the day every env is on after this call, if the batch is (still) known to be in lock step: a plain step moves all
  // of them to the next day; the terminal step, an in-kernel autoreset or unknown state ends the knowledge
  int32_t uni_next = (!autoreset && b.uni_t >= 0 && b.uni_t + 1 < b.uni_nd) ? b.uni_t + 1 : -1;
  if (capturing) {
    if (!b.canon_valid) { p.kernel = -1; return p; }
    b.graph_captured = 1;
    if (autoreset) b.graph_autoreset = 1;
  }
  if (b.graph_captured) uni_next = -1;
  if (wide_wanted) {
    const bool packed = !given && !autoreset && !unpacked_flag && b.pk_static_ok && b.uni_t >= 0 &&
                        b.budget_bound <= W2A_BK_PACKED_MAX_BUDGET && !b.graph_captured;
    if (packed) {
      if (!b.pk_valid) {  // entering the packed form (once per episode): the canonical words are current
        d.pack_state();
        b.pk_valid = 1;
      }
      p.kernel = W2A_BK_STEP_PACKED; p.uni_t = b.uni_t;
      b.canon_valid = 0;
      b.pk_t = uni_next >= 0 ? uni_next : b.uni_t;  // the terminal step leaves t where it is (env.py:256-259)
      b.uni_t = uni_next;
      b.last_step_kernel = W2A_BK_STEP_PACKED;
      return p;
    }
    p.kernel = W2A_BK_STEP_WIDE;
  }
  bk_ensure_canonical(b, d);
  bk_canonical_modified(b, true);
  b.uni_t = uni_next;
  b.last_step_kernel = p.kernel;
  return p;
}

// w2a_rollout / w2a_rollout_posterior_mean: n_steps days, or to the end of every env's episode. A batch in lock step
// stays in lock step: every env runs the same days, or all of them reach their last day. Returns the lock-step day
// the call STARTED from (-1 unknown).
template <class Dev>
static inline int32_t bk_rollout_begin(W2aBook &b, Dev &d, int32_t n_steps) {
  const int32_t day = b.uni_t;
  bk_ensure_canonical(b, d);
  bk_canonical_modified(b, true);
  b.uni_t = (day >= 0 && day + n_steps < b.uni_nd && !b.graph_captured) ? day + n_steps : -1;
  return day;
}
// which sampled-reward rollout kernel serves the call (fixes: any W2A_FIX_* bit set)
static inline int bk_rollout_kernel(W2aBook &b, int32_t start_day, bool fixes, bool mfma_built, bool wide_built) {
  int k = (wide_built && b.has_order) ? W2A_BK_ROLLOUT_WIDE : W2A_BK_ROLLOUT_4LANE;
  if (b.rm_valid && b.has_order && !fixes && start_day >= 0 && mfma_built) k = W2A_BK_ROLLOUT_MFMA;
  b.last_rollout_kernel = k;
  return k;
}

// w2a_sort_episodes: a relabelling -- reads and rewrites the canonical words, the batch stays in lock step
template <class Dev>
static inline void bk_sort(W2aBook &b, Dev &d) {
  bk_ensure_canonical(b, d);
  bk_canonical_modified(b, true);
  b.rm_valid = 0;
  b.perm_valid = 0;  // every env index now holds another episode: the column grouping is stale
}
static inline void bk_grouped(W2aBook &b) { b.perm_valid = b.graph_autoreset ? 0 : 1; }
static inline void bk_order_set(W2aBook &b) { b.has_order = 1; b.rm_valid = 0; }
static inline void bk_rm_prepared(W2aBook &b) { b.rm_valid = b.graph_autoreset ? 0 : 1; }

// w2a_invalidate: the caller has overwritten the state buffer (its canonical part): forget every derived form
static inline void bk_invalidate(W2aBook &b) {
  b.budget_bound = b.budget_bound_known = W2A_BK_UNKNOWN;  // what was known described another buffer
  b.foreign = 1;
  b.pk_valid = 0; b.canon_valid = 1; b.uni_t = -1; b.perm_valid = 0;
  b.rm_valid = 0;  // feature rows may have changed behind the handle: the matrix-core rollout's tile list is stale
}

#endif  // W2A_BOOKKEEPING_H
