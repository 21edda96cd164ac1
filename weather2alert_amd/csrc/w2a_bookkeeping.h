// w2a_bookkeeping.h -- host-side bookkeeping of a libw2a handle: which form of the per-env step state is current,
// what the handle knows about days, which derived structures (column grouping, visiting order, tile lists)
// still belong to the episodes the envs hold, and -- from that -- which kernel an entry point may launch.
// (Nothing about budgets: until round 5 the handle kept an upper bound of every budget in the state buffer, because the
// packed form held them in 16 bits -- eleven fields and three functions of this header, four of its eight known holes.
// The packed kernel now serves any budget by itself: w2a_common.hip.h, pk_budget16.)
//
// Plain C++17, no HIP: the decisions live here, the launches behind a small `Dev` interface, so that the same code
// runs (a) inside libw2a.so (w2a_kernels.hip: Dev = the HIP launches on the caller's stream) and (b) on a CPU under
// AddressSanitizer / UBSan against a recording stub that knows which form of the state is really current
// (tests/bookkeeping_check.cpp: random call sequences AND an exhaustive walk of the abstract state space, driven by
// tests/test_bookkeeping_cpu.py). A wrong flag here means silently wrong rewards on the GPU, which is why this part
// must be testable without one.
//
// Dev must provide:   void pack_state();                  canonical words -> lock-step mirror (k_pack_state; the day of
//                                                         every 64-env tile goes into the mirror's day word)
//                     void unpack_state(int32_t n_days);  mirror -> canonical words (k_unpack_state)
//                     void poison_mirror();               the mirror's day words := W2A_PK_DAY_POISON (see graph_packed)
//
// The reference has no counterpart (its env holds one Python object per episode, env.py:162-180); what the kernels
// compute from either form of the state is env.py:238-262.
#ifndef W2A_BOOKKEEPING_H
#define W2A_BOOKKEEPING_H

#include <stdint.h>

struct W2aBook {
  // Per-env step state exists in two forms (w2a_common.hip.h, StateArrays): the canonical words and the 16-B lock-step
  // mirror. At least one of them is always current; both after a read-back of the packed form.
  int pk_valid, canon_valid;
  int pk_static_ok;      // table dims fit the mirror's bit fields (T <= 255, S < 65536, n_samples <= 1024, S_w*Y < 2^22)
  int lock;              // the batch is KNOWN to be in lock step: every env on the same day of an episode of the one
                         // length there is, so they also finish -- and, with an in-kernel autoreset, restart -- together.
                         // What the mirror (its day lives on the device, one word per 64-env tile) and the matrix-core
                         // rollout need. Steps of every kind (the terminal one too) and rollouts keep it; masked resets
                         // and w2a_invalidate end it
  int32_t uni_t;         // that day while the HOST knows it, else -1 (lock may still hold: after the terminal step, and
                         // on any handle with a recorded graph, whose replays advance days behind the host's back)
  int32_t uni_nd;        // the one episode length of the tables, -1 if (county, year) pairs differ in length
  // hipGraphs. A recorded step kernel is tied to the FORM of the state it steps, and a replay runs it without any of
  // this bookkeeping. So the form a recorded kernel reads must be current whenever a replay can happen = at every
  // boundary between two API calls:
  int graph_canon;       // a step kernel on the CANONICAL words was recorded: they stay current for good (no packed step
                         // ever again on this handle)
  int graph_packed;      // a PACKED step was recorded (the mirror was already current, so no conversion is in the graph;
                         // the kernel reads the day from the mirror, so replays advance it themselves): from now on the
                         // mirror is the primary form -- at the end of every API call either it is current and the
                         // canonical words count as a scratch copy (canon_valid = 0: a replay may outdate them), or, where
                         // the batch can no longer be packed (lock step lost: a masked reset, a restored checkpoint),
                         // its day words are POISONED so that a replay raises W2A_ST_STALE_GRAPH instead of stepping
                         // stale state
  int graph_autoreset;   // a recorded step carried W2A_STEP_AUTORESET: replays re-draw episodes at any time -- a column
                         // grouping, a tile list, row counts are never again reported valid
  int poisoned;          // the mirror's day words hold the poison value
  int perm_valid;        // the column grouping (w2a_group_by_column) belongs to the episodes the envs hold
  int has_order;         // a visiting order exists (any permutation is correct; it may be stale = unsorted)
  int rm_valid;          // the matrix-core rollout's tile list belongs to the current episodes and order
  int has_order_ws;      // w2a_rollout_order_attach: whole-batch resets also count the envs of every feature row and
  int hist_valid;        // give each env its rank inside its row (k_reset) -- and these still belong to the episodes held
  int last_step_kernel, last_rollout_kernel;
};

enum { W2A_BK_STEP_CLASSIC = 0, W2A_BK_STEP_WIDE = 1, W2A_BK_STEP_PACKED = 2 };
enum { W2A_BK_ROLLOUT_4LANE = 0, W2A_BK_ROLLOUT_WIDE = 1, W2A_BK_ROLLOUT_MFMA = 2 };

static inline void bk_init(W2aBook &b, bool pk_static_ok, int32_t uni_nd) {
  b.pk_valid = 0; b.canon_valid = 1; b.pk_static_ok = pk_static_ok ? 1 : 0;
  b.lock = 0; b.uni_t = -1; b.uni_nd = uni_nd;
  b.graph_canon = 0; b.graph_packed = 0; b.graph_autoreset = 0; b.poisoned = 0;
  b.perm_valid = 0; b.has_order = 0; b.rm_valid = 0; b.has_order_ws = 0; b.hist_valid = 0;
  b.last_step_kernel = -1; b.last_rollout_kernel = -1;
}
static inline bool bk_any_graph(const W2aBook &b) { return b.graph_canon || b.graph_packed; }

// something is about to READ the canonical words
template <class Dev>
static inline void bk_ensure_canonical(W2aBook &b, Dev &d) {
  if (b.canon_valid) return;
  d.unpack_state(b.uni_nd);
  b.canon_valid = 1;
}
// the canonical words are about to be MODIFIED by something that does not maintain the mirror
static inline void bk_canonical_modified(W2aBook &b, bool keeps_lockstep) {
  b.pk_valid = 0;
  if (!keeps_lockstep) { b.lock = 0; b.uni_t = -1; }
}

// the tables allow the lock-step mirror at all: one episode length, dims inside its bit fields (any budget is served)
static inline bool bk_packed_eligible(const W2aBook &b) {
  return b.pk_static_ok && b.uni_nd > 0;
}
// the batch may be stepped in the packed form right now
static inline bool bk_can_pack(const W2aBook &b) { return bk_packed_eligible(b) && b.lock && !b.graph_canon; }

// End of every entry point that may have changed which form is current (all that take a Dev). Keeps the invariant of
// graph_packed handles: the mirror current and the canonical words a scratch copy, or the mirror poisoned.
template <class Dev>
static inline void bk_end_call(W2aBook &b, Dev &d) {
  if (!b.graph_packed) return;
  if (b.pk_valid) { b.canon_valid = 0; return; }  // a replay may step the mirror at any time from here on
  if (bk_can_pack(b)) {  // the canonical words were modified (they are current: one form always is)
    d.pack_state();
    b.pk_valid = 1; b.canon_valid = 0; b.poisoned = 0;
  } else if (!b.poisoned) {
    d.poison_mirror();
    b.poisoned = 1;
  }
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
  // every env on day 0 of an episode of the one length there is. Replays of a recorded step move the day behind the
  // host's back: handles with one do not claim to know it
  if (!masked && b.uni_nd > 0) b.lock = 1;
  if (!masked && b.uni_nd > 0 && !bk_any_graph(b)) b.uni_t = 0;
  // a whole-batch reset with an attached order workspace also leaves the feature-row counts and per-env ranks there
  b.hist_valid = (!masked && b.has_order_ws && !b.graph_autoreset) ? 1 : 0;
}

// w2a_reset_device_rng_sorted: a whole-batch reset whose new episodes land on the env indices in coefficient-row order. To the
// flags it is a whole-batch reset -- except that its second pass does not do the rank atomics of an attached order workspace
template <class Dev>
static inline void bk_reset_sorted(W2aBook &b, Dev &d) {
  bk_reset(b, d, false, false);
  b.hist_valid = 0;
}

// The k_reset launch bk_reset prepared did not happen: the state is what it was, except that the conversion a masked
// reset / w2a_observe on the packed form asked for did run and left both forms current. Derived structures are dropped.
static inline void bk_reset_rollback(W2aBook &b, const W2aBook &before, bool observe_only, bool masked) {
  const bool unpacked = (masked || observe_only) && !before.canon_valid;
  b = before;
  if (unpacked) b.canon_valid = 1;
  b.hist_valid = 0; b.perm_valid = 0; b.rm_valid = 0;
}

struct BkStepPlan {
  int kernel;            // W2A_BK_STEP_*; < 0: refused (the capture would have to record a conversion of the state's form)
  int32_t uni_nd;        // kernel argument of the packed variant (a table constant; the day is read on the device)
  int converted;         // 1: pack_state ran, 2: unpack_state ran (what bk_step_rollback keeps)
};
// w2a_step. wide_wanted: the 64-envs-per-wave kernel serves this call (batch size / W2A_STEP_WIDE / REWARD_GIVEN, and
// W2A_STEP_CLASSIC not set). capturing: the stream is recording a hipGraph.
template <class Dev>
static inline BkStepPlan bk_step(W2aBook &b, Dev &d, bool wide_wanted, bool autoreset, bool given, bool unpacked_flag,
                                 bool capturing) {
  BkStepPlan p;
  p.kernel = W2A_BK_STEP_CLASSIC; p.uni_nd = b.uni_nd; p.converted = 0;
  // (an in-kernel autoreset does not stand in the way: a batch in lock step restarts together, tile by tile)
  bool packed = wide_wanted && !given && !unpacked_flag && bk_can_pack(b);
  if (capturing) {
    // no conversion launch may be recorded: a replay would convert again, from words the replayed steps have outdated
    if (packed && !b.pk_valid) packed = false;
    if (!packed && !b.canon_valid) { p.kernel = -1; return p; }
  }
  if (autoreset) {  // envs that finish draw new episodes inside the kernel: the column grouping, the feature-row tile
    b.perm_valid = 0;  // list and the row counts go stale
    b.rm_valid = 0;
    b.hist_valid = 0;
  }
  if (capturing) {
    if (packed) b.graph_packed = 1;
    else b.graph_canon = 1;
    if (autoreset) b.graph_autoreset = 1;
  }
  // the day every env is on after this call, while the host can know it: a plain step moves all of them to the next
  // day; the terminal step, an in-kernel autoreset, unknown state or a recorded graph end the knowledge of the DAY (lock
  // step itself survives them all: envs that are on one day finish, and restart, together)
  int32_t uni_next = (!autoreset && b.uni_t >= 0 && b.uni_t + 1 < b.uni_nd) ? b.uni_t + 1 : -1;
  if (bk_any_graph(b)) uni_next = -1;
  if (packed) {
    if (!b.pk_valid) {  // entering the packed form (once per episode): the canonical words are current
      d.pack_state();
      b.pk_valid = 1; b.poisoned = 0;
      p.converted = 1;
    }
    p.kernel = W2A_BK_STEP_PACKED;
    b.canon_valid = 0;
    b.uni_t = uni_next;
    b.last_step_kernel = W2A_BK_STEP_PACKED;
    return p;
  }
  if (wide_wanted) p.kernel = W2A_BK_STEP_WIDE;
  if (!b.canon_valid) p.converted = 2;
  bk_ensure_canonical(b, d);
  bk_canonical_modified(b, true);
  b.uni_t = uni_next;
  b.last_step_kernel = p.kernel;
  return p;
}
// The launch bk_step planned did not happen (hipLaunchKernel failed): the state is what it was before the call, except
// that a conversion which did run has left BOTH forms current. `before` = the handle's book before bk_step.
static inline void bk_step_rollback(W2aBook &b, const W2aBook &before, const BkStepPlan &p) {
  const W2aBook after = b;  // conservative: keep what was recorded
  b = before;
  b.graph_canon = after.graph_canon; b.graph_packed = after.graph_packed; b.graph_autoreset = after.graph_autoreset;
  if (bk_any_graph(b)) b.uni_t = -1;
  if (p.converted == 1) { b.pk_valid = 1; b.poisoned = 0; }
  if (p.converted == 2) b.canon_valid = 1;
}

// w2a_rollout / w2a_rollout_posterior_mean: n_steps days, or to the end of every env's episode. A batch in lock step
// stays in lock step: every env runs the same days, or all of them reach their last day. Returns whether the batch
// is known to be in lock step (the matrix-core kernel needs it; it reads the day from the state itself).
template <class Dev>
static inline bool bk_rollout_begin(W2aBook &b, Dev &d, int32_t n_steps) {
  const int32_t day = b.uni_t;
  bk_ensure_canonical(b, d);
  bk_canonical_modified(b, true);
  b.uni_t = (day >= 0 && day + n_steps < b.uni_nd && !bk_any_graph(b)) ? day + n_steps : -1;
  return b.lock != 0;
}
// which sampled-reward rollout kernel serves the call (fixes: any W2A_FIX_* bit set)
static inline int bk_rollout_kernel(W2aBook &b, bool lock, bool fixes, bool mfma_built, bool wide_built) {
  int k = (wide_built && b.has_order) ? W2A_BK_ROLLOUT_WIDE : W2A_BK_ROLLOUT_4LANE;
  if (b.rm_valid && b.has_order && !fixes && lock && mfma_built) k = W2A_BK_ROLLOUT_MFMA;
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
  b.hist_valid = 0;  // ... and so are the per-env ranks inside the feature rows
}
static inline void bk_grouped(W2aBook &b) { b.perm_valid = b.graph_autoreset ? 0 : 1; }
// w2a_rollout_order_attach with ANOTHER workspace: it holds neither row counts nor (drops_order) a visiting order yet
static inline void bk_order_attach(W2aBook &b, bool drops_order = false) {
  b.has_order_ws = 1; b.hist_valid = 0;
  if (drops_order) { b.has_order = 0; b.rm_valid = 0; }
}
static inline void bk_order_set(W2aBook &b) { b.has_order = 1; b.rm_valid = 0; }
static inline void bk_rm_prepared(W2aBook &b) { b.rm_valid = b.graph_autoreset ? 0 : 1; }

// w2a_invalidate: the caller has overwritten the state buffer (its canonical part): forget every derived form
static inline void bk_invalidate(W2aBook &b) {
  b.pk_valid = 0; b.canon_valid = 1; b.lock = 0; b.uni_t = -1; b.perm_valid = 0;
  b.rm_valid = 0;  // feature rows may have changed behind the handle: the matrix-core rollout's tile list is stale
  b.hist_valid = 0;
  b.poisoned = 0;  // the caller's copy covered the mirror's day words too: whatever they hold now, it is not the poison
                   // (found by tools/sequence_fuzz.py, seed 505 sequence 357: a checkpoint restored over a poisoned
                   // mirror brought real days back, and a replayed packed step stepped stale state instead of refusing)
}

#endif  // W2A_BOOKKEEPING_H
