/* w2a.h -- C ABI of the MI355X-native vectorised HeatAlertEnv hot path (libw2a.so).
 *
 * The reference (NSAPH-Projects/weather2alert) is pure Python and exposes no FFI; its
 * boundary for this path is the Gymnasium Env API of src/weather2alert/env.py. Each entry
 * point below names the reference lines it replaces. The Python host class
 * (weather2alert_amd/env.py) keeps the reference's ctor/reset/step surface and calls these
 * through ctypes with torch-ROCm tensor data_ptr()s; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (PyTorch) unless marked host.
 *     The library allocates nothing on the device and frees nothing but the handle.
 *   - All calls are asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream). Two calls wait for the device: w2a_create (once per handle: it uploads the slot map and
 *     scans the tables on the NULL stream, then hipDeviceSynchronize) and w2a_read_status (waits for `stream`
 *     only, to read the status word back). Nothing else synchronises or allocates. What may be RECORDED into a
 *     hipGraph is w2a_step (w2a_state_bytes below says what a recording implies) and, while they need no
 *     conversion of the state's form, w2a_posterior_mean_reward / w2a_policy_actions / w2a_get_state; every
 *     other entry point that launches work fails with W2A_ERR_STATE while `stream` is capturing -- a replay
 *     would run it without the handle's bookkeeping (episode boundaries inside a graph: W2A_STEP_AUTORESET).
 *   - Return value: 0 = W2A_OK, negative = error (w2a_last_error() gives the text). Nothing
 *     throws across the ABI. Arguments are validated on the host before any launch; values
 *     that live in device arrays (episode tuples, actions) are range-checked inside the
 *     kernels, which clamp them and set a bit in the device status word instead of faulting.
 *   - A handle is not thread-safe; one handle per (process, device).
 */
#ifndef W2A_H
#define W2A_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define W2A_ABI_VERSION 18
#define W2A_ROW_FLOATS 32 /* floats per feature / weight row: one 128-B line */

enum {
  W2A_OK = 0,
  W2A_ERR_ARG = -1,      /* NULL pointer, non-positive size, bad enum */
  W2A_ERR_SCHEMA = -2,   /* table dims / slot layout the kernels cannot serve */
  W2A_ERR_HIP = -3,      /* a HIP runtime call failed */
  W2A_ERR_STATE = -4     /* state buffer too small / misaligned */
};

/* bits of the device status word (w2a_read_status) */
enum {
  W2A_ST_BAD_EPISODE = 1, /* reset tuple out of range (reference: KeyError env.py:127 / ValueError :121) */
  W2A_ST_BAD_ACTION = 2,  /* action not in {0,1} (reference action_space = Discrete(2), env.py:95) */
  W2A_ST_STEP_AFTER_DONE = 4, /* step() on a finished episode without autoreset */
  W2A_ST_STALE_GRAPH = 8  /* a replayed hipGraph holds a w2a_step on the packed lock-step form of the state, and that form
                             could not be kept current (the batch left lock step: a masked reset, a restored
                             checkpoint): the replayed step did NOTHING -- see w2a_state_bytes */
};

/* action buffer element types accepted by w2a_step */
enum { W2A_ACT_I32 = 0, W2A_ACT_I64 = 1, W2A_ACT_U8 = 2 };

/* w2a_step flags */
enum {
  W2A_STEP_AUTORESET = 1, /* same-step autoreset with the device RNG (needs w2a_set_autoreset): envs whose terminal step
                             has just run draw their next episode inside the step kernel and return its first
                             observation -- for batches that are not in lock step, and for loops recorded into a
                             hipGraph, where no reset can be launched between two steps (both step kernels serve it; a
                             batch that IS in lock step restarts together and stays on the packed form) */
  W2A_STEP_NO_OBS = 2,    /* reward-only: skip the observation write */
  W2A_STEP_CLASSIC = 8,   /* force the 4-lanes-per-env kernel where the 64-envs-per-wave one would be chosen (same
                             results up to the order of the fp64 additions; for A/B measurements and tests) */
  W2A_STEP_WIDE = 32,     /* force the 64-envs-per-wave kernel for small batches too (by default it serves batches of
                             >= 131 072 envs, the 4-lanes-per-env kernel smaller ones: the faster one on MI355X) */
  W2A_STEP_UNPACKED = 128, /* do not use the lock-step mirror of the per-env state (below; same results; for A/B
                             measurements and tests) */
  W2A_STEP_NEXT_STEP = 256, /* with W2A_STEP_AUTORESET: the restart happens on the call AFTER the terminal step (Gymnasium's
                             AutoresetMode.NEXT_STEP): the terminal step leaves the env finished with its stale
                             observation; the next call ignores that env's action, draws its next episode and returns
                             the episode's first observation with reward 0 and done 0 */
  W2A_STEP_NO_CAPTURE = 512, /* the caller drives episode boundaries from the host (it counts days and launches a reset
                             after the terminal step): such a loop cannot be recorded into a hipGraph -- a replay would
                             reset at a fixed position of the graph, or never -- so the call fails with W2A_ERR_STATE
                             while `stream` is capturing instead of recording a step (no cost otherwise: the capture
                             status is queried anyway) */
  W2A_STEP_SKIP_FINISHED = 64, /* with W2A_STEP_REWARD_GIVEN: envs whose episode is over are left untouched (reward
                             written as 0, done 1, state / return / observation unchanged, no status bit): policy
                             loops over batches that are not in lock step */
  W2A_STEP_REWARD_GIVEN = 16 /* `reward` is an INPUT: it already holds today's reward of every env
                             (w2a_posterior_mean_reward on the same state and actions); the step does everything
                             else of env.py:238-262 and accumulates that reward into the episode return */
};

/* budget sampling of reset(sample_budget=..., sample_budget_type=...), env.py:172-177 */
enum { W2A_BUDGET_FIXED = 0, W2A_BUDGET_LESS_THAN = 1, W2A_BUDGET_CENTERED = 2 };

/* Dense tables, compiled on the host by weather2alert_amd/tables.py.
 *
 * Replaces what HeatAlertEnv.__init__ builds (env.py:49-85): the merged (fips, year, date)
 * feature frame and the posterior coefficient tensors.
 *
 * Internal slot layout of a 32-float row (how lanes split a row is a kernel detail, not part of the ABI):
 *   slots  0..23  table-sourced columns (reward features first, in merged-column order)
 *   slots 24..27  run-time fields: alert_lag1, alert_streak, remaining_budget, alert_2wks(agent)
 *   slot   28     25th table-sourced column if the schema has one (else 0)
 *   slot   29     bias input (the table stores 1.0)
 *   slot   30     0/1 flag "heat_qi > 0.5" (the effectiveness gate of env.py:218, decided by the table
 *                 compiler on the file's float64 value), zero coefficient; kernels test it with > 0.5f
 *   slot   31     zero
 * W rows use the same slots (zero where a slot has no coefficient), so a reward logit is a
 * plain 32-wide dot product. obs_slot[j] maps observation column j (reference order,
 * env.py:186-195: the 28 episode columns then 'alert_2wks') to its slot.
 */
typedef struct w2a_tables {
  const float *X;                 /* [T][S_w*Y][32]  day-major feature rows            */
  const int32_t *n_days;          /* [S_w*Y]  episode length, 0 = (county, year) absent */
  const int32_t *B0;              /* [S_w*Y]  default budget = remaining_budget at day 0 (env.py:169) */
  const float *W;                 /* [S*n_samples][2][32]  head 0 baseline, 1 effectiveness */
  const int32_t *fips_to_weather; /* [S]  weight column -> county row of X, -1 = no weather */
  const int32_t *sim_cnt;         /* [S]  |similar(county) ∩ fips_list|  (env.py:115-117)  */
  int32_t T, S_w, Y, S, n_samples;
  int32_t n_obs;                  /* observation width (29 with the reference schema)    */
  int32_t obs_slot[W2A_ROW_FLOATS]; /* obs column -> slot, first n_obs entries valid       */
  int32_t slot_heat_qi;           /* slot of the 'heat_qi' feature (informational; the gate reads slot 30) */
  /* optional, for the corrected-semantics flags (w2a_set_semantics): */
  const int32_t *sim_ptr;         /* [S+1] CSR of similar(county) ∩ fips_list, confounders order (W2A_FIX_AUGMENT)     */
  const int32_t *sim_idx;         /* weight-column index of each similar county                                       */
  int32_t slot_alerts_2wks;       /* slot of the historical 'alerts_2wks' column, -1 = absent (W2A_FIX_ALERTS_2WKS)    */
  /* optional (NULL = off): the gate flags of slot 30 as a bitmap, [T][gate_words] uint32, bit (row & 31) of word
   * row >> 5 for row = county_w * Y + year_i. Lets the 64-envs-per-wave step kernel know in its first phase whether
   * the effectiveness row is needed at all (alert today AND heat_qi > 0.5, env.py:218-221) before any row is gathered */
  const uint32_t *gate_bits;
  int32_t gate_words;             /* words per day = ceil(S_w * Y / 32)                                               */
} w2a_tables;

typedef struct w2a_env w2a_env; /* opaque handle: pointers + dims only */

/* Decoded per-env state, for tests / checkpointing: every field is an int32 [num_envs]
 * device array supplied by the caller (NULL = skip). */
typedef struct w2a_state_view {
  int32_t *t, *used, *streak, *hist14, *last_actual, *at_budget, *budget, *n_days;
  int32_t *county_w, *year_i, *coef_col, *sample, *sticky_budget, *episode_no;
  int32_t *finished;     /* 1 once the terminal step of the current episode has run (done was returned) */
  float *episode_return; /* running return of the current episode */
} w2a_state_view;

int w2a_abi_version(void);
const char *w2a_last_error(void);

/* Bytes of caller-owned device memory one handle needs for `num_envs` envs (256-B aligned): 40 B per env of
 * canonical state (episode record, read-only step constants, counters + return) and a 16-B lock-step mirror (+ one day
 * word per 64 envs). While the handle knows the batch to be in lock step -- every env reset together by an unmasked
 * reset, one episode length for every (county, year), plain steps and rollouts since -- the day and the episode length
 * are the same for every env; the 64-envs-per-wave step kernel then streams 8 + 8 B of packed state per env in and 8 B
 * out instead of 12 + 12 and 12, with the day in the mirror's per-tile day word (needs T <= 255, S < 65536, n_samples
 * <= 1024, S_w * Y < 2^22 -- table properties; budgets may be anything: those the mirror's 16-bit field cannot hold are
 * read from the canonical words inside the kernel; other tables use the canonical arrays). The library converts between
 * the two forms by itself whenever an entry point needs the other one; a caller that rewrites the state buffer behind
 * the library's back (checkpoint restore) must call w2a_invalidate.
 * Stream capture. Every step kernel reads the day from device memory, so a w2a_step recorded into a hipGraph steps
 * correctly on every replay. What a recording fixes is the FORM of the state its kernel steps (no conversion between the
 * forms is ever recorded: w2a_step fails with W2A_ERR_STATE if the form its kernel needs is not current -- step once
 * eagerly, or call w2a_get_state, before capturing). From then on the handle keeps that form current at the end of
 * every call, so that a replay may come at any time:
 *   - recorded on the packed form (a lock-step batch of >= 131 072 envs, or W2A_STEP_WIDE, after one eager step): the
 *     mirror stays the primary form; calls that work on the canonical words (resets, rollouts, state reads) convert
 *     back before they return. Where the batch can no longer be packed (a masked reset, w2a_invalidate) the mirror is
 *     marked stale on the device and a replay of the recorded step does nothing but raise W2A_ST_STALE_GRAPH -- until
 *     the next whole-batch reset (of any kind: device RNG or the caller's tuples, with or without budgets) makes it
 *     packable again.
 *   - recorded on the canonical form: the handle never uses the packed form again.
 *   Replays advance days behind the host's back: a handle with a recorded step answers W2A_Q_LOCKSTEP_DAY with -1 (it
 *   still knows WHETHER the batch is in lock step, which is all the packed step and the matrix-core rollout need).
 *   Lock step survives a recorded W2A_STEP_AUTORESET step too (envs that are on one day finish, and restart, together:
 *   W2A_Q_LOCKSTEP stays 1 and the batch stays on the packed form); what such a handle never again reports valid is
 *   what belongs to particular EPISODES, which replays re-draw behind the host's back: the day, the column grouping
 *   (w2a_posterior_mean_reward), the tile list (w2a_rollout_mfma_prepare), the row counts of the visiting order.
 * The decisions are plain C++ in csrc/w2a_bookkeeping.h (run on the CPU under sanitizers, randomly and exhaustively, by
 * tests/test_bookkeeping_cpu.py). */
size_t w2a_state_bytes(int64_t num_envs);

/* Replaces HeatAlertEnv.__init__ (env.py:20-105) for `num_envs` envs whose global ids are
 * env_gid0 .. env_gid0+num_envs-1 (the device RNG is keyed by global id, so results do not
 * depend on how envs are sharded over GPUs). `tables` (host struct of device pointers) is
 * copied; `state` must stay alive until w2a_destroy. `status` is a caller-owned int32. */
int w2a_create(const w2a_tables *tables, int64_t num_envs, int64_t env_gid0, void *state, size_t state_bytes,
               int32_t *status, w2a_env **out);
void w2a_destroy(w2a_env *env);

/* Replaces reset() (env.py:133-184) with the episode tuples chosen by the caller (the host
 * replays NumPy's draws for seed parity, or injects them): per env the weather county row,
 * year index, coefficient column (env.py:117/121), posterior sample (env.py:160) and budget
 * (env.py:167-178). mask (uint8, nullable) selects which envs reset. Writes the first
 * observation rows into obs [num_envs][n_obs] (nullable). */
int w2a_reset(w2a_env *env, const int32_t *county_w, const int32_t *year_i, const int32_t *coef_col,
              const int32_t *sample, const int32_t *budget, const uint8_t *mask, float *obs, void *stream);

/* reset() with every draw of env.py:145-177 made on the device from a counter-based RNG keyed
 * by (seed, global env id, per-env episode number). location < 0 draws the county
 * (env.py:151-152), otherwise it is the weight-column index of the requested county.
 * budget_kw < 0 means "budget=None". Budget stickiness (env.py:167-170) is kept per env
 * unless `sticky` is 0. restart_episodes != 0: the (selected) envs' episode counters restart at 0, so a
 * reset with an explicit seed reproduces the same episodes every time (env.py:143-145 re-creates the
 * Generator from the seed); 0: the counters advance, which is what an autoreset does. */
int w2a_reset_device_rng(w2a_env *env, uint64_t seed, int32_t location, int augment, int32_t budget_kw,
                         int sample_budget_mode, int sticky, int restart_episodes, const uint8_t *mask, float *obs,
                         void *stream);

/* Parameters the same-step autoreset of w2a_step uses (same meaning as above). They travel to the step kernels as
 * arguments: a w2a_step recorded into a hipGraph keeps the parameters it was recorded with. */
int w2a_set_autoreset(w2a_env *env, uint64_t seed, int32_t location, int augment, int32_t budget_kw,
                      int sample_budget_mode, int sticky);

/* Replaces step() (env.py:238-262) for all envs: budget gate, history update, feature-row
 * gather, the two 28-term logits against the env's posterior draw, reward, termination, and
 * the next observation. actions: [num_envs] of `action_dtype`. obs [num_envs][n_obs] f32,
 * reward [num_envs] f32, done [num_envs] uint8. On a terminal step without autoreset the
 * obs rows are left untouched (the reference returns the stale observation, env.py:257-262);
 * with autoreset they hold the new episode's first observation and last_return (nullable,
 * f32 [num_envs]) receives the finished episode's return. */
int w2a_step(w2a_env *env, const void *actions, int action_dtype, float *obs, float *reward, uint8_t *done,
             float *last_return, int flags, void *stream);

/* episode_order="sorted" (no reference counterpart; opt-in): after a reset, relabel the envs so that env
 * indices follow the coefficient row (column, draw); envs of one row keep their order (a stable sort). The
 * multiset of episodes is unchanged -- only which env index holds which episode -- but neighbouring envs now
 * share table lines, which the step kernel's gathers turn into L2 hits. The whole per-env record moves
 * (episode tuple, budget, sticky budget, episode number).
 * workspace: caller-owned, w2a_sort_workspace_bytes(num_envs) bytes, 256-B aligned. */
size_t w2a_sort_workspace_bytes(int64_t num_envs);
int w2a_sort_episodes(w2a_env *env, void *workspace, size_t workspace_bytes, void *stream);
/* The same relabelling fused into the whole-batch device-RNG reset that precedes it (what episode_order="sorted" does once
 * per episode): w2a_reset_device_rng + w2a_sort_episodes + w2a_observe in three launches and no moved record. Pass 1 draws
 * every env's next episode and keeps only its sort key -- the coefficient row (column, draw) -- with the env's sticky
 * budget and episode number; a stable radix sort of (key, env index); pass 2 is k_reset with "index e receives the
 * episode env src[e] draws" (same global id, sticky budget and episode number as that env's own draw: the result is bit
 * for bit that of the three calls above) and writes state and first observations (obs nullable) in place.
 * Arguments as w2a_reset_device_rng without a mask; workspace as w2a_sort_episodes. Returns 1 (nothing done) when the key
 * does not fit 32 bits (S * n_samples > 2^32): the caller runs the three calls instead. */
int w2a_reset_device_rng_sorted(w2a_env *env, uint64_t seed, int32_t location, int augment, int32_t budget_kw,
                                int sample_budget_mode, int sticky, int restart_episodes, float *obs, void *workspace,
                                size_t workspace_bytes, void *stream);

/* reward_mode = "posterior_mean" (the legacy env's eval mode, _deprecated/env.py:332-342: `posterior_indices =
 * np.arange(n_posterior_samples) if eval_mode`, `np.mean([_get_reward(i, ...)])`, on today's reward form
 * env.py:197-226): the reward of every env is the mean over ALL posterior draws of its coefficient column
 * instead of the one draw of the episode. One grouped fp64 contraction per step: per column
 * [envs x 32 slots] * [32 slots x 2 heads x n_samples draws], sigmoid / gate / mean epilogue.
 *   w2a_group_by_column        after EVERY reset -- and after anything else that changes which episode an env index
 *                              holds: w2a_sort_episodes, a w2a_step with W2A_STEP_AUTORESET (both mark the grouping
 *                              stale, and w2a_posterior_mean_reward then refuses to run) --: sorts the env ids by coefficient column into `workspace`
 *                              (caller-owned, w2a_group_workspace_bytes(num_envs, S, n_samples), 256-B aligned, must
 *                              stay alive while w2a_posterior_mean_reward is used) and writes a pre-scaled fp64 copy
 *                              of W there. W rows that give slot 28, 30 or 31 a coefficient are honoured (w2a_create
 *                              scans for them once) at the price of a wider contraction;
 *   w2a_posterior_mean_reward  before w2a_step(..., W2A_STEP_REWARD_GIVEN) with the SAME actions: writes
 *                              reward [num_envs] f32 from the pre-step state. Same budget gate as the step
 *                              (env.py:242-246): an alert attempted at budget counts as no alert. */
/*   w2a_set_posterior_kernel   which kernel computes the contraction (same results to ~1e-7; speed differs):
 *                              W2A_PM_VECTOR (default) fp64 FMAs on the vector ALU with DPP-broadcast coefficients,
 *                              W2A_PM_MATRIX_F64 the fp64 matrix-core form (v_mfma_f64_16x16x4_f64; on MI355X the fp64
 *                              matrix rate equals the fp64 vector rate, so it is the slower of the two),
 *                              W2A_PM_MATRIX_I8 the int8 matrix-core form (v_mfma_i32_16x16x64_i8 on exact fixed-point
 *                              digits of both operands, int32 accumulation; logits within ~2e-6 by an a-priori bound,
 *                              columns outside the fixed-point range take an exact fp64 path inside the kernel).
 *                              The choice also decides whether w2a_rollout_posterior_mean's one-launch kernel applies
 *                              (it is built on the vector form). */
enum { W2A_PM_VECTOR = 0, W2A_PM_MATRIX_F64 = 1, W2A_PM_MATRIX_I8 = 2 };
int w2a_set_posterior_kernel(w2a_env *env, int kernel);
size_t w2a_group_workspace_bytes(int64_t num_envs, int32_t S, int32_t n_samples);
int w2a_group_by_column(w2a_env *env, void *workspace, size_t workspace_bytes, void *stream);
int w2a_posterior_mean_reward(w2a_env *env, const void *actions, int action_dtype, float *reward, void *stream);

/* First observation (env.py:181) of every env from its packed state; valid right after a reset
 * (t == 0 for every env, else W2A_ST_STEP_AFTER_DONE is raised). Used after w2a_sort_episodes. */
int w2a_observe(w2a_env *env, float *obs, void *stream);

/* Opt-in corrections of reference quirks. Default 0 = faithful to env.py, which is
 * what every parity claim refers to; each bit is independent:
 *   ALERTS_2WKS (Q1) the agent's 14-day alert count replaces the historical 'alerts_2wks' column (env.py:191
 *                    writes it to a new key instead), so it reaches the reward through that coefficient
 *   LAG         (Q3) alert_lag1 is yesterday's actual action (env.py:190 reads the buffer after today's append)
 *   PENALTY     (Q5) an alert attempted at budget costs reward -1 (env.py:223-224 is dead code)
 *   OBS         (Q6) step() returns the row of the next day with the updated state (env.py:257-259 returns the
 *                    current day's row and a stale row on the terminal step)
 *   AUGMENT     (Q8) similar_climate_counties uses the drawn county's weather and its true coefficient column
 *                    (env.py:116-127 indexes the filtered list and keeps the requested county's weather);
 *                    applies to device-RNG resets, host-tuple resets pass what they want
 *   (Q9, per-episode budgets, is the `sticky = 0` argument of the reset entry points.) */
enum { W2A_FIX_ALERTS_2WKS = 1, W2A_FIX_LAG = 2, W2A_FIX_PENALTY = 4, W2A_FIX_OBS = 8, W2A_FIX_AUGMENT = 16,
       W2A_FIX_ALL = 31 };
int w2a_set_semantics(w2a_env *env, uint32_t fixes);

/* On-device policy rollout (replaces a Python loop of `action = policy(obs); env.step(action)`
 * such as env.py:265-277): every env runs up to n_steps days, or to the end of its episode, inside one launch
 * with its coefficient rows held in registers. The policy sees what the reference's agent would see before
 * acting on day t: the lagging observation (row of day t-1), the remaining budget and the day. */
enum { W2A_POLICY_NEVER = 0, W2A_POLICY_ALWAYS = 1, W2A_POLICY_BERNOULLI = 2, W2A_POLICY_THRESHOLD = 3,
       W2A_POLICY_TABLE = 4 };
typedef struct w2a_policy {
  int32_t kind;
  float p;                /* BERNOULLI: P(alert); drawn from the counter RNG keyed (seed, env id, episode, day) */
  int32_t obs_col;        /* THRESHOLD: observation column (reference order) of a table-sourced feature ...      */
  float threshold;        /* ... alert iff obs[obs_col] > threshold                                              */
  int32_t obs_lag;        /* THRESHOLD: 1 = the agent sees yesterday's row (faithful, Q6); 0 = today's row        */
  int32_t require_budget; /* 1: never attempt an alert with remaining_budget <= 0                                */
  const uint8_t *table;   /* TABLE: device uint8 [T][table_R], action = table[day][min(remaining_budget, R-1)]   */
  int32_t table_R;
  uint64_t seed;
} w2a_policy;
/* Outputs (device, nullable): ret_out f32 [n] rewards summed over the days run by this call, alerts_out i32 [n]
 * alerts issued, attempts_over_budget i32 [n] alerts attempted at budget (silently dropped, Q5), alert_mask /
 * attempt_mask u32 [n][mask_words] bit d = alert issued / attempted on day d (the reference's actual_ and
 * attempted_alert_buffer, env.py:239,248), last_return f32 [n] episode return of envs that finished,
 * ret_snapshot f32 [n] the running episode return after the step that leaves t == n_days - 2, the moment the
 * reference's logging callbacks read the env (callbacks.py:47-48,128-132; untouched if that step is not in this call). */
int w2a_rollout(w2a_env *env, const w2a_policy *policy, int32_t n_steps, float *ret_out, int32_t *alerts_out,
                int32_t *attempts_over_budget, uint32_t *alert_mask, uint32_t *attempt_mask, int32_t mask_words,
                float *last_return, float *ret_snapshot, void *stream);
/* Optional, speed only: let w2a_rollout visit the envs in the order of their feature rows (envs that share a
 * (county, year) sit in the same wave and read the same table lines every day). Results are those of any other
 * order -- per-env outputs, RNG streams and state stay indexed by env id. Call after a reset (the order of an
 * earlier episode stays valid as a permutation, it is just no longer sorted). workspace: caller-owned,
 * w2a_rollout_order_workspace_bytes(num_envs, S_w * Y) bytes, 256-B aligned, must stay alive while w2a_rollout is
 * used -- and, once attached, until w2a_destroy.
 *   w2a_rollout_order_attach  (host only) hands the workspace over ahead of time: from then on every whole-batch reset
 *                             also counts the envs of each feature row and ranks every env inside its row there (the
 *                             first pass of the counting sort, hidden behind the reset's observation stores), and
 *                             w2a_rollout_order is left with a scan and an atomic-free placement. w2a_rollout_order
 *                             attaches its workspace itself on first use. */
size_t w2a_rollout_order_workspace_bytes(int64_t num_envs, int64_t table_rows);
int w2a_rollout_order_attach(w2a_env *env, void *workspace, size_t workspace_bytes);
int w2a_rollout_order(w2a_env *env, void *workspace, size_t workspace_bytes, void *stream);

/* Optional, speed only: let w2a_rollout compute the table-sourced part of both logits -- 27 of their 30 terms, which do
 * not depend on the agent's actions -- for all envs of a (county, year) and 16 days at a time on the int8 matrix cores
 * (exact fixed-point digits, int32 accumulation; csrc/w2a_rollout_mfma.hip.h), leaving 3 + 3 fp64 FMAs per env-day on
 * the vector ALU instead of 30 + 30. Call after w2a_rollout_order of the episode (it lists the tiles of envs sharing a
 * feature row from that order and, once per table, builds a digit table of W in the workspace). Used by w2a_rollout
 * while the handle knows the batch to be in lock step and no corrected-semantics flag is set; results agree with the
 * other rollout kernels to the accuracy of the fixed point (returns within ~1e-6 relative), integers identical.
 * workspace: caller-owned, w2a_rollout_mfma_workspace_bytes(...) bytes, 256-B aligned, alive while w2a_rollout is used. */
size_t w2a_rollout_mfma_workspace_bytes(int64_t num_envs, int64_t table_rows, int32_t S, int32_t n_samples);
int w2a_rollout_mfma_prepare(w2a_env *env, void *workspace, size_t workspace_bytes, void *stream);

/* w2a_rollout with the posterior-mean reward (w2a_posterior_mean_reward's value every day), whole episode in one launch:
 * same arguments and outputs as w2a_rollout; needs w2a_group_by_column after the last reset. Built on the kernel selected
 * by w2a_set_posterior_kernel: W2A_PM_MATRIX_I8 (k_pm_rollout_i8) or W2A_PM_VECTOR (k_pm_rollout). Returns 1 (nothing
 * done) when no one-launch kernel applies -- more posterior draws than one LDS staging pass holds (112), the vector form
 * with coefficients on slots 28/30/31, W2A_PM_MATRIX_F64 -- and the caller runs the per-day sequence w2a_policy_actions,
 * w2a_posterior_mean_reward, w2a_step. */
int w2a_rollout_posterior_mean(w2a_env *env, const w2a_policy *policy, int32_t n_steps, float *ret_out,
                               int32_t *alerts_out, int32_t *attempts_over_budget, uint32_t *alert_mask,
                               uint32_t *attempt_mask, int32_t mask_words, float *last_return, float *ret_snapshot,
                               void *stream);

/* One day of a built-in policy: actions i32 [n] of every env from its pre-step state -- the same policy evaluation,
 * lagging observation and budget gate as w2a_rollout (finished envs get action 0) -- for policy loops whose step is
 * w2a_step, e.g. with w2a_posterior_mean_reward. Nullable accumulators, updated for today: alerts i32 [n] += alert
 * issued, attempts_over_budget i32 [n] += alert attempted at budget, alert_mask / attempt_mask u32 [n][mask_words]
 * |= bit (day) -- the caller zeroes them before the first day. */
int w2a_policy_actions(w2a_env *env, const w2a_policy *policy, int32_t *actions, int32_t *alerts,
                       int32_t *attempts_over_budget, uint32_t *alert_mask, uint32_t *attempt_mask, int32_t mask_words,
                       void *stream);

/* Decode the packed state into the caller's arrays (see w2a_state_view). */
int w2a_get_state(w2a_env *env, const w2a_state_view *view, void *stream);

/* What the handle knows (host-side bookkeeping, no device work): the day every env is on if the batch is known to be in
 * lock step AND the host can know the day (-1 otherwise: not in lock step, past the terminal step, or a step of this
 * handle was recorded into a hipGraph); whether the tables allow the lock-step mirror at all; which of the two
 * forms of the step state is current; whether the batch is known to be in lock step. */
enum { W2A_Q_LOCKSTEP = 6,           /* 1: every env is known to be on the same day of an episode of the one length there is */
       W2A_Q_LOCKSTEP_DAY = 0, W2A_Q_PACKED_ELIGIBLE = 1, W2A_Q_PACKED_CURRENT = 2, W2A_Q_CANONICAL_CURRENT = 3,
       W2A_Q_LAST_ROLLOUT_KERNEL = 4 /* what the last w2a_rollout launched: -1 none yet, 0 k_rollout (4 lanes per env),
                                        1 k_rollout64 (lane = env), 2 k_rollout_mfma (int8 matrix cores) */,
       W2A_Q_LAST_STEP_KERNEL = 5    /* what the last w2a_step launched: -1 none yet, 0 k_step (4 lanes per env),
                                        1 k_step64 on the canonical state words, 2 k_step64 on the lock-step mirror */ };
int w2a_query(w2a_env *env, int what);

/* The caller has overwritten the state buffer (e.g. restored a checkpoint of its canonical part) on `stream`: forget
 * every derived form (lock-step mirror, column grouping, row counts, what is known about days). Host-side bookkeeping
 * only -- plus, on a handle with a recorded packed step, one small launch on `stream` that marks the mirror stale --:
 * no wait, nothing read back. (Until ABI 17 the call scanned the restored buffer for its largest budget and waited for
 * the stream, and w2a_set_budget_bound let a caller state one: the packed form held budgets in 16 bits and the handle had
 * to bound them from the host side. The packed kernel now serves any budget; both are gone.)
 * Fails with W2A_ERR_STATE while `stream` is recording a hipGraph. */
int w2a_invalidate(w2a_env *env, void *stream);

/* Synchronise `stream`, read and clear the device status word (host int out). */
int w2a_read_status(w2a_env *env, int32_t *status_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* W2A_H */
