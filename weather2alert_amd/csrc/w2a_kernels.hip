// w2a_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of the vectorised HeatAlertEnv.
//
// What is computed follows the reference src/weather2alert/env.py:
//   reset  :133-184 (+ _get_episode :107-131)      -> draw_episode() / k_reset
//   _get_obs :186-195, _get_reward :197-226, step :238-262 -> k_step (and k_rollout: many days per launch)
// How it is computed is MI355X-first:
//   * (k_step, k_reset, k_rollout) an env is served by a 4-lane group of a 64-wide wavefront (16 envs per wave, 64
//     per 256-thread workgroup); lane l owns floats 8l..8l+7 of the env's 128-byte feature row
//     and of its two 128-byte coefficient rows, so every gather is 16-B loads that together
//     cover whole 128-B lines (LANES = 8 / 2 were measured slower);
//   * feature rows are stored day-major ([T][county*year][32]); all envs of a lock-step
//     batch read the same ~1 MB day slice, which stays in each XCD's 4 MiB L2;
//   * the 28-term logits are accumulated in fp64 (products of f32 inputs are exact in
//     fp64, the sum carries ~1e-16 relative error, cf. the reference's float64 sum at
//     env.py:207-217) and reduced over the group's lanes with DPP quad_perm moves -- no LDS
//     traffic, no ds_bpermute;
//   * the packed [N][29] f32 observation rows of a wave (16 x 116 B = 1856 contiguous bytes)
//     are transposed through a 2-KB LDS tile and leave as 116 coalesced non-temporal 16-B stores;
//   * per-env state is two 16-B words (cold: episode tuple, hot: counters) read as group
//     broadcast loads; the hot word is written back as whole 128-B lines per wave;
//   * the effectiveness coefficient row is fetched only for envs that issue an alert today
//     (it enters the reward through eff * actual, env.py:221): half the gather traffic;
//   * workgroup -> env-tile mapping is XCD-aware (logical_block); k_step64 (w2a_step64.hip.h) is the lean
//     64-envs-per-wave form of the same step for large batches with faithful semantics; the posterior-mean reward contraction
//     (w2a_posterior.hip.h) exists as a vector-ALU kernel (default) and as matrix-unit (MFMA) kernels, selected at
//     run time by w2a_set_posterior_kernel; the matrix unit's other user is the policy rollout k_rollout_mfma
//     (w2a_rollout_mfma.hip.h: the table-sourced part of both logits per (county, year) tile as int8 MFMAs).
//
// No fallback path exists: without this library (or without a ROCm device) constructing an env raises.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "w2a.h"

#include "w2a_common.hip.h"
#include "w2a_step.hip.h"
#include "w2a_step64.hip.h"
#include "w2a_posterior.hip.h"
#include "w2a_posterior_i8.hip.h"
#include "w2a_reset.hip.h"
#include "w2a_rollout.hip.h"
#include "w2a_rollout_i8.hip.h"
#include "w2a_rollout_mfma.hip.h"
#include "w2a_sort.hip.h"

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                   \
  do {                                                                  \
    hipError_t _e = (expr);                                             \
    if (_e != hipSuccess) return fail(W2A_ERR_HIP, #expr ": %s", hipGetErrorString(_e)); \
  } while (0)

// k_rollout is compiled per (policy kind, day bitmaps wanted, corrected-semantics flags set)
// k_pm_rollout per (policy kind, day bitmaps wanted)
template <int KIND>
static void launch_pm_rollout_kind(bool masks, unsigned grid, hipStream_t s, const PmRolloutArgs &a) {
  if (masks) hipLaunchKernelGGL((k_pm_rollout<7, KIND, true>), dim3(grid), dim3(PMV_THREADS), 0, s, a);
  else hipLaunchKernelGGL((k_pm_rollout<7, KIND, false>), dim3(grid), dim3(PMV_THREADS), 0, s, a);
}
static void launch_pm_rollout(int kind, bool masks, unsigned grid, hipStream_t s, const PmRolloutArgs &a) {
  switch (kind) {
    case W2A_POLICY_ALWAYS: launch_pm_rollout_kind<W2A_POLICY_ALWAYS>(masks, grid, s, a); break;
    case W2A_POLICY_BERNOULLI: launch_pm_rollout_kind<W2A_POLICY_BERNOULLI>(masks, grid, s, a); break;
    case W2A_POLICY_THRESHOLD: launch_pm_rollout_kind<W2A_POLICY_THRESHOLD>(masks, grid, s, a); break;
    case W2A_POLICY_TABLE: launch_pm_rollout_kind<W2A_POLICY_TABLE>(masks, grid, s, a); break;
    default: launch_pm_rollout_kind<W2A_POLICY_NEVER>(masks, grid, s, a); break;
  }
}

#ifndef W2A_ROLLOUT_MFMA
#define W2A_ROLLOUT_MFMA 1  // 1: w2a_rollout uses k_rollout_mfma when w2a_rollout_mfma_prepare has run for the episode (A/B)
#endif
#ifndef W2A_ROLLOUT_WIDE
#define W2A_ROLLOUT_WIDE 1  // 1: lane = env day loop (k_rollout64) when a visiting order is set; 0: always 4 lanes per env
#endif
template <int KIND>
static void launch_rollout_kind(bool masks, bool fixes, unsigned grid, hipStream_t s, const RolloutArgs &a) {
  if (W2A_ROLLOUT_WIDE && a.order) {
    const unsigned g64 = (unsigned)((((a.n + BLOCK - 1) / BLOCK) + 7) / 8 * 8);
    if (masks && fixes) hipLaunchKernelGGL((k_rollout64<KIND, true, true>), dim3(g64), dim3(BLOCK), 0, s, a);
    else if (masks) hipLaunchKernelGGL((k_rollout64<KIND, true, false>), dim3(g64), dim3(BLOCK), 0, s, a);
    else if (fixes) hipLaunchKernelGGL((k_rollout64<KIND, false, true>), dim3(g64), dim3(BLOCK), 0, s, a);
    else hipLaunchKernelGGL((k_rollout64<KIND, false, false>), dim3(g64), dim3(BLOCK), 0, s, a);
    return;
  }
  if (masks && fixes) hipLaunchKernelGGL((k_rollout<KIND, true, true>), dim3(grid), dim3(BLOCK), 0, s, a);
  else if (masks) hipLaunchKernelGGL((k_rollout<KIND, true, false>), dim3(grid), dim3(BLOCK), 0, s, a);
  else if (fixes) hipLaunchKernelGGL((k_rollout<KIND, false, true>), dim3(grid), dim3(BLOCK), 0, s, a);
  else hipLaunchKernelGGL((k_rollout<KIND, false, false>), dim3(grid), dim3(BLOCK), 0, s, a);
}
static void launch_rollout(int kind, bool masks, bool fixes, unsigned grid, hipStream_t s, const RolloutArgs &a) {
  switch (kind) {
    case W2A_POLICY_ALWAYS: launch_rollout_kind<W2A_POLICY_ALWAYS>(masks, fixes, grid, s, a); break;
    case W2A_POLICY_BERNOULLI: launch_rollout_kind<W2A_POLICY_BERNOULLI>(masks, fixes, grid, s, a); break;
    case W2A_POLICY_THRESHOLD: launch_rollout_kind<W2A_POLICY_THRESHOLD>(masks, fixes, grid, s, a); break;
    case W2A_POLICY_TABLE: launch_rollout_kind<W2A_POLICY_TABLE>(masks, fixes, grid, s, a); break;
    default: launch_rollout_kind<W2A_POLICY_NEVER>(masks, fixes, grid, s, a); break;
  }
}

template <int KIND>
static void launch_rollout_mfma_kind(bool masks, unsigned grid, hipStream_t s, const RmArgs &ra) {
  if (masks) hipLaunchKernelGGL((k_rollout_mfma<KIND, true>), dim3(grid), dim3(64 * RM_WAVES), 0, s, ra);
  else hipLaunchKernelGGL((k_rollout_mfma<KIND, false>), dim3(grid), dim3(64 * RM_WAVES), 0, s, ra);
}
static void launch_rollout_mfma(int kind, bool masks, unsigned grid, hipStream_t s, const RmArgs &ra) {
  switch (kind) {
    case W2A_POLICY_ALWAYS: launch_rollout_mfma_kind<W2A_POLICY_ALWAYS>(masks, grid, s, ra); break;
    case W2A_POLICY_BERNOULLI: launch_rollout_mfma_kind<W2A_POLICY_BERNOULLI>(masks, grid, s, ra); break;
    case W2A_POLICY_THRESHOLD: launch_rollout_mfma_kind<W2A_POLICY_THRESHOLD>(masks, grid, s, ra); break;
    case W2A_POLICY_TABLE: launch_rollout_mfma_kind<W2A_POLICY_TABLE>(masks, grid, s, ra); break;
    default: launch_rollout_mfma_kind<W2A_POLICY_NEVER>(masks, grid, s, ra); break;
  }
}

extern "C" {

int w2a_abi_version(void) { return W2A_ABI_VERSION; }
const char *w2a_last_error(void) { return g_err; }

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// header (slot map) + cold 16 B + hot3 12 B + stepc 12 B per env + the lock-step mirror (8 + 8 B per env, one day word
// per 64-env tile), every array on its own 256-B boundary (for odd env counts the 12-B arrays would otherwise start at a
// 4-B boundary and their 12-B loads straddle lines)
size_t w2a_state_bytes(int64_t num_envs) {
  if (num_envs <= 0) return 0;
  const size_t n = (size_t)num_envs;
  return HDR_BYTES + align256(16 * n) + 2 * align256(12 * n) + 2 * align256(8 * n) + align256(4 * ((n + 63) / 64));
}

// ---- which form of the step state is current (StateArrays::pk_hot / pk_c): decided in w2a_bookkeeping.h; this is its
// device side, the two conversion launches on the caller's stream
struct HipDev {
  w2a_env *env;
  hipStream_t s;
  void pack_state() {
    hipLaunchKernelGGL(k_pack_state, dim3((unsigned)((env->n + 255) / 256)), dim3(256), 0, s, env->st, env->n);
  }
  void unpack_state(int32_t n_days) {
    hipLaunchKernelGGL(k_unpack_state, dim3((unsigned)((env->n + 255) / 256)), dim3(256), 0, s, env->st, env->n, n_days);
  }
  void poison_mirror() {
    const int64_t tiles = (env->n + 63) / 64;
    hipLaunchKernelGGL(k_poison_mirror, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, s, env->st, env->n);
  }
};
// end of an entry point that may have changed which form of the state is current (w2a_bookkeeping.h: bk_end_call)
static void end_call(w2a_env *env, hipStream_t s) {
  HipDev d{env, s};
  bk_end_call(env->bk, d);
}
// The entry points that launch something other than a step kernel are not recorded into hipGraphs: a replay would run
// them without the bookkeeping below (a recorded reset re-draws episodes under a column grouping / row counts the handle
// still calls valid; a recorded conversion converts again from words the replayed steps have outdated). Loops that need
// episode boundaries inside a graph use W2A_STEP_AUTORESET (include/w2a.h).
static bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (s && hipStreamIsCapturing(s, &cs) == hipSuccess) return cs != hipStreamCaptureStatusNone;
  (void)hipGetLastError();
  return false;
}
#define REFUSE_WHILE_CAPTURING(who, stream)                                                                            \
  do {                                                                                                                 \
    if (stream_is_capturing((hipStream_t)(stream)))                                                                    \
      return fail(W2A_ERR_STATE, who ": the stream is recording a hipGraph; only w2a_step can be recorded (episode "    \
                                     "boundaries inside a graph: W2A_STEP_AUTORESET)");                                \
  } while (0)
// something is about to read the canonical words; false (with the error text set): the read would have to record a
// conversion of the state's form into a hipGraph
static bool ensure_canonical(w2a_env *env, hipStream_t s, const char *who) {
  if (!env->bk.canon_valid && stream_is_capturing(s)) {
    fail(W2A_ERR_STATE, "%s: the state is in its packed lock-step form and a conversion cannot be recorded into a hipGraph", who);
    return false;
  }
  HipDev d{env, s};
  bk_ensure_canonical(env->bk, d);
  return true;
}
__global__ void k_table_scan(const int32_t *n_days, int32_t rows, int32_t *out) {  // out: min nd, max nd
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  atomicMin(&out[0], n_days[i]);
  atomicMax(&out[1], n_days[i]);
}

int w2a_create(const w2a_tables *t, int64_t num_envs, int64_t env_gid0, void *state, size_t state_bytes,
               int32_t *status, w2a_env **out) {
  if (!t || !state || !status || !out) return fail(W2A_ERR_ARG, "w2a_create: NULL argument");
  if (num_envs <= 0 || env_gid0 < 0) return fail(W2A_ERR_ARG, "w2a_create: num_envs must be > 0 and env_gid0 >= 0");
  if (!t->X || !t->n_days || !t->B0 || !t->W || !t->fips_to_weather || !t->sim_cnt)
    return fail(W2A_ERR_ARG, "w2a_create: NULL table pointer");
  if (t->T <= 0 || t->T > 1023 || t->S_w <= 0 || t->Y <= 0 || t->S <= 0 || t->n_samples <= 0)
    return fail(W2A_ERR_SCHEMA, "w2a_create: table dims out of range (need 0 < T <= 1023, positive S_w, Y, S, n_samples)");
  if ((int64_t)t->S_w * t->Y > 0x7FFFFFFFll / 2 || (int64_t)t->S * t->n_samples > 0x7FFFFFFFll / 2 ||
      (int64_t)t->T * t->S_w * t->Y > 0x7FFFFFFFll)
    return fail(W2A_ERR_SCHEMA, "w2a_create: table too large for 32-bit row indices");
  if ((int64_t)t->T * t->S_w * t->Y * (ROWF / 4) > 0x7FFFFFFFll || (int64_t)t->S * t->n_samples * (2 * ROWF / 4) > 0x7FFFFFFFll ||
      num_envs > (1ll << 27))
    return fail(W2A_ERR_SCHEMA, "w2a_create: tables / env count exceed the 32-bit offset range of the kernels");
  if (t->n_samples > (1 << SAMPLE_BITS) || t->S >= (1 << (32 - SAMPLE_BITS)))
    return fail(W2A_ERR_SCHEMA, "w2a_create: need n_samples <= 4096 and S < 2^20");
  if (t->n_obs <= 0 || t->n_obs > W2A_ROW_FLOATS - 1)
    return fail(W2A_ERR_SCHEMA, "w2a_create: n_obs must be in 1..31 (the observation tile keeps one scratch column)");
  if (state_bytes < w2a_state_bytes(num_envs)) return fail(W2A_ERR_STATE, "w2a_create: state buffer too small");
  if (((uintptr_t)state & 255) || ((uintptr_t)t->X & 15) || ((uintptr_t)t->W & 15))
    return fail(W2A_ERR_STATE, "w2a_create: state must be 256-B aligned, X and W 16-B aligned");
  int32_t slot_obs[ROWF];
  for (int i = 0; i < ROWF; ++i) slot_obs[i] = -1;
  for (int j = 0; j < t->n_obs; ++j) {
    int s = t->obs_slot[j];
    if (s < 0 || s >= ROWF || slot_obs[s] != -1) return fail(W2A_ERR_SCHEMA, "w2a_create: obs_slot is not an injective map into 0..31");
    slot_obs[s] = j;
  }
  w2a_env *h = new w2a_env();
  h->tb.X = reinterpret_cast<const float4 *>(t->X);
  h->tb.n_days = t->n_days;
  h->tb.B0 = t->B0;
  h->tb.W = reinterpret_cast<const float4 *>(t->W);
  h->tb.fips_to_weather = t->fips_to_weather;
  h->tb.sim_cnt = t->sim_cnt;
  h->tb.sim_ptr = t->sim_ptr;
  h->tb.sim_idx = t->sim_idx;
  h->tb.slot_hist2w = t->slot_alerts_2wks;
  h->tb.gate_bits = t->gate_bits;
  h->tb.gate_words = t->gate_words;
  if (t->gate_bits && (int64_t)t->gate_words * 32 < (int64_t)t->S_w * t->Y) {
    delete h;
    return fail(W2A_ERR_SCHEMA, "w2a_create: gate_words * 32 must cover S_w * Y rows");
  }
  h->tb.fixes = 0;
  h->tb.T = t->T; h->tb.S_w = t->S_w; h->tb.Y = t->Y; h->tb.S = t->S; h->tb.n_samples = t->n_samples;
  h->tb.n_obs = t->n_obs;
  h->n = num_envs;
  h->gid0 = env_gid0;
  h->slot_obs = reinterpret_cast<const int32_t *>(state);
  h->st.cold = reinterpret_cast<uint4 *>((char *)state + HDR_BYTES);
  h->st.hot3 = reinterpret_cast<u3 *>((char *)h->st.cold + align256(16 * (size_t)num_envs));
  h->st.stepc = reinterpret_cast<u3 *>((char *)h->st.hot3 + align256(12 * (size_t)num_envs));
  h->st.pk_hot = reinterpret_cast<uint2 *>((char *)h->st.stepc + align256(12 * (size_t)num_envs));
  h->st.pk_c = reinterpret_cast<uint2 *>((char *)h->st.pk_hot + align256(8 * (size_t)num_envs));
  h->st.pk_day = reinterpret_cast<uint32_t *>((char *)h->st.pk_c + align256(8 * (size_t)num_envs));
  bk_init(h->bk, t->T <= 255 && t->S < 65536 && t->n_samples <= 1024 && (int64_t)t->S_w * t->Y < (1 << 22), -1);
  h->status = status;
  h->has_autoreset = 0;
  h->perm = nullptr;
  h->order = nullptr;
  h->prep = nullptr;
  h->pm_kernel = W2A_PM_VECTOR;
  h->xmax_ws = nullptr;
  h->order_ws = nullptr; h->order_cnt = h->order_rank = h->order_start = h->order_tile_start = nullptr; h->rm_ws = nullptr;
  for (int j = 0; j < ROWF; ++j) h->obs_slot_host[j] = j < t->n_obs ? t->obs_slot[j] : -1;
  hipError_t e1 = hipMemcpy(state, slot_obs, sizeof(slot_obs), hipMemcpyHostToDevice);
  hipError_t e2 = hipMemset(status, 0, sizeof(int32_t));
  if (e1 != hipSuccess || e2 != hipSuccess) { delete h; return fail(W2A_ERR_HIP, "w2a_create: header upload failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
  int64_t blocks = (num_envs + 255) / 256;
  hipLaunchKernelGGL(k_init_state, dim3((unsigned)blocks), dim3(256), 0, 0, h->st, num_envs);
  hipError_t e0 = hipGetLastError();
  if (e0 != hipSuccess) { delete h; return fail(W2A_ERR_HIP, "w2a_create: k_init_state launch failed: %s", hipGetErrorString(e0)); }
  // header word 32: scratch for the one-off scan of the coefficient rows (see k_scan_tail_slots)
  int32_t *scan_flag = reinterpret_cast<int32_t *>(state) + ROWF;
  const int64_t w_rows = (int64_t)t->S * t->n_samples * 2;
  hipError_t e4 = hipMemset(scan_flag, 0, sizeof(int32_t));
  hipLaunchKernelGGL(k_scan_tail_slots, dim3((unsigned)((w_rows + 255) / 256)), dim3(256), 0, 0, h->tb.W, w_rows, scan_flag);
  hipError_t e3 = hipDeviceSynchronize();
  int32_t tail_used = 1;
  if (e3 == hipSuccess && e4 == hipSuccess) e3 = hipMemcpy(&tail_used, scan_flag, sizeof(int32_t), hipMemcpyDeviceToHost);
  if (e3 != hipSuccess || e4 != hipSuccess) { delete h; return fail(W2A_ERR_HIP, "w2a_create: init kernels failed: %s", hipGetErrorString(e3 != hipSuccess ? e3 : e4)); }
  h->w_tail_used = tail_used;
  {  // one episode length for every (county, year)? (eligibility of the lock-step mirror)
    int32_t scan[2] = {0x7FFFFFFF, 0};
    int32_t *d_scan = reinterpret_cast<int32_t *>(state) + ROWF + 1;  // header words 33..34
    hipError_t e5 = hipMemcpy(d_scan, scan, sizeof(scan), hipMemcpyHostToDevice);
    const int32_t rows = t->S_w * t->Y;
    hipLaunchKernelGGL(k_table_scan, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, t->n_days, rows, d_scan);
    if (e5 == hipSuccess) e5 = hipMemcpy(scan, d_scan, sizeof(scan), hipMemcpyDeviceToHost);
    if (e5 != hipSuccess) { delete h; return fail(W2A_ERR_HIP, "w2a_create: table scan failed: %s", hipGetErrorString(e5)); }
    h->bk.uni_nd = (scan[0] == scan[1] && scan[0] > 0) ? scan[0] : -1;
  }
  *out = h;
  return W2A_OK;
}

void w2a_destroy(w2a_env *env) { delete env; }

// tiles of ENVS_PER_BLOCK envs, rounded up to a multiple of 8 workgroups (one share per XCD, see logical_block)
static unsigned grid_for(int64_t n) {
  int64_t tiles = (n + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
  return (unsigned)(((tiles + 7) / 8) * 8);
}

static int launch_reset(w2a_env *env, ResetArgs &a, void *stream) {
  // arguments first: nothing below may fail for a reason the caller can fix once the bookkeeping has been told of the reset
  if (a.obs && ((uintptr_t)a.obs & 15)) return fail(W2A_ERR_ARG, "reset: obs must be 16-B aligned");
  REFUSE_WHILE_CAPTURING("reset / observe", stream);
  const W2aBook before = env->bk;
  {
    HipDev d{env, (hipStream_t)stream};
    bk_reset(env->bk, d, a.from_tuples == 2, a.mask != nullptr);
  }
  a.tb = env->tb; a.slot_obs = env->slot_obs; a.st = env->st;
  a.status = env->status; a.n = env->n; a.gid0 = env->gid0;
  hipError_t e = hipSuccess;
  if (env->bk.hist_valid && a.from_tuples != 2) {  // whole-batch reset, order workspace attached: row counts and per-env ranks come with it
    e = hipMemsetAsync(env->order_cnt, 0, 4 * (size_t)env->tb.S_w * env->tb.Y, (hipStream_t)stream);
    a.order_cnt = env->order_cnt; a.order_rank = env->order_rank;
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_reset, dim3(grid_for(env->n)), dim3(BLOCK), 0, (hipStream_t)stream, a);
    e = hipGetLastError();
  }
  if (e != hipSuccess) {
    // the reset did not happen: the state is what it was, except that a conversion bk_reset asked for (a masked reset or
    // w2a_observe on the packed form) did run and left both forms current; derived structures are dropped (conservative)
    bk_reset_rollback(env->bk, before, a.from_tuples == 2, a.mask != nullptr);
    end_call(env, (hipStream_t)stream);
    return fail(W2A_ERR_HIP, "reset: launch failed: %s", hipGetErrorString(e));
  }
  end_call(env, (hipStream_t)stream);
  return W2A_OK;
}

int w2a_reset(w2a_env *env, const int32_t *county_w, const int32_t *year_i, const int32_t *coef_col,
              const int32_t *sample, const int32_t *budget, const uint8_t *mask, float *obs, void *stream) {
  if (!env || !county_w || !year_i || !coef_col || !sample) return fail(W2A_ERR_ARG, "w2a_reset: NULL argument");
  ResetArgs a;
  memset(&a, 0, sizeof(a));
  a.county_w = county_w; a.year_i = year_i; a.coef_col = coef_col; a.sample = sample; a.budget = budget;
  a.mask = mask; a.obs = obs; a.from_tuples = 1;
  return launch_reset(env, a, stream);
}

static int fill_cfg(const w2a_env *env, ResetCfg &rc, uint64_t seed, int32_t location, int augment, int32_t budget_kw,
                    int mode, int sticky) {
  if (location >= env->tb.S) return fail(W2A_ERR_ARG, "reset: location index outside fips_list");
  if (mode < W2A_BUDGET_FIXED || mode > W2A_BUDGET_CENTERED) return fail(W2A_ERR_ARG, "reset: bad sample_budget_mode");
  rc.seed = seed; rc.location = location; rc.augment = augment ? 1 : 0; rc.budget_kw = budget_kw;
  rc.sample_mode = mode; rc.sticky = sticky ? 1 : 0;
  return W2A_OK;
}

int w2a_reset_device_rng(w2a_env *env, uint64_t seed, int32_t location, int augment, int32_t budget_kw,
                         int sample_budget_mode, int sticky, int restart_episodes, const uint8_t *mask, float *obs,
                         void *stream) {
  if (!env) return fail(W2A_ERR_ARG, "w2a_reset_device_rng: NULL handle");
  ResetArgs a;
  memset(&a, 0, sizeof(a));
  int rc = fill_cfg(env, a.rc, seed, location, augment, budget_kw, sample_budget_mode, sticky);
  if (rc) return rc;
  a.mask = mask; a.obs = obs; a.from_tuples = 0; a.restart = restart_episodes ? 1 : 0;
  return launch_reset(env, a, stream);
}

int w2a_set_autoreset(w2a_env *env, uint64_t seed, int32_t location, int augment, int32_t budget_kw,
                      int sample_budget_mode, int sticky) {
  if (!env) return fail(W2A_ERR_ARG, "w2a_set_autoreset: NULL handle");
  int rc = fill_cfg(env, env->autoreset, seed, location, augment, budget_kw, sample_budget_mode, sticky);
  if (rc) return rc;
  env->has_autoreset = 1;
  return W2A_OK;
}

#include "w2a_step_dispatch.hip.h"

static size_t cub_sort_bytes(int64_t n) {
  size_t b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
  return b;
}

// The 32-bit (key, index) radix sorts go to rocprim directly, on its onesweep algorithm with a configuration of their own
// (tools/exp_sort_configs.hip, profiles/r06/exp_sort_configs.log; 1 048 576 pairs, 17 key bits):
//   * merge-sort limit 0: the library's default sends batches of up to 1 048 576 items -- exactly BASELINE's 1 M envs -- to
//     a merge sort (block sort + 10 merge passes, 150 us whatever the bit range);
//   * 9 bits per pass instead of 8: the 17 bits of a coefficient row are two passes instead of three;
//   * 1024 threads x 8 items per block: 128 blocks per pass. Smaller blocks fill more CUs and are SLOWER (256 x 8: 104 us
//     against 53): a pass is bound by its chain of decoupled look-backs, one link per block, not by bandwidth.
// 53 us against the library's 105 (17 bits) / 138 (the 28 bits of round 6's first key, which had the feature row as its
// minor part). Stable, like every LSD radix sort.
using Sort32Config =
    rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                               rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 9,
                                                                   rocprim::block_radix_rank_algorithm::match>,
                               0>;
static hipError_t sort32_pairs(void *tmp, size_t &bytes, const uint32_t *k_in, uint32_t *k_out, const uint32_t *v_in,
                               uint32_t *v_out, size_t n, unsigned end_bit, hipStream_t s) {
  return rocprim::radix_sort_pairs<Sort32Config>(tmp, bytes, k_in, k_out, v_in, v_out, n, 0u, end_bit, s);
}
static size_t cub_sort32_bytes(int64_t n) {
  size_t b = 0;
  (void)sort32_pairs(nullptr, b, nullptr, nullptr, nullptr, nullptr, (size_t)n, 32u, nullptr);
  return b;
}
// what w2a_reset_device_rng_sorted lays out in the same workspace: keys in / out, indices in / out, {sticky, episode} pairs
static size_t sorted_reset_bytes(int64_t num_envs) {
  const size_t n = (size_t)num_envs;
  return 4 * align256(4 * n) + align256(8 * n) + align256(cub_sort32_bytes(num_envs));
}

size_t w2a_sort_workspace_bytes(int64_t num_envs) {
  if (num_envs <= 0 || num_envs > (1ll << 27)) return 0;
  size_t n = (size_t)num_envs;
  const size_t relabel = align256(8 * n) * 2 + align256(4 * n) * 2 + align256(16 * n) + align256(12 * n) * 2 +
                         align256(cub_sort_bytes(num_envs));
  const size_t fused = sorted_reset_bytes(num_envs);
  return relabel > fused ? relabel : fused;
}

int w2a_sort_episodes(w2a_env *env, void *workspace, size_t workspace_bytes, void *stream) {
  if (!env || !workspace) return fail(W2A_ERR_ARG, "w2a_sort_episodes: NULL argument");
  if (workspace_bytes < w2a_sort_workspace_bytes(env->n)) return fail(W2A_ERR_STATE, "w2a_sort_episodes: workspace too small");
  if ((uintptr_t)workspace & 255) return fail(W2A_ERR_STATE, "w2a_sort_episodes: workspace must be 256-B aligned");
  REFUSE_WHILE_CAPTURING("w2a_sort_episodes", stream);
  const size_t n = (size_t)env->n;
  char *p = (char *)workspace;
  uint64_t *k_in = (uint64_t *)p;  p += align256(8 * n);
  uint64_t *k_out = (uint64_t *)p; p += align256(8 * n);
  uint32_t *i_in = (uint32_t *)p;  p += align256(4 * n);
  uint32_t *i_out = (uint32_t *)p; p += align256(4 * n);
  uint4 *cold_t = (uint4 *)p;      p += align256(16 * n);
  u3 *hot_t = (u3 *)p;             p += align256(12 * n);
  u3 *stepc_t = (u3 *)p;           p += align256(12 * n);
  size_t cub_bytes = cub_sort_bytes(env->n);
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  {
    HipDev d{env, s};
    bk_sort(env->bk, d);  // a relabelling: the batch stays in lock step; grouping and tile lists go stale
  }
  hipLaunchKernelGGL(k_sort_keys, dim3(blocks), dim3(256), 0, s, env->st.cold, k_in, i_in, env->n);
  HIP_TRY(hipGetLastError());
  // the high word only -- env indices follow the coefficient row; envs of one row keep their order (stable): the same
  // permutation as w2a_reset_device_rng_sorted's. The bit range stops at the packed word's highest used bit and never at
  // 64: for up to 1 M items rocprim merge-sorts with a comparator whose mask is built as (T(1) << end_bit) - 1, which for
  // end_bit = 64 is a shift by the type's width -- it then compares the LOW word (seen on gfx950: the batch came back in
  // feature-row order)
  int col_bits = 1;
  while (col_bits < 31 - SAMPLE_BITS && ((uint32_t)(env->tb.S - 1) >> col_bits)) ++col_bits;
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(p, cub_bytes, k_in, k_out, i_in, i_out, (int)n, 32, 32 + SAMPLE_BITS + col_bits, s));
  StateArrays tmp;
  tmp.cold = cold_t; tmp.hot3 = hot_t; tmp.stepc = stepc_t;
  hipLaunchKernelGGL(k_permute_state, dim3(blocks), dim3(256), 0, s, env->st, i_out, tmp, env->n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(env->st.cold, cold_t, 16 * n, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(env->st.hot3, hot_t, 12 * n, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(env->st.stepc, stepc_t, 12 * n, hipMemcpyDeviceToDevice, s));
  end_call(env, s);
  return W2A_OK;
}

int w2a_reset_device_rng_sorted(w2a_env *env, uint64_t seed, int32_t location, int augment, int32_t budget_kw,
                                int sample_budget_mode, int sticky, int restart_episodes, float *obs, void *workspace,
                                size_t workspace_bytes, void *stream) {
  if (!env || !workspace) return fail(W2A_ERR_ARG, "w2a_reset_device_rng_sorted: NULL argument");
  if (workspace_bytes < w2a_sort_workspace_bytes(env->n)) return fail(W2A_ERR_STATE, "w2a_reset_device_rng_sorted: workspace too small");
  if ((uintptr_t)workspace & 255) return fail(W2A_ERR_STATE, "w2a_reset_device_rng_sorted: workspace must be 256-B aligned");
  if (obs && ((uintptr_t)obs & 15)) return fail(W2A_ERR_ARG, "w2a_reset_device_rng_sorted: obs must be 16-B aligned");
  ResetArgs a;
  memset(&a, 0, sizeof(a));
  const int rc = fill_cfg(env, a.rc, seed, location, augment, budget_kw, sample_budget_mode, sticky);
  if (rc) return rc;
  // the coefficient row as a 32-bit key (74 600 values on the reference's tables). More than 2^32 rows take the general
  // path (w2a_reset_device_rng + w2a_sort_episodes + w2a_observe): the caller is told with return value 1
  const uint64_t key_space = (uint64_t)env->tb.S * (uint64_t)env->tb.n_samples;
  if (key_space > 0xFFFFFFFFull) return 1;
  REFUSE_WHILE_CAPTURING("w2a_reset_device_rng_sorted", stream);
  int end_bit = 1;
  while (end_bit < 32 && (key_space - 1) >> end_bit) ++end_bit;
  const size_t n = (size_t)env->n;
  char *p = (char *)workspace;
  uint32_t *k_in = (uint32_t *)p;  p += align256(4 * n);
  uint32_t *k_out = (uint32_t *)p; p += align256(4 * n);
  uint32_t *i_in = (uint32_t *)p;  p += align256(4 * n);
  uint32_t *i_out = (uint32_t *)p; p += align256(4 * n);
  uint2 *zw = (uint2 *)p;          p += align256(8 * n);
  size_t cub_bytes = cub_sort32_bytes(env->n);
  hipStream_t s = (hipStream_t)stream;
  // pass 1 reads `cold` only (written by every reset path, never stale in either form of the step state)
  hipLaunchKernelGGL(k_reset_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, env->tb, a.rc, env->st, env->n, env->gid0,
                     restart_episodes ? 1 : 0, k_in, i_in, zw);
  HIP_TRY(hipGetLastError());
  HIP_TRY(sort32_pairs(p, cub_bytes, k_in, k_out, i_in, i_out, n, (unsigned)end_bit, s));  // stable
  const W2aBook before = env->bk;
  {
    HipDev d{env, s};
    bk_reset_sorted(env->bk, d);
  }
  a.obs = obs; a.from_tuples = 3; a.src_idx = i_out; a.src_zw = zw;
  a.tb = env->tb; a.slot_obs = env->slot_obs; a.st = env->st;
  a.status = env->status; a.n = env->n; a.gid0 = env->gid0;
  hipLaunchKernelGGL(k_reset, dim3(grid_for(env->n)), dim3(BLOCK), 0, s, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    bk_reset_rollback(env->bk, before, false, false);
    end_call(env, s);
    return fail(W2A_ERR_HIP, "w2a_reset_device_rng_sorted: launch failed: %s", hipGetErrorString(e));
  }
  end_call(env, s);
  return W2A_OK;
}

static size_t cub_group_bytes(int64_t n) { return cub_sort32_bytes(n); }  // the same onesweep sort (10 key bits: 2 passes of 9 + 1)

static size_t wd_bytes(int32_t S, int32_t n_samples) { return align256((size_t)S * n_samples * 2 * ROWF * sizeof(double)); }
// tiles of the posterior-mean kernel: at most one partial tile per column on top of n / PMV_THREADS full ones
static size_t max_tiles(int64_t n, int32_t S) { return (size_t)((n + PMV_THREADS - 1) / PMV_THREADS) + (size_t)S; }
static size_t tile_bytes(int64_t n, int32_t S) { return align256(16 * max_tiles(n, S)) + 2 * align256(4 * (size_t)S) + 256; }
// the int8 matrix-core kernel: its own tile list (PI8_ROWS positions per tile), digit planes and scales of W, column
// flags, slot maxima / scales
static size_t max_tiles_i8(int64_t n, int32_t S) { return (size_t)((n + PI8_ROWS - 1) / PI8_ROWS) + (size_t)S; }
static size_t i8_bytes(int64_t n, int32_t S, int32_t n_samples) {
  const size_t rows = (size_t)S * n_samples * 2;
  return align256(16 * max_tiles_i8(n, S)) + 256 + align256(rows * ROWF * 4) + align256(rows * 4) + align256(4 * (size_t)S) + 3 * 256;
}

size_t w2a_group_workspace_bytes(int64_t num_envs, int32_t S, int32_t n_samples) {
  if (num_envs <= 0 || num_envs > (1ll << 27) || S <= 0 || n_samples <= 0) return 0;
  return align256(4 * (size_t)num_envs) * 4 + align256(16 * (size_t)num_envs) + wd_bytes(S, n_samples) +
         tile_bytes(num_envs, S) + i8_bytes(num_envs, S, n_samples) + align256(cub_group_bytes(num_envs));
}

int w2a_group_by_column(w2a_env *env, void *workspace, size_t workspace_bytes, void *stream) {
  if (!env || !workspace) return fail(W2A_ERR_ARG, "w2a_group_by_column: NULL argument");
  if (workspace_bytes < w2a_group_workspace_bytes(env->n, env->tb.S, env->tb.n_samples)) return fail(W2A_ERR_STATE, "w2a_group_by_column: workspace too small");
  if ((uintptr_t)workspace & 255) return fail(W2A_ERR_STATE, "w2a_group_by_column: workspace must be 256-B aligned");
  REFUSE_WHILE_CAPTURING("w2a_group_by_column", stream);
  const size_t n = (size_t)env->n;
  char *p = (char *)workspace;
  uint32_t *perm = (uint32_t *)p;  p += align256(4 * n);  // first two: stay in use after the call
  env->prep = (uint4 *)p;          p += align256(16 * n);
  double *wd = (double *)p;        p += wd_bytes(env->tb.S, env->tb.n_samples);
  uint4 *tiles = (uint4 *)p;       p += align256(16 * max_tiles(env->n, env->tb.S));
  uint32_t *col_start = (uint32_t *)p; p += align256(4 * (size_t)env->tb.S);
  uint32_t *col_end = (uint32_t *)p;   p += align256(4 * (size_t)env->tb.S);
  uint32_t *n_tiles = (uint32_t *)p;   p += 256;
  const size_t w_rows = (size_t)env->tb.S * env->tb.n_samples * 2;
  uint4 *tiles_i8 = (uint4 *)p;        p += align256(16 * max_tiles_i8(env->n, env->tb.S));
  uint32_t *n_tiles_i8 = (uint32_t *)p; p += 256;
  uint32_t *wq = (uint32_t *)p;        p += align256(w_rows * ROWF * 4);
  float *wscale = (float *)p;          p += align256(w_rows * 4);
  uint32_t *colflag = (uint32_t *)p;   p += align256(4 * (size_t)env->tb.S);
  uint32_t *xmax_bits = (uint32_t *)p; p += 256;  // [32] slot maxima (float bits), scanned once per table
  uint32_t *bmax = (uint32_t *)p;      p += 256;
  float *xs = (float *)p;              p += 256;  // [64] slot scales
  uint32_t *k_in = (uint32_t *)p;  p += align256(4 * n);  // sort keys, then the inverse permutation (stays in use)
  uint32_t *k_out = (uint32_t *)p; p += align256(4 * n);
  uint32_t *i_in = (uint32_t *)p;  p += align256(4 * n);
  size_t cub_bytes = cub_group_bytes(env->n);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_group_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, env->st.stepc, k_in, i_in, env->n);
  HIP_TRY(hipGetLastError());
  int bits = 1;
  while ((1 << bits) < env->tb.S) ++bits;
  HIP_TRY(sort32_pairs(p, cub_bytes, k_in, k_out, i_in, perm, n, (unsigned)bits, s));
  // tiles of <= PMV_THREADS sorted positions of one column each
  HIP_TRY(hipMemsetAsync(col_start, 0, 2 * align256(4 * (size_t)env->tb.S), s));
  hipLaunchKernelGGL(k_tile_bounds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, k_out, col_start, col_end, env->n);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_tile_list, dim3(1), dim3(1024), 0, s, col_start, col_end, env->tb.S, tiles, n_tiles, (uint32_t)PMV_THREADS);
  hipLaunchKernelGGL(k_tile_list, dim3(1), dim3(1024), 0, s, col_start, col_end, env->tb.S, tiles_i8, n_tiles_i8, (uint32_t)PI8_ROWS);
  HIP_TRY(hipGetLastError());
  env->tiles = tiles;
  env->n_tiles = n_tiles;
  // fixed-point operands of the int8 matrix-core kernel: slot maxima of the table (once per table and workspace), the
  // episode's largest budget, slot scales, digit planes + epilogue scale of every coefficient row, column flags
  if (env->xmax_ws != workspace) {
    HIP_TRY(hipMemsetAsync(xmax_bits, 0, 256, s));
    hipLaunchKernelGGL(k_pi8_slot_max, dim3(2048), dim3(256), 0, s, env->tb.X,
                       (int64_t)env->tb.T * env->tb.S_w * env->tb.Y * (ROWF / 4), xmax_bits);
    HIP_TRY(hipGetLastError());
    env->xmax_ws = workspace;
  }
  HIP_TRY(hipMemsetAsync(bmax, 0, 4, s));
  HIP_TRY(hipMemsetAsync(colflag, 0, 4 * (size_t)env->tb.S, s));
  hipLaunchKernelGGL(k_pi8_budget_max, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, env->st.stepc, env->n, bmax);
  hipLaunchKernelGGL(k_pi8_scales, dim3(1), dim3(64), 0, s, xmax_bits, bmax, env->tb.T, xs);
  hipLaunchKernelGGL(k_pi8_wq, dim3((unsigned)((w_rows + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const float *>(env->tb.W), xs, (int64_t)w_rows, env->tb.n_samples, wq, wscale, colflag);
  HIP_TRY(hipGetLastError());
  env->tiles_i8 = tiles_i8; env->n_tiles_i8 = n_tiles_i8; env->wq = wq; env->wscale = wscale; env->colflag = colflag;
  env->xs = xs;
  // fp64 copy of the coefficient rows, scaled by -log2(e), for the lane = env form of the reward kernel
  const int64_t w_count = (int64_t)env->tb.S * env->tb.n_samples * 2 * ROWF;
  hipLaunchKernelGGL(k_pm_wd, dim3((unsigned)((w_count + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const float *>(env->tb.W), wd, w_count);
  HIP_TRY(hipGetLastError());
  env->wd = wd;
  // sorted position of every env, kept in the first key buffer
  hipLaunchKernelGGL(k_group_inverse, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, perm, k_in, env->n);
  HIP_TRY(hipGetLastError());
  env->inv = k_in;
  env->perm = perm;
  bk_grouped(env->bk);
  return W2A_OK;
}

int w2a_posterior_mean_reward(w2a_env *env, const void *actions, int action_dtype, float *reward, void *stream) {
  if (!env || !actions || !reward) return fail(W2A_ERR_ARG, "w2a_posterior_mean_reward: NULL argument");
  if (action_dtype < W2A_ACT_I32 || action_dtype > W2A_ACT_U8) return fail(W2A_ERR_ARG, "w2a_posterior_mean_reward: bad action_dtype");
  if (!env->bk.perm_valid)
    return fail(W2A_ERR_STATE, "w2a_posterior_mean_reward: call w2a_group_by_column after every reset (the grouping of "
                               "envs by coefficient column is stale)");
  if (env->tb.fixes) return fail(W2A_ERR_ARG, "w2a_posterior_mean_reward: not available with corrected-semantics flags");
  PosteriorArgs a;
  memset(&a, 0, sizeof(a));
  a.tb = env->tb; a.st = env->st; a.inv = env->inv; a.perm = env->perm; a.prep = env->prep; a.actions = actions; a.act_dtype = action_dtype;
  a.reward = reward; a.status = env->status; a.n = env->n; a.wd = env->wd; a.tiles = env->tiles; a.n_tiles = env->n_tiles;
  if (!ensure_canonical(env, (hipStream_t)stream, "w2a_posterior_mean_reward")) return W2A_ERR_STATE;
  hipLaunchKernelGGL(k_pm_prep, dim3((unsigned)((env->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  if (env->pm_kernel == W2A_PM_MATRIX_I8) {
    PmI8Args b;
    b.p = a; b.tiles = env->tiles_i8; b.n_tiles = env->n_tiles_i8; b.wq = env->wq; b.wscale = env->wscale;
    b.colflag = env->colflag; b.xs = env->xs;
    const unsigned grid = (unsigned)((max_tiles_i8(env->n, env->tb.S) + 7) / 8 * 8);
    hipLaunchKernelGGL(k_posterior_mean_i8, dim3(grid), dim3(PI8_THREADS), 0, (hipStream_t)stream, b);
  } else if (env->pm_kernel == W2A_PM_MATRIX_F64) {
    const unsigned grid = (unsigned)((env->n + PM_ROWS - 1) / PM_ROWS);
    if (env->w_tail_used) hipLaunchKernelGGL(k_posterior_mean<8>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_posterior_mean<7>, dim3(grid), dim3(BLOCK), 0, (hipStream_t)stream, a);
  } else {
    const unsigned grid = (unsigned)((max_tiles(env->n, env->tb.S) + 7) / 8 * 8);
    if (env->w_tail_used) hipLaunchKernelGGL(k_posterior_mean_v<8>, dim3(grid), dim3(PMV_THREADS), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_posterior_mean_v<7>, dim3(grid), dim3(PMV_THREADS), 0, (hipStream_t)stream, a);
  }
  HIP_TRY(hipGetLastError());
  end_call(env, (hipStream_t)stream);
  return W2A_OK;
}

int w2a_set_posterior_kernel(w2a_env *env, int kernel) {
  if (!env) return fail(W2A_ERR_ARG, "w2a_set_posterior_kernel: NULL handle");
  if (kernel != W2A_PM_VECTOR && kernel != W2A_PM_MATRIX_F64 && kernel != W2A_PM_MATRIX_I8)
    return fail(W2A_ERR_ARG, "w2a_set_posterior_kernel: kernel must be W2A_PM_VECTOR, W2A_PM_MATRIX_F64 or W2A_PM_MATRIX_I8");
  env->pm_kernel = kernel;
  return W2A_OK;
}

int w2a_observe(w2a_env *env, float *obs, void *stream) {
  if (!env || !obs) return fail(W2A_ERR_ARG, "w2a_observe: NULL argument");
  ResetArgs a;
  memset(&a, 0, sizeof(a));
  a.obs = obs; a.from_tuples = 2;
  return launch_reset(env, a, stream);
}

int w2a_set_semantics(w2a_env *env, uint32_t fixes) {
  if (!env) return fail(W2A_ERR_ARG, "w2a_set_semantics: NULL handle");
  if (fixes & ~(uint32_t)W2A_FIX_ALL) return fail(W2A_ERR_ARG, "w2a_set_semantics: unknown W2A_FIX_* bit");
  if ((fixes & W2A_FIX_AUGMENT) && (!env->tb.sim_ptr || !env->tb.sim_idx))
    return fail(W2A_ERR_ARG, "w2a_set_semantics: W2A_FIX_AUGMENT needs sim_ptr/sim_idx in the tables");
  if ((fixes & W2A_FIX_ALERTS_2WKS) && env->tb.slot_hist2w >= W2A_ROW_FLOATS)
    return fail(W2A_ERR_SCHEMA, "w2a_set_semantics: slot_alerts_2wks out of range");
  env->tb.fixes = fixes;
  return W2A_OK;
}

size_t w2a_rollout_order_workspace_bytes(int64_t num_envs, int64_t table_rows) {
  if (num_envs <= 0 || num_envs > (1ll << 27) || table_rows <= 0 || table_rows > 0x7FFFFFFFll) return 0;
  return 2 * align256(4 * (size_t)num_envs) + align256(4 * (size_t)table_rows) + 2 * align256(4 * ((size_t)table_rows + 1));
}

static int order_attach(w2a_env *env, void *workspace, size_t workspace_bytes, const char *who) {
  const int64_t rows = (int64_t)env->tb.S_w * env->tb.Y;
  if (workspace_bytes < w2a_rollout_order_workspace_bytes(env->n, rows)) return fail(W2A_ERR_STATE, "%s: workspace too small", who);
  if ((uintptr_t)workspace & 255) return fail(W2A_ERR_STATE, "%s: workspace must be 256-B aligned", who);
  if (env->order_ws == workspace) return W2A_OK;
  const size_t n = (size_t)env->n;
  char *p = (char *)workspace;
  uint32_t *order = (uint32_t *)p;         p += align256(4 * n);
  env->order_rank = (uint32_t *)p;         p += align256(4 * n);
  env->order_cnt = (uint32_t *)p;          p += align256(4 * (size_t)rows);
  env->order_start = (uint32_t *)p;        p += align256(4 * ((size_t)rows + 1));
  env->order_tile_start = (uint32_t *)p;
  env->order_ws = workspace;
  // the order of an earlier workspace stays a valid permutation; this one holds none yet
  const bool drops = env->order && env->order != order;
  if (drops) env->order = nullptr;
  bk_order_attach(env->bk, drops);
  return W2A_OK;
}

int w2a_rollout_order_attach(w2a_env *env, void *workspace, size_t workspace_bytes) {
  if (!env || !workspace) return fail(W2A_ERR_ARG, "w2a_rollout_order_attach: NULL argument");
  return order_attach(env, workspace, workspace_bytes, "w2a_rollout_order_attach");
}

int w2a_rollout_order(w2a_env *env, void *workspace, size_t workspace_bytes, void *stream) {
  if (!env || !workspace) return fail(W2A_ERR_ARG, "w2a_rollout_order: NULL argument");
  REFUSE_WHILE_CAPTURING("w2a_rollout_order", stream);
  const int rc = order_attach(env, workspace, workspace_bytes, "w2a_rollout_order");
  if (rc) return rc;
  const int64_t rows = (int64_t)env->tb.S_w * env->tb.Y;
  const size_t n = (size_t)env->n;
  uint32_t *order = (uint32_t *)workspace;  // stays in use after the call
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (!env->bk.hist_valid) {  // the last whole-batch reset did not leave the counts (or something re-drew episodes since)
    HIP_TRY(hipMemsetAsync(env->order_cnt, 0, 4 * (size_t)rows, s));
    hipLaunchKernelGGL(k_order_rank, dim3(blocks), dim3(256), 0, s, env->st.cold, env->order_cnt, env->order_rank, env->n);
  }
  hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(1024), 0, s, env->order_cnt, (int32_t)rows, env->order_start, env->order_tile_start);
  hipLaunchKernelGGL(k_order_place, dim3(blocks), dim3(256), 0, s, env->st.cold, env->order_start, env->order_rank, order, env->n);
  HIP_TRY(hipGetLastError());
  env->order = order;
  bk_order_set(env->bk);
  return W2A_OK;
}

// ---- matrix-core rollout: workspace, preparation -------------------------------------------------------------------
// subtiles of <= 16 envs of one feature row (at most one partial one per row); a wave of k_rollout_mfma takes four
static size_t rm_max_tiles(int64_t n, int64_t rows) { return (size_t)((n + 15) / 16) + (size_t)rows; }
size_t w2a_rollout_mfma_workspace_bytes(int64_t num_envs, int64_t table_rows, int32_t S, int32_t n_samples) {
  if (num_envs <= 0 || num_envs > (1ll << 27) || table_rows <= 0 || table_rows > 0x7FFFFFFFll || S <= 0 || n_samples <= 0) return 0;
  const size_t w_rows = (size_t)S * n_samples * 2;
  return align256(16 * rm_max_tiles(num_envs, table_rows)) + 256 + align256(w_rows * ROWF * 4) + 3 * 256;
}

int w2a_rollout_mfma_prepare(w2a_env *env, void *workspace, size_t workspace_bytes, void *stream) {
  if (!env || !workspace) return fail(W2A_ERR_ARG, "w2a_rollout_mfma_prepare: NULL argument");
  REFUSE_WHILE_CAPTURING("w2a_rollout_mfma_prepare", stream);
  const int64_t rows = (int64_t)env->tb.S_w * env->tb.Y;
  if (workspace_bytes < w2a_rollout_mfma_workspace_bytes(env->n, rows, env->tb.S, env->tb.n_samples))
    return fail(W2A_ERR_STATE, "w2a_rollout_mfma_prepare: workspace too small");
  if ((uintptr_t)workspace & 255) return fail(W2A_ERR_STATE, "w2a_rollout_mfma_prepare: workspace must be 256-B aligned");
  if (!env->order || !env->order_start)
    return fail(W2A_ERR_STATE, "w2a_rollout_mfma_prepare: call w2a_rollout_order for this episode first");
  const size_t w_rows = (size_t)env->tb.S * env->tb.n_samples * 2;
  char *p = (char *)workspace;
  uint4 *tiles = (uint4 *)p;            p += align256(16 * rm_max_tiles(env->n, rows));
  uint32_t *n_tiles = (uint32_t *)p;    p += 256;
  uint32_t *wq = (uint32_t *)p;         p += align256(w_rows * ROWF * 4);
  uint32_t *xmax_bits = (uint32_t *)p;  p += 256;
  uint32_t *bmax = (uint32_t *)p;       p += 256;
  float *xs = (float *)p;               p += 256;
  hipStream_t s = (hipStream_t)stream;
  if (env->rm_ws != workspace) {  // once per table and workspace: slot scales and the digit table of W
    HIP_TRY(hipMemsetAsync(xmax_bits, 0, 512, s));  // slot maxima and the (unused here) budget maximum
    hipLaunchKernelGGL(k_pi8_slot_max, dim3(2048), dim3(256), 0, s, env->tb.X,
                       (int64_t)env->tb.T * env->tb.S_w * env->tb.Y * (ROWF / 4), xmax_bits);
    hipLaunchKernelGGL(k_pi8_scales, dim3(1), dim3(64), 0, s, xmax_bits, bmax, env->tb.T, xs);
    hipLaunchKernelGGL(k_rm_wq, dim3((unsigned)((w_rows + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const float *>(env->tb.W), xs, (int64_t)w_rows, wq);
    HIP_TRY(hipGetLastError());
    env->rm_ws = workspace;
  }
  // tiles of <= 64 consecutive positions of the visiting order that share one feature row: one thread per row, from the
  // row starts / tile starts the order's scan left in its workspace
  hipLaunchKernelGGL(k_rm_tiles, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, env->order_start, env->order_tile_start,
                     (int32_t)rows, tiles, n_tiles);
  HIP_TRY(hipGetLastError());
  env->rm_tiles = tiles; env->rm_n_tiles = n_tiles; env->rm_wq = wq;
  env->rm_xs = xs;
  bk_rm_prepared(env->bk);
  return W2A_OK;
}

int w2a_rollout(w2a_env *env, const w2a_policy *policy, int32_t n_steps, float *ret_out, int32_t *alerts_out,
                int32_t *attempts_over_budget, uint32_t *alert_mask, uint32_t *attempt_mask, int32_t mask_words,
                float *last_return, float *ret_snapshot, void *stream) {
  if (!env || !policy) return fail(W2A_ERR_ARG, "w2a_rollout: NULL argument");
  if (n_steps <= 0) return fail(W2A_ERR_ARG, "w2a_rollout: n_steps must be positive");
  REFUSE_WHILE_CAPTURING("w2a_rollout", stream);
  if (policy->kind < W2A_POLICY_NEVER || policy->kind > W2A_POLICY_TABLE) return fail(W2A_ERR_ARG, "w2a_rollout: bad policy kind");
  if (policy->kind == W2A_POLICY_TABLE && (!policy->table || policy->table_R <= 0))
    return fail(W2A_ERR_ARG, "w2a_rollout: tabular policy needs table [T][table_R] and table_R > 0");
  if ((alert_mask || attempt_mask) && mask_words * 32 < env->tb.T)
    return fail(W2A_ERR_ARG, "w2a_rollout: alert_mask / attempt_mask need ceil(T/32) words per env");
  RolloutArgs a;
  memset(&a, 0, sizeof(a));
  a.tb = env->tb; a.st = env->st; a.status = env->status; a.n = env->n; a.gid0 = env->gid0;
  a.pol = *policy;
  a.pol_slot = 0;
  if (policy->kind == W2A_POLICY_THRESHOLD) {
    if (policy->obs_col < 0 || policy->obs_col >= env->tb.n_obs) return fail(W2A_ERR_ARG, "w2a_rollout: obs_col outside the observation");
    int slot = env->obs_slot_host[policy->obs_col];
    if (slot >= 24 && slot <= 27) return fail(W2A_ERR_ARG, "w2a_rollout: threshold policies read table-sourced columns only");
    a.pol_slot = slot;
  }
  a.n_steps = n_steps; a.ret_out = ret_out; a.alerts_out = alerts_out; a.attempts_over_budget = attempts_over_budget;
  a.alert_mask = alert_mask; a.attempt_mask = attempt_mask; a.mask_words = mask_words; a.last_return = last_return;
  a.ret_snapshot = ret_snapshot;
  a.order = env->order;
  hipStream_t s = (hipStream_t)stream;
  HipDev dv{env, s};
  // does the handle know every env to be on the same day? A batch in lock step stays in lock step: every env runs the
  // same n_steps days, or all of them reach their last day
  const bool lock = bk_rollout_begin(env->bk, dv, n_steps);
  const int rkernel = bk_rollout_kernel(env->bk, lock, env->tb.fixes != 0, W2A_ROLLOUT_MFMA != 0, W2A_ROLLOUT_WIDE != 0);
  if (alert_mask) HIP_TRY(hipMemsetAsync(alert_mask, 0, (size_t)env->n * mask_words * sizeof(uint32_t), s));
  if (attempt_mask) HIP_TRY(hipMemsetAsync(attempt_mask, 0, (size_t)env->n * mask_words * sizeof(uint32_t), s));
  if (rkernel == W2A_BK_ROLLOUT_MFMA) {
    // the table-sourced part of the logits on the int8 matrix cores (w2a_rollout_mfma.hip.h): needs the episode's
    // feature-row tile list (w2a_rollout_mfma_prepare) and a batch in lock step
    RmArgs ra;
    ra.r = a; ra.tiles = env->rm_tiles; ra.n_tiles = env->rm_n_tiles; ra.wq = env->rm_wq; ra.xs = env->rm_xs;
    const size_t wgs = ((rm_max_tiles(env->n, (int64_t)env->tb.S_w * env->tb.Y) + 3) / 4 + RM_WAVES - 1) / RM_WAVES;
    launch_rollout_mfma(policy->kind, alert_mask || attempt_mask || ret_snapshot, (unsigned)((wgs + 7) / 8 * 8), s, ra);
    HIP_TRY(hipGetLastError());
    end_call(env, s);
    return W2A_OK;
  }
  launch_rollout(policy->kind, alert_mask || attempt_mask || ret_snapshot, env->tb.fixes != 0, grid_for(env->n), s, a);
  HIP_TRY(hipGetLastError());
  end_call(env, s);
  return W2A_OK;
}

int w2a_rollout_posterior_mean(w2a_env *env, const w2a_policy *policy, int32_t n_steps, float *ret_out,
                               int32_t *alerts_out, int32_t *attempts_over_budget, uint32_t *alert_mask,
                               uint32_t *attempt_mask, int32_t mask_words, float *last_return, float *ret_snapshot,
                               void *stream) {
  if (!env || !policy) return fail(W2A_ERR_ARG, "w2a_rollout_posterior_mean: NULL argument");
  if (n_steps <= 0) return fail(W2A_ERR_ARG, "w2a_rollout_posterior_mean: n_steps must be positive");
  REFUSE_WHILE_CAPTURING("w2a_rollout_posterior_mean", stream);
  if (policy->kind < W2A_POLICY_NEVER || policy->kind > W2A_POLICY_TABLE) return fail(W2A_ERR_ARG, "w2a_rollout_posterior_mean: bad policy kind");
  if (policy->kind == W2A_POLICY_TABLE && (!policy->table || policy->table_R <= 0))
    return fail(W2A_ERR_ARG, "w2a_rollout_posterior_mean: tabular policy needs table [T][table_R] and table_R > 0");
  if ((alert_mask || attempt_mask) && mask_words * 32 < env->tb.T)
    return fail(W2A_ERR_ARG, "w2a_rollout_posterior_mean: alert_mask / attempt_mask need ceil(T/32) words per env");
  if (!env->bk.perm_valid)
    return fail(W2A_ERR_STATE, "w2a_rollout_posterior_mean: call w2a_group_by_column after every reset");
  if (env->tb.fixes) return fail(W2A_ERR_ARG, "w2a_rollout_posterior_mean: not available with corrected-semantics flags");
  // not applicable (more draws than one staging pass holds; vector form: coefficients on slots 28/30/31; the fp64 matrix
  // kernel has no one-launch form): the caller runs the per-day calls
  const bool i8 = env->pm_kernel == W2A_PM_MATRIX_I8;
  if (env->pm_kernel == W2A_PM_MATRIX_F64 || (i8 && env->tb.n_samples > PI8_NPAD) ||
      (!i8 && (env->tb.n_samples > W2A_PMV_NPAD || env->w_tail_used)))
    return 1;
  PmRolloutArgs pa;
  memset(&pa, 0, sizeof(pa));
  RolloutArgs &a = pa.r;
  a.tb = env->tb; a.st = env->st; a.status = env->status; a.n = env->n; a.gid0 = env->gid0;
  a.pol = *policy;
  if (policy->kind == W2A_POLICY_THRESHOLD) {
    if (policy->obs_col < 0 || policy->obs_col >= env->tb.n_obs) return fail(W2A_ERR_ARG, "w2a_rollout_posterior_mean: obs_col outside the observation");
    int slot = env->obs_slot_host[policy->obs_col];
    if (slot >= 24 && slot <= 27) return fail(W2A_ERR_ARG, "w2a_rollout_posterior_mean: threshold policies read table-sourced columns only");
    a.pol_slot = slot;
  }
  a.n_steps = n_steps; a.ret_out = ret_out; a.alerts_out = alerts_out; a.attempts_over_budget = attempts_over_budget;
  a.alert_mask = alert_mask; a.attempt_mask = attempt_mask; a.mask_words = mask_words; a.last_return = last_return;
  a.ret_snapshot = ret_snapshot;
  pa.perm = env->perm; pa.tiles = env->tiles; pa.n_tiles = env->n_tiles; pa.wd = env->wd;
  hipStream_t s = (hipStream_t)stream;
  {
    HipDev dv{env, s};
    (void)bk_rollout_begin(env->bk, dv, n_steps);
  }
  if (alert_mask) HIP_TRY(hipMemsetAsync(alert_mask, 0, (size_t)env->n * mask_words * sizeof(uint32_t), s));
  if (attempt_mask) HIP_TRY(hipMemsetAsync(attempt_mask, 0, (size_t)env->n * mask_words * sizeof(uint32_t), s));
  if (i8) {
    PmI8RolloutArgs ia;
    ia.r = a; ia.perm = env->perm; ia.tiles = env->tiles_i8; ia.n_tiles = env->n_tiles_i8; ia.wq = env->wq;
    ia.wscale = env->wscale; ia.colflag = env->colflag; ia.xs = env->xs;
    const unsigned grid8 = (unsigned)((max_tiles_i8(env->n, env->tb.S) + 7) / 8 * 8);
    hipLaunchKernelGGL(k_pm_rollout_i8, dim3(grid8), dim3(PI8_THREADS), 0, s, ia);
    HIP_TRY(hipGetLastError());
    end_call(env, s);
    return W2A_OK;
  }
  const unsigned grid = (unsigned)((max_tiles(env->n, env->tb.S) + 7) / 8 * 8);
  launch_pm_rollout(policy->kind, alert_mask || attempt_mask || ret_snapshot, grid, s, pa);
  HIP_TRY(hipGetLastError());
  end_call(env, s);
  return W2A_OK;
}

int w2a_policy_actions(w2a_env *env, const w2a_policy *policy, int32_t *actions, int32_t *alerts,
                       int32_t *attempts_over_budget, uint32_t *alert_mask, uint32_t *attempt_mask, int32_t mask_words,
                       void *stream) {
  if (!env || !policy || !actions) return fail(W2A_ERR_ARG, "w2a_policy_actions: NULL argument");
  if (policy->kind < W2A_POLICY_NEVER || policy->kind > W2A_POLICY_TABLE) return fail(W2A_ERR_ARG, "w2a_policy_actions: bad policy kind");
  if (policy->kind == W2A_POLICY_TABLE && (!policy->table || policy->table_R <= 0))
    return fail(W2A_ERR_ARG, "w2a_policy_actions: tabular policy needs table [T][table_R] and table_R > 0");
  if ((alert_mask || attempt_mask) && mask_words * 32 < env->tb.T)
    return fail(W2A_ERR_ARG, "w2a_policy_actions: alert_mask / attempt_mask need ceil(T/32) words per env");
  PolicyArgs a;
  memset(&a, 0, sizeof(a));
  a.tb = env->tb; a.st = env->st; a.n = env->n; a.gid0 = env->gid0;
  a.pol = *policy;
  if (policy->kind == W2A_POLICY_THRESHOLD) {
    if (policy->obs_col < 0 || policy->obs_col >= env->tb.n_obs) return fail(W2A_ERR_ARG, "w2a_policy_actions: obs_col outside the observation");
    int slot = env->obs_slot_host[policy->obs_col];
    if (slot >= 24 && slot <= 27) return fail(W2A_ERR_ARG, "w2a_policy_actions: threshold policies read table-sourced columns only");
    a.pol_slot = slot;
  }
  a.actions = actions; a.alerts = alerts; a.attempts_over_budget = attempts_over_budget;
  a.alert_mask = alert_mask; a.attempt_mask = attempt_mask; a.mask_words = mask_words;
  if (!ensure_canonical(env, (hipStream_t)stream, "w2a_policy_actions")) return W2A_ERR_STATE;
  hipLaunchKernelGGL(k_policy_actions, dim3((unsigned)((env->n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  end_call(env, (hipStream_t)stream);
  return W2A_OK;
}

int w2a_get_state(w2a_env *env, const w2a_state_view *view, void *stream) {
  if (!env || !view) return fail(W2A_ERR_ARG, "w2a_get_state: NULL argument");
  int64_t blocks = (env->n + 255) / 256;
  if (!ensure_canonical(env, (hipStream_t)stream, "w2a_get_state")) return W2A_ERR_STATE;
  hipLaunchKernelGGL(k_get_state, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, env->st, env->n, env->tb.Y, env->tb.n_samples, *view);
  HIP_TRY(hipGetLastError());
  end_call(env, (hipStream_t)stream);
  return W2A_OK;
}

int w2a_query(w2a_env *env, int what) {
  if (!env) return fail(W2A_ERR_ARG, "w2a_query: NULL handle");
  switch (what) {
    case W2A_Q_LOCKSTEP_DAY: return env->bk.uni_t;
    case W2A_Q_LOCKSTEP: return env->bk.lock;
    case W2A_Q_PACKED_ELIGIBLE: return bk_packed_eligible(env->bk) ? 1 : 0;
    case W2A_Q_PACKED_CURRENT: return env->bk.pk_valid;
    case W2A_Q_CANONICAL_CURRENT: return env->bk.canon_valid;
    case W2A_Q_LAST_ROLLOUT_KERNEL: return env->bk.last_rollout_kernel;
    case W2A_Q_LAST_STEP_KERNEL: return env->bk.last_step_kernel;
    default: return fail(W2A_ERR_ARG, "w2a_query: unknown item");
  }
}

int w2a_invalidate(w2a_env *env, void *stream) {
  if (!env) return fail(W2A_ERR_ARG, "w2a_invalidate: NULL handle");
  hipStream_t s = (hipStream_t)stream;
  REFUSE_WHILE_CAPTURING("w2a_invalidate", s);
  // nothing the handle derived from the old buffer is trusted; the restored budgets need no looking at (the packed kernel
  // serves any budget: until round 5 this call scanned the buffer for its largest one and waited for the stream)
  bk_invalidate(env->bk);
  end_call(env, s);
  return W2A_OK;
}

int w2a_read_status(w2a_env *env, int32_t *status_out, void *stream) {
  if (!env || !status_out) return fail(W2A_ERR_ARG, "w2a_read_status: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(hipMemcpyAsync(status_out, env->status, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemsetAsync(env->status, 0, sizeof(int32_t), s));
  HIP_TRY(hipStreamSynchronize(s));
  return W2A_OK;
}

}  // extern "C"
