// w2a_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI of the vectorised HeatAlertEnv.
//
// What is computed follows the reference src/weather2alert/env.py:
//   reset  :133-184 (+ _get_episode :107-131)      -> draw_episode() / k_reset
//   _get_obs :186-195, _get_reward :197-226, step :238-262 -> k_step (and k_rollout: many days per launch)
// How it is computed is MI355X-first:
//   * an env is served by a 4-lane group of a 64-wide wavefront (16 envs per wave, 64 per
//     256-thread workgroup); lane l owns floats 8l..8l+7 of the env's 128-byte feature row
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
//   * workgroup -> env-tile mapping is XCD-aware (logical_block); the dense reward precompute
//     (k_logit_table) is the only MFMA user (fp64 16x16x4).
//
// No fallback path exists: without this library (or without a ROCm device) constructing an env raises.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "w2a.h"

#ifndef LANES
#define LANES 4  // lanes per env: 8, 4, 2 or 1 (A/B-tested on MI355X)
#endif
#define ROWF 32
#define QUADS (ROWF / 4 / LANES)  // float4 per lane per 32-float row
#ifndef BLOCK
#define BLOCK 256  // threads per workgroup (measured at 1 M envs: 64..256 within 2 %, 512 is 4 % slower)
#endif
#define ENVS_PER_BLOCK (BLOCK / LANES)
#define ENVS_PER_WAVE (64 / LANES)
#define HDR_BYTES 256
#ifndef W2A_NT_OBS
#define W2A_NT_OBS 1  // observation rows leave with non-temporal stores (they are not re-read by the env)
#endif
#ifndef W2A_NT_STATE
#define W2A_NT_STATE 0  // A/B: non-temporal loads/stores for the streamed per-env state, actions, reward, done
#endif
#ifndef W2A_NT_W
#define W2A_NT_W 0      // A/B: non-temporal loads for the gathered coefficient rows
#endif
#ifndef W2A_XCD_SWIZZLE
#define W2A_XCD_SWIZZLE 1  // consecutive env tiles on the same XCD (workgroups are dealt round-robin over 8 XCDs)
#endif
// Logical tile of a workgroup. With the swizzle, XCD k (blockIdx % 8 == k, observed placement; only speed
// depends on it) walks the k-th contiguous eighth of the env range, so neighbouring envs share an L2: partial
// output lines (reward, done) merge there, and with episode_order="sorted" each XCD touches one eighth of W / L.
__device__ __forceinline__ uint32_t logical_block(uint32_t b, uint32_t per_xcd) {
#if W2A_XCD_SWIZZLE
  return (b & 7u) * per_xcd + (b >> 3);
#else
  return b;
#endif
}
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_state(const uint4 *p) {
#if W2A_NT_STATE
  v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_state(uint4 *p, uint4 v) {
#if W2A_NT_STATE
  v4u w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<v4u *>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ float4 ld_w(const float4 *p) {
#if W2A_NT_W
  v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
#define RT_QUAD 6    // float4 index of the run-time slots 24..27
#define GATE_QUAD 7  // float4 index holding the gate copy (slot 30 = .z)

// ----------------------------------------------------------------------------------------
// packed state
// ----------------------------------------------------------------------------------------
// cold (uint4): x = ep_row (county_w*Y + year_i), y = coef_col << 12 | sample (n_samples <= 4096),
//               z = sticky budget (int, -1 unset), w = episode number
// hot  (uint4): x = dyn0: t[0:10) used[10:20) streak[20:30) last_actual[30] at_budget[31]
//               y = dyn1: hist14[0:14) n_days[16:26) finished[31]
//               z = episode return (f32 bits), w = budget (int)
#define D0_T(d) ((d) & 1023u)
#define D0_USED(d) (((d) >> 10) & 1023u)
#define D0_STREAK(d) (((d) >> 20) & 1023u)
#define D0_LAST(d) (((d) >> 30) & 1u)
#define D0_ATB(d) (((d) >> 31) & 1u)
#define D1_HIST(d) ((d) & 0x3FFFu)
#define D1_NDAYS(d) (((d) >> 16) & 1023u)
#define D1_FIN(d) (((d) >> 31) & 1u)

__device__ __forceinline__ uint32_t pack_d0(uint32_t t, uint32_t used, uint32_t streak, uint32_t last, uint32_t atb) {
  return (t & 1023u) | ((used > 1023u ? 1023u : used) << 10) | ((streak > 1023u ? 1023u : streak) << 20) |
         (last << 30) | (atb << 31);
}
__device__ __forceinline__ uint32_t pack_d1(uint32_t hist, uint32_t ndays, uint32_t fin) {
  return (hist & 0x3FFFu) | ((ndays & 1023u) << 16) | (fin << 31);
}

struct DevTables {
  const float4 *X;
  const int32_t *n_days;
  const int32_t *B0;
  const float4 *W;
  const int32_t *fips_to_weather;
  const int32_t *sim_cnt;
  const int32_t *weather_to_fips;  // [S_w] inverse of fips_to_weather (-1: county has no coefficients); nullable
  const double2 *L;                // [T][S_w*Y][n_samples] {baseline, gated effectiveness} exogenous logits; nullable
  const float4 *Wendo;             // [S*n_samples][2] run-time-slot coefficients (slots 24..27) per head; nullable
  const int32_t *sim_ptr;          // [S+1] CSR of similar(county) ∩ fips_list (only for W2A_FIX_AUGMENT); nullable
  const int32_t *sim_idx;
  int32_t T, S_w, Y, S, n_samples, n_obs;
  int32_t slot_hist2w;             // table slot of the historical 'alerts_2wks' column (-1: none)
  uint32_t fixes;                  // W2A_FIX_* bits: opt-in corrections of reference quirks (0 = faithful)
};
// overwrite component `idx` (0 .. 4*QUADS-1) of a lane's row fragment without dynamic register indexing
__device__ __forceinline__ void set_comp(float4 *x, int idx, float v) {
#pragma unroll
  for (int q = 0; q < QUADS; ++q) {
    if (idx == 4 * q) x[q].x = v;
    if (idx == 4 * q + 1) x[q].y = v;
    if (idx == 4 * q + 2) x[q].z = v;
    if (idx == 4 * q + 3) x[q].w = v;
  }
}
#define SAMPLE_BITS 12
#define PACK_W(coef_col, sample) (((uint32_t)(coef_col) << SAMPLE_BITS) | (uint32_t)(sample))
#define W_COL(y) ((y) >> SAMPLE_BITS)
#define W_SAMPLE(y) ((y) & ((1u << SAMPLE_BITS) - 1u))

struct ResetCfg {
  uint64_t seed;
  int32_t location;
  int32_t augment;
  int32_t budget_kw;
  int32_t sample_mode;
  int32_t sticky;
};

struct w2a_env {
  DevTables tb;
  int64_t n;
  int64_t gid0;
  const int32_t *slot_obs;  // [32] slot -> obs column (-1 none), in the state header
  uint4 *cold;
  uint4 *hot;
  int32_t *status;
  ResetCfg autoreset;
  int has_autoreset;
  int32_t obs_slot_host[ROWF];
};

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, const char *a = "") {
  snprintf(g_err, sizeof(g_err), fmt, a);
  return code;
}

// ----------------------------------------------------------------------------------------
// counter-based RNG (restated in oracle/heatalert_oracle.py: devrng_*)
// ----------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t w2a_mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__device__ __forceinline__ uint64_t rng_stream(uint64_t seed, uint64_t gid, uint64_t episode) {
  uint64_t h = w2a_mix64(seed + 0x9E3779B97F4A7C15ull * (gid + 1));
  return w2a_mix64(h ^ (episode * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull));
}
// uniform integer in [0, n): multiply-shift on the high 32 bits of the slot's word
__device__ __forceinline__ uint32_t rng_bounded(uint64_t stream, uint32_t slot, uint32_t n) {
  uint64_t u = w2a_mix64(stream + (uint64_t)(slot + 1) * 0x9E3779B97F4A7C15ull) >> 32;
  return (uint32_t)((u * (uint64_t)n) >> 32);
}

// ----------------------------------------------------------------------------------------
// cross-lane helpers (8-lane groups inside a DPP row of 16)
// ----------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// all-reduce (sum) over the 8 lanes of a group; every lane ends with the total
__device__ __forceinline__ double group_sum(double v) {
  if (LANES >= 2) v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]  : lane ^ 1
  if (LANES >= 4) v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]  : lane ^ 2
  if (LANES >= 8) v += dpp_f64<0x141>(v);  // row_half_mirror      : lane -> 7 - lane (other quad of the group)
  return v;
}

__device__ __forceinline__ float sigmoid_f32(float z) {
  // 1/(1+exp(-z)); exp(-z) overflows to +inf for z << 0 which gives exactly 0, and z = -inf
  // (closed effectiveness gate) also gives exactly 0.
  float e = __expf(-z);                      // v_mul + v_exp_f32
  return __builtin_amdgcn_rcpf(1.0f + e);    // v_rcp_f32 (1 ulp); rcp(+inf) = 0
}

// ----------------------------------------------------------------------------------------
// episode draw shared by the reset kernel and the same-step autoreset (env.py:145-178)
// ----------------------------------------------------------------------------------------
struct Episode {
  uint32_t ep_row, ep_w, ndays;
  int32_t budget, sticky;
  uint32_t bad;
};

__device__ __forceinline__ Episode draw_episode(const DevTables &tb, const ResetCfg &rc, uint64_t gid,
                                                uint32_t episode_no, int32_t sticky_in) {
  Episode e;
  uint64_t st = rng_stream(rc.seed, gid, episode_no);
  uint32_t bad = 0;
  uint32_t county = rc.location < 0 ? rng_bounded(st, 0, (uint32_t)tb.S) : (uint32_t)rc.location;
  if (county >= (uint32_t)tb.S) { county = 0; bad = 1; }
  uint32_t coef_col = county;
  if (rc.augment) {
    int32_t ns = tb.sim_cnt[county];
    if (ns <= 0) { bad = 1; ns = 1; }
    coef_col = rng_bounded(st, 1, (uint32_t)ns);  // position inside the filtered list
    if ((tb.fixes & W2A_FIX_AUGMENT) && tb.sim_idx) {
      // corrected augmentation: the drawn similar county supplies both the weather and the coefficients
      county = (uint32_t)tb.sim_idx[tb.sim_ptr[county] + (int32_t)coef_col];
      coef_col = county;
    }
  }
  uint32_t year_i = rng_bounded(st, 2, (uint32_t)tb.Y);
  uint32_t sample = rng_bounded(st, 3, (uint32_t)tb.n_samples);
  int32_t cw = tb.fips_to_weather[county];
  if (cw < 0) { cw = 0; bad = 1; }
  e.ep_row = (uint32_t)cw * (uint32_t)tb.Y + year_i;
  e.ep_w = PACK_W(coef_col, sample);
  int32_t nd = tb.n_days[e.ep_row];
  if (nd <= 0) { bad = 1; nd = 1; }
  e.ndays = (uint32_t)nd;
  int32_t b = (rc.sticky && sticky_in >= 0) ? sticky_in : (rc.budget_kw < 0 ? tb.B0[e.ep_row] : rc.budget_kw);
  if (b < 0) b = 0;
  if (rc.sample_mode == W2A_BUDGET_LESS_THAN) {
    b = (int32_t)rng_bounded(st, 4, (uint32_t)b + 1u);
  } else if (rc.sample_mode == W2A_BUDGET_CENTERED) {
    // rng.integers(0.5*b, 1.5*b + 1): NumPy truncates the float bounds
    int32_t lo = (int32_t)(0.5 * (double)b), hi = (int32_t)(1.5 * (double)b + 1.0);
    b = lo + (int32_t)rng_bounded(st, 4, (uint32_t)(hi - lo));
  }
  e.budget = b;
  e.sticky = rc.sticky ? b : -1;  // self.budget keeps the (sampled) value (env.py:167-178, Q9)
  e.bad = bad;
  return e;
}

// ----------------------------------------------------------------------------------------
// observation tile: wave-level transpose through LDS, 16-B coalesced stores
// ----------------------------------------------------------------------------------------
// x        : this lane's QUADS float4 of the row (slots 4*(l*QUADS+q)..), run-time fields already patched
// so       : obs column of each of those slots (-1 = not part of the observation)
// write_me : this env's row must be written (false -> keep what is in memory)
__device__ __forceinline__ void store_obs_tile(float *__restrict__ obs, float *tile, int64_t wave_env0, int64_t n,
                                               int n_obs, int lane, int grp, const float4 *x, const int4 *so,
                                               bool write_me) {
  float *row = tile + grp * n_obs;
#pragma unroll
  for (int q = 0; q < QUADS; ++q) {
    if (so[q].x >= 0) row[so[q].x] = x[q].x;
    if (so[q].y >= 0) row[so[q].y] = x[q].y;
    if (so[q].z >= 0) row[so[q].z] = x[q].z;
    if (so[q].w >= 0) row[so[q].w] = x[q].w;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const bool full = (wave_env0 + ENVS_PER_WAVE <= n);
  const bool all_write = __all(write_me || (wave_env0 + grp >= n));
  float *dst = obs + wave_env0 * n_obs;
  if (full && all_write) {
    const int chunks = (ENVS_PER_WAVE * n_obs) >> 2;  // ENVS_PER_WAVE*n_obs floats is a multiple of 4
#pragma unroll
    for (int c0 = 0; c0 < (ENVS_PER_WAVE * ROWF) / 4; c0 += 64) {
      const int c = c0 + lane;
      if (c < chunks) {
        v4f v = reinterpret_cast<const v4f *>(tile)[c];
#if W2A_NT_OBS
        __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst) + c);
#else
        reinterpret_cast<v4f *>(dst)[c] = v;
#endif
      }
    }
  } else {
    // ragged tail or some env of the wave keeps its stale row: element-wise, masked
    const int total = ENVS_PER_WAVE * n_obs;
    const unsigned long long wm = __ballot(write_me);  // taken before the loop: every lane still active
    for (int i = lane; i < total; i += 64) {
      int g = i / n_obs;
      bool w = (wm >> (g * LANES)) & 1ull;
      if (w && wave_env0 + g < n) dst[i] = tile[i];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// ----------------------------------------------------------------------------------------
// step kernel
// ----------------------------------------------------------------------------------------
struct StepArgs {
  DevTables tb;
  const int32_t *slot_obs;
  uint4 *cold;
  uint4 *hot;
  const void *actions;
  float *obs;
  float *reward;
  uint8_t *done;
  float *last_return;
  int32_t *status;
  int64_t n;
  int64_t gid0;
  ResetCfg rc;
  int32_t act_dtype;
};

#ifndef W2A_MIN_WAVES
#define W2A_MIN_WAVES 7  // waves/SIMD the plain step variants are compiled for (<= 72 VGPRs): the kernel is
#endif                   // latency-bound and measured faster at full occupancy
// The in-kernel autoreset variants carry the episode draw and would spill at 64 VGPRs (measured 1.4x slower),
// so they keep the compiler's own allocation; lock-step batches use the plain variant + k_reset instead.
// FIXES: compiled-in support for the W2A_FIX_* corrections; the faithful variants carry none of that code.
__device__ __forceinline__ int32_t load_action(const StepArgs &a, uint32_t e) {
  if (a.act_dtype == W2A_ACT_I32) return reinterpret_cast<const int32_t *>(a.actions)[e];
  if (a.act_dtype == W2A_ACT_I64) return (int32_t) reinterpret_cast<const int64_t *>(a.actions)[e];
  return reinterpret_cast<const uint8_t *>(a.actions)[e];
}

// One tile = the 16 envs of a wave, one day: everything of env.py:238-262 after the per-env state and action
// have been loaded (the callers differ in how they schedule those first-hop loads).
template <bool AUTORESET, bool WRITE_OBS, bool TABLE, bool FIXES>
__device__ __forceinline__ void step_tile(const StepArgs &a, float *s_tile_wave, int64_t wave_env0, int lane, int l,
                                          int grp, bool valid, uint32_t e, const uint4 cold, const uint4 hot,
                                          int32_t act) {
  uint32_t st_bits = 0;
  if (act != 0 && act != 1) { st_bits |= W2A_ST_BAD_ACTION; act = 1; }

  const uint32_t t = D0_T(hot.x), used = D0_USED(hot.x), streak = D0_STREAK(hot.x);
  const uint32_t hist = D1_HIST(hot.y), ndays = D1_NDAYS(hot.y);
  const int32_t budget = (int32_t)hot.w;
  if (D1_FIN(hot.y)) st_bits |= W2A_ST_STEP_AFTER_DONE;

  // env.py:242-250  budget gate, history
  const uint32_t atb = ((int32_t)used == budget) ? 1u : 0u;
  const uint32_t actual = (act == 1 && atb) ? 0u : (uint32_t)act;
  const uint32_t used2 = used + actual;
  const uint32_t hist2 = ((hist << 1) | actual) & 0x3FFFu;

  // gathers: feature row of day t (pre-increment, Q6) and the env's coefficients. Offsets are 32-bit
  // (table sizes are validated in w2a_create) so the loads use the scalar-base + vgpr-offset form.
  const uint32_t rows_per_day = (uint32_t)(a.tb.S_w * a.tb.Y);
  const uint32_t day_row = t * rows_per_day + cold.x;
  const uint32_t wrow = W_COL(cold.y) * (uint32_t)a.tb.n_samples + W_SAMPLE(cold.y);
  float4 x[QUADS];
  int4 so[QUADS];
#pragma unroll
  for (int q = 0; q < QUADS; ++q) {
    x[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (WRITE_OBS || !TABLE) x[q] = a.tb.X[day_row * (ROWF / 4) + l * QUADS + q];
    if (WRITE_OBS) so[q] = reinterpret_cast<const int4 *>(a.slot_obs)[l * QUADS + q];
  }
  // env.py:190-193 run-time fields (slots 24..27)
  const uint32_t fx = FIXES ? a.tb.fixes : 0u;
  // alert_lag1: today's action for t>0 (Q3); W2A_FIX_LAG: yesterday's
  const float f_lag1 = (t > 0) ? (float)((fx & W2A_FIX_LAG) ? D0_LAST(hot.x) : actual) : 0.0f;
  const float f_streak = (float)streak;                  // streak before today's action (Q4)
  const float f_rem = (float)(budget - (int32_t)used2);  // remaining_budget
  const float f_a2w = (float)__popc(hist2);              // agent's 14-day count ('alert_2wks', Q1)
  if (l == RT_QUAD / QUADS) x[RT_QUAD % QUADS] = make_float4(f_lag1, f_streak, f_rem, f_a2w);
  // W2A_FIX_ALERTS_2WKS: the agent's count also replaces the historical 'alerts_2wks' column, so it feeds the reward
  if ((fx & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0 && l == a.tb.slot_hist2w / (4 * QUADS))
    set_comp(x, a.tb.slot_hist2w % (4 * QUADS), f_a2w);
  double zb, ze;
  if (TABLE) {
    // exogenous part of both logits (incl. bias and the heat_qi gate) was precomputed by k_logit_table;
    // add the four run-time terms. Every lane of the group computes the same value (broadcast loads).
#ifdef W2A_EXP_L_DAY0  // timing experiment only (wrong results): every day reads the day-0 slice, which stays cached
    const double2 lv = a.tb.L[(size_t)cold.x * (uint32_t)a.tb.n_samples + W_SAMPLE(cold.y)];
#else
    const double2 lv = a.tb.L[(size_t)day_row * (uint32_t)a.tb.n_samples + W_SAMPLE(cold.y)];
#endif
    const float4 qb = a.tb.Wendo[wrow * 2];
    const float4 qe = a.tb.Wendo[wrow * 2 + 1];
    zb = fma((double)f_lag1, (double)qb.x, lv.x);
    zb = fma((double)f_streak, (double)qb.y, zb);
    zb = fma((double)f_rem, (double)qb.z, zb);
    zb = fma((double)f_a2w, (double)qb.w, zb);
    ze = fma((double)f_lag1, (double)qe.x, lv.y);
    ze = fma((double)f_streak, (double)qe.y, ze);
    ze = fma((double)f_rem, (double)qe.z, ze);
    ze = fma((double)f_a2w, (double)qe.w, ze);
  } else {
    const float4 *wp = a.tb.W + wrow * (2 * ROWF / 4) + l * QUADS;
    float4 wb[QUADS], we[QUADS];
    // The effectiveness logit only enters the reward through eff * actual (env.py:221): without an alert today
    // its coefficient row is not fetched at all (most env-days: alerts are budget-limited) -- half the
    // coefficient traffic. The lanes of such envs are masked out of the load; the reward is bit-identical.
    const bool need_eff = actual != 0u;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
      wb[q] = ld_w(wp + q);
      we[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (need_eff) we[q] = ld_w(wp + ROWF / 4 + q);
    }
    // env.py:207-217: two 28-term dot products, fp64 accumulation
    zb = 0.0;
    ze = 0.0;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
      const double x0 = (double)x[q].x, x1 = (double)x[q].y, x2 = (double)x[q].z, x3 = (double)x[q].w;
      zb = fma(x0, (double)wb[q].x, zb);
      zb = fma(x1, (double)wb[q].y, zb);
      zb = fma(x2, (double)wb[q].z, zb);
      zb = fma(x3, (double)wb[q].w, zb);
      ze = fma(x0, (double)we[q].x, ze);
      ze = fma(x1, (double)we[q].y, ze);
      ze = fma(x2, (double)we[q].z, ze);
      ze = fma(x3, (double)we[q].w, ze);
    }
    // effectiveness gate heat_qi > 0.5 (env.py:218): slot 30 holds the 0/1 gate flag with a zero
    // coefficient; a closed gate drives the logit to -inf so that sigmoid() is exactly 0
    if (l == GATE_QUAD / QUADS && !(x[GATE_QUAD % QUADS].z > 0.5f)) ze = -__builtin_inf();
    zb = group_sum(zb);
    ze = group_sum(ze);
  }
  const float base = sigmoid_f32((float)zb);
  const float eff = sigmoid_f32((float)ze);
  // env.py:221
  float r = -(1000.0f / 152.0f) * base * (1.0f - eff * (float)actual);
  if ((fx & W2A_FIX_PENALTY) && act == 1 && atb) r = -1.0f;  // env.py:223-224 made live (dead in the reference, Q5)

  const bool done = (t + 1 >= ndays);  // env.py:256
  const uint32_t t2 = done ? t : t + 1;
  const uint32_t streak2 = done ? streak : (actual ? streak + 1 : 0);  // env.py:260
  const float ret = __uint_as_float(hot.z) + r;

  uint4 hot2 = make_uint4(pack_d0(t2, used2, streak2, actual, atb), pack_d1(hist2, ndays, done ? 1u : 0u),
                          __float_as_uint(ret), (uint32_t)budget);
  uint4 cold2 = cold;
  bool write_row = !done;
  if (WRITE_OBS && (fx & W2A_FIX_OBS)) {
    // corrected observation (Q6): the row of the day the next action applies to, with the state as updated
    // by today's action; on the terminal step the last row (not a stale copy)
    write_row = true;
    if (!done) {
#pragma unroll
      for (int q = 0; q < QUADS; ++q) x[q] = a.tb.X[(day_row + rows_per_day) * (ROWF / 4) + l * QUADS + q];
      if (l == RT_QUAD / QUADS) x[RT_QUAD % QUADS] = make_float4((float)actual, (float)streak2, f_rem, f_a2w);
      if ((fx & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0 && l == a.tb.slot_hist2w / (4 * QUADS))
        set_comp(x, a.tb.slot_hist2w % (4 * QUADS), f_a2w);
    }
  }
  if (AUTORESET) {
    if (done) {
      // same-step autoreset: draw the next episode, emit its first observation (env.py:162-181)
      Episode ep = draw_episode(a.tb, a.rc, (uint64_t)(a.gid0 + e), cold.w + 1, (int32_t)cold.z);
      if (ep.bad) st_bits |= W2A_ST_BAD_EPISODE;
      cold2 = make_uint4(ep.ep_row, ep.ep_w, (uint32_t)ep.sticky, cold.w + 1);
      hot2 = make_uint4(pack_d0(0, 0, 0, 0, 0), pack_d1(0, ep.ndays, 0), __float_as_uint(0.0f), (uint32_t)ep.budget);
      if (WRITE_OBS) {
#pragma unroll
        for (int q = 0; q < QUADS; ++q) x[q] = a.tb.X[ep.ep_row * (ROWF / 4) + l * QUADS + q];
        if (l == RT_QUAD / QUADS) x[RT_QUAD % QUADS] = make_float4(0.0f, 0.0f, (float)ep.budget, 0.0f);
        if ((fx & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0 && l == a.tb.slot_hist2w / (4 * QUADS))
          set_comp(x, a.tb.slot_hist2w % (4 * QUADS), 0.0f);
      }
      write_row = true;
    }
  }
  if (valid && l == 0) {
    st_state(a.hot + e, hot2);
#if W2A_NT_STATE
    __builtin_nontemporal_store(r, a.reward + e);
    __builtin_nontemporal_store((uint8_t)(done ? 1 : 0), a.done + e);
#else
    a.reward[e] = r;
    a.done[e] = done ? 1 : 0;
#endif
    if (done) {
      if (a.last_return) a.last_return[e] = ret;
      if (AUTORESET) a.cold[e] = cold2;
    }
    if (st_bits) atomicOr(a.status, (int)st_bits);
  }
  if (WRITE_OBS) {
    store_obs_tile(a.obs, s_tile_wave, wave_env0, a.n, a.tb.n_obs, lane, grp, x, so, write_row);
  }
}

template <bool AUTORESET, bool WRITE_OBS, bool TABLE, bool FIXES>
__global__ __launch_bounds__(BLOCK, (AUTORESET || FIXES) ? 1 : W2A_MIN_WAVES) void k_step(const StepArgs a) {
  __shared__ __attribute__((aligned(16))) float s_tile[BLOCK / 64][ENVS_PER_WAVE * ROWF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = lane & (LANES - 1);
  const int grp = lane / LANES;
  const uint32_t lb = logical_block(blockIdx.x, gridDim.x >> 3);  // grid is a multiple of 8 workgroups
  const int64_t wave_env0 = ((int64_t)lb * (BLOCK / 64) + wave) * ENVS_PER_WAVE;
  if (wave_env0 >= a.n) return;  // whole wave past the end (padding tiles); no barrier is used below
  const int64_t env = wave_env0 + grp;
  const bool valid = env < a.n;
  const uint32_t e = (uint32_t)(valid ? env : (a.n - 1));  // clamp: inactive groups shadow the last env, never store
  const uint4 cold = ld_state(a.cold + e);
  const uint4 hot = ld_state(a.hot + e);
  const int32_t act = load_action(a, e);
  step_tile<AUTORESET, WRITE_OBS, TABLE, FIXES>(a, s_tile[wave], wave_env0, lane, l, grp, valid, e, cold, hot, act);
}

// ----------------------------------------------------------------------------------------
// reset kernels (same 8-lane geometry so the observation tile code is shared)
// ----------------------------------------------------------------------------------------
struct ResetArgs {
  DevTables tb;
  const int32_t *slot_obs;
  uint4 *cold;
  uint4 *hot;
  const int32_t *county_w, *year_i, *coef_col, *sample, *budget;  // host-tuple mode
  const uint8_t *mask;
  float *obs;
  int32_t *status;
  int64_t n;
  int64_t gid0;
  ResetCfg rc;
  int32_t from_tuples;  // 0: device RNG draw, 1: caller's tuples, 2: observe only (state untouched)
};

__global__ __launch_bounds__(BLOCK) void k_reset(const ResetArgs a) {
  __shared__ __attribute__((aligned(16))) float s_tile[BLOCK / 64][ENVS_PER_WAVE * ROWF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = lane & (LANES - 1);
  const int grp = lane / LANES;
  const uint32_t lb = logical_block(blockIdx.x, gridDim.x >> 3);
  const int64_t wave_env0 = ((int64_t)lb * (BLOCK / 64) + wave) * ENVS_PER_WAVE;
  if (wave_env0 >= a.n) return;
  const int64_t env = wave_env0 + grp;
  const bool valid = env < a.n;
  const uint32_t e = (uint32_t)(valid ? env : (a.n - 1));
  const bool sel = a.mask ? (a.mask[e] != 0) : true;
  uint4 cold = a.cold[e];
  uint32_t bad = 0;
  Episode ep;
  if (a.from_tuples == 2) {
    // observe only (w2a_observe): first observation of an already reset env, state untouched
    const uint4 hot = a.hot[e];
    ep.ep_row = cold.x;
    ep.budget = (int32_t)hot.w;
    if (D0_T(hot.x) != 0) bad = 4;
  } else if (a.from_tuples) {
    int32_t cw = a.county_w[e], yi = a.year_i[e], cc = a.coef_col[e], sm = a.sample[e];
    if (cw < 0 || cw >= a.tb.S_w) { cw = 0; bad = 1; }
    if (yi < 0 || yi >= a.tb.Y) { yi = 0; bad = 1; }
    if (cc < 0 || cc >= a.tb.S) { cc = 0; bad = 1; }
    if (sm < 0 || sm >= a.tb.n_samples) { sm = 0; bad = 1; }
    ep.ep_row = (uint32_t)cw * (uint32_t)a.tb.Y + (uint32_t)yi;
    ep.ep_w = PACK_W(cc, sm);
    // the logit-table path needs coefficient column == the weather county's own column
    if (a.tb.weather_to_fips && a.tb.weather_to_fips[cw] != cc) bad |= 2;
    int32_t nd = a.tb.n_days[ep.ep_row];
    if (nd <= 0) { nd = 1; bad |= 1; }
    ep.ndays = (uint32_t)nd;
    ep.budget = a.budget ? a.budget[e] : a.tb.B0[ep.ep_row];
    ep.sticky = (int32_t)cold.z;
  } else {
    ep = draw_episode(a.tb, a.rc, (uint64_t)(a.gid0 + e), cold.w + 1, (int32_t)cold.z);
    bad = ep.bad;
  }
  float4 x[QUADS];
  int4 so[QUADS];
#pragma unroll
  for (int q = 0; q < QUADS; ++q) {
    x[q] = a.tb.X[ep.ep_row * (ROWF / 4) + l * QUADS + q];  // day 0
    so[q] = reinterpret_cast<const int4 *>(a.slot_obs)[l * QUADS + q];
  }
  if (l == RT_QUAD / QUADS) x[RT_QUAD % QUADS] = make_float4(0.0f, 0.0f, (float)ep.budget, 0.0f);
  if ((a.tb.fixes & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0 && l == a.tb.slot_hist2w / (4 * QUADS))
    set_comp(x, a.tb.slot_hist2w % (4 * QUADS), 0.0f);  // the agent's (empty) history replaces the column
  if (valid && sel && l == 0) {
    if (a.from_tuples != 2) {
      a.cold[e] = make_uint4(ep.ep_row, ep.ep_w, (uint32_t)ep.sticky, cold.w + 1);
      a.hot[e] = make_uint4(pack_d0(0, 0, 0, 0, 0), pack_d1(0, ep.ndays, 0), __float_as_uint(0.0f), (uint32_t)ep.budget);
    }
    if (bad & 1) atomicOr(a.status, (int)W2A_ST_BAD_EPISODE);
    if (bad & 2) atomicOr(a.status, (int)W2A_ST_TABLE_MISMATCH);
    if (bad & 4) atomicOr(a.status, (int)W2A_ST_STEP_AFTER_DONE);
  }
  if (a.obs) store_obs_tile(a.obs, s_tile[wave], wave_env0, a.n, a.tb.n_obs, lane, grp, x, so, sel);
}

__global__ void k_init_state(uint4 *cold, uint4 *hot, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    cold[i] = make_uint4(0u, 0u, 0xFFFFFFFFu, 0xFFFFFFFFu);  // sticky = -1, episode_no = -1 (first reset -> 0)
    hot[i] = make_uint4(0u, pack_d1(0, 1, 1), 0u, 0u);
  }
}

__global__ void k_get_state(const uint4 *cold, const uint4 *hot, int64_t n, int32_t Y, int32_t n_samples,
                            w2a_state_view v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 c = cold[i], h = hot[i];
  if (v.t) v.t[i] = (int32_t)D0_T(h.x);
  if (v.used) v.used[i] = (int32_t)D0_USED(h.x);
  if (v.streak) v.streak[i] = (int32_t)D0_STREAK(h.x);
  if (v.last_actual) v.last_actual[i] = (int32_t)D0_LAST(h.x);
  if (v.at_budget) v.at_budget[i] = (int32_t)D0_ATB(h.x);
  if (v.hist14) v.hist14[i] = (int32_t)D1_HIST(h.y);
  if (v.n_days) v.n_days[i] = (int32_t)D1_NDAYS(h.y);
  if (v.budget) v.budget[i] = (int32_t)h.w;
  if (v.episode_return) v.episode_return[i] = __uint_as_float(h.z);
  if (v.county_w) v.county_w[i] = (int32_t)(c.x / (uint32_t)Y);
  if (v.year_i) v.year_i[i] = (int32_t)(c.x % (uint32_t)Y);
  if (v.coef_col) v.coef_col[i] = (int32_t)W_COL(c.y);
  if (v.sample) v.sample[i] = (int32_t)W_SAMPLE(c.y);
  if (v.sticky_budget) v.sticky_budget[i] = (int32_t)c.z;
  if (v.episode_no) v.episode_no[i] = (int32_t)c.w;
}


// ----------------------------------------------------------------------------------------
// logit-table precompute ("dense reward GEMM", BASELINE configs[3]/[4])
// ----------------------------------------------------------------------------------------
// For every weather county c with coefficient column cc = weather_to_fips[c]:
//     D_c [M = Y*T rows (t-major)] [N = 2*n_samples cols (2*s + head)]  =  A_c [M][K=28] * B_c [K][N]
// A_c = the county's feature rows (table slots 0..23 and 28..31: exogenous features, bias input,
// gate copy and pad, the last two with zero coefficients), B_c = its posterior coefficient rows.
// fp64 MFMA (v_mfma_f64_16x16x4_f64: A one f64 per lane A[l&15][l>>4], B[l>>4][l&15], D col = l&15,
// row = (l>>4) + 4*reg) keeps the 1e-5 reward bar: products of f32 inputs are exact in fp64.
// The heat_qi gate (env.py:218) is folded in: effectiveness logits of closed-gate rows are -inf.
// Output L[(t*R + c*Y + y)][s] = {baseline, effectiveness} (double2), the layout k_step<TABLE> gathers.
#define LT_K 28
#define LT_NT 13  // n-tiles (16 cols) staged per pass: 208 columns = 2*100 samples padded
typedef double double4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lt_slot(int k) { return k < 24 ? k : k + 4; }

struct LogitArgs {
  DevTables tb;
  double *L;
  int32_t msplit;
};

__global__ __launch_bounds__(BLOCK) void k_logit_table(const LogitArgs a) {
  __shared__ float sB[LT_K][LT_NT * 16];
  const int c = blockIdx.x;
  const int cc = a.tb.weather_to_fips[c];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Y = a.tb.Y, M = a.tb.Y * a.tb.T, N2 = 2 * a.tb.n_samples;
  const int R = a.tb.S_w * a.tb.Y;
  if (cc < 0) {
    // uniform per workgroup: a county without coefficients is never gathered; define its rows as zero
    // (the caller's buffer is not pre-cleared: a 2 GB memset in front of this kernel costs as much as it does)
    for (int m = blockIdx.y; m < M; m += a.msplit) {
      const int tz = m / Y, yz = m - tz * Y;
      double *row = a.L + ((size_t)tz * R + (size_t)c * Y + yz) * (size_t)N2;
      for (int n = tid; n < N2; n += BLOCK) row[n] = 0.0;
    }
    return;
  }
  const int mtiles = (M + 15) >> 4;
  const float *Xf = reinterpret_cast<const float *>(a.tb.X);
  const float *Wf = reinterpret_cast<const float *>(a.tb.W);
  const int q = lane >> 4, col = lane & 15;
  for (int n0 = 0; n0 < N2; n0 += LT_NT * 16) {
    __syncthreads();
    for (int idx = tid; idx < LT_K * LT_NT * 16; idx += BLOCK) {
      const int nn = idx / LT_K, k = idx - nn * LT_K;
      const int n = n0 + nn;
      float v = 0.0f;
      if (n < N2) v = Wf[((size_t)(cc * a.tb.n_samples + (n >> 1)) * 2 + (n & 1)) * ROWF + lt_slot(k)];
      sB[k][nn] = v;
    }
    __syncthreads();
    const int ntiles = min(LT_NT, (N2 - n0 + 15) >> 4);
    // m-tiles of this wave, software-pipelined: the next tile's A fragments and gate values are requested
    // BEFORE this tile's stores are issued. vmcnt retires in order and counts stores, so loads issued after
    // ~26 KB of stores would wait for all of them to drain (measured: 2.8 TB/s of writes instead of ~5).
    const int mstep = (BLOCK / 64) * a.msplit;
    float a_nx[LT_K / 4], g_nx[4];
    auto request = [&](int mt_) {
      const int m = min(mt_ * 16 + col, M - 1);
      const int tA = m / Y, yA = m - tA * Y;
      const float *xr = Xf + ((size_t)tA * R + (size_t)c * Y + yA) * ROWF;
#pragma unroll
      for (int ks = 0; ks < LT_K / 4; ++ks) a_nx[ks] = xr[lt_slot(4 * ks + q)];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mm = min(mt_ * 16 + q + 4 * j, M - 1);
        const int tj = mm / Y, yj = mm - tj * Y;
        g_nx[j] = Xf[((size_t)tj * R + (size_t)c * Y + yj) * ROWF + 30];
      }
    };
    int mt = blockIdx.y * (BLOCK / 64) + wave;
    if (mt < mtiles) request(mt);
    for (; mt < mtiles; mt += mstep) {
      double af[LT_K / 4];
#pragma unroll
      for (int ks = 0; ks < LT_K / 4; ++ks) af[ks] = (double)a_nx[ks];
      // this lane's 4 output rows: q + 4j
      size_t orow[4];
      bool ok[4], gate[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mj = mt * 16 + q + 4 * j;
        ok[j] = mj < M;
        const int mm = min(mj, M - 1);
        const int tj = mm / Y, yj = mm - tj * Y;
        orow[j] = (size_t)tj * R + (size_t)c * Y + yj;
        gate[j] = g_nx[j] > 0.5f;
      }
      if (mt + mstep < mtiles) request(mt + mstep);
      for (int nt = 0; nt < ntiles; ++nt) {
        double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < LT_K / 4; ++ks) {
          const double b = (double)sB[4 * ks + q][nt * 16 + col];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], b, acc, 0, 0, 0);
        }
        // epilogue: lanes (2k, 2k+1) hold adjacent columns of the same 4 rows; swap halves over DPP so that
        // each lane owns a 16-B {col 2k, col 2k+1} pair of two rows -> 2 x 16-B stores instead of 4 x 8-B
        const int n = n0 + nt * 16 + col;
        const bool odd = (lane & 1) != 0;
        double v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = acc[j];
          if ((n & 1) && !gate[j]) v[j] = -__builtin_inf();  // odd column = effectiveness head: closed gate
        }
        const double r0 = dpp_f64<0xB1>(odd ? v[0] : v[1]);
        const double r1 = dpp_f64<0xB1>(odd ? v[2] : v[3]);
        const int na = n & ~1;  // first column of the pair
        if (na < N2) {          // N2 is even, so the pair is in range together
          // even lane stores rows j = 0, 2; odd lane rows j = 1, 3 (selects, no dynamic register indexing)
          const double2 p0 = odd ? make_double2(r0, v[1]) : make_double2(v[0], r0);
          const double2 p1 = odd ? make_double2(r1, v[3]) : make_double2(v[2], r1);
          const size_t ra = odd ? orow[1] : orow[0], rb = odd ? orow[3] : orow[2];
          const bool oka = odd ? ok[1] : ok[0], okb = odd ? ok[3] : ok[2];
          if (oka) *reinterpret_cast<double2 *>(a.L + ra * (size_t)N2 + na) = p0;
          if (okb) *reinterpret_cast<double2 *>(a.L + rb * (size_t)N2 + na) = p1;
        }
      }
    }
  }
}

// Wendo[i] = {W[i][0][24..27], W[i][1][24..27]}: the run-time-slot coefficients, 32 B per (column, draw)
__global__ void k_pack_wendo(const float4 *W, float4 *Wendo, int64_t rows) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) {
    Wendo[2 * i] = W[i * (2 * ROWF / 4) + 6];
    Wendo[2 * i + 1] = W[i * (2 * ROWF / 4) + ROWF / 4 + 6];
  }
}

// ----------------------------------------------------------------------------------------
// on-device policy rollout: many days per launch, coefficients kept in registers
// ----------------------------------------------------------------------------------------
// Same per-day arithmetic as k_step (env.py:238-262) for up to n_steps days or until the episode ends; the
// action comes from a policy evaluated in the kernel on what the reference's agent would see: the lagging
// observation (row of day t-1, Q6), the remaining budget and the day. No observation rows are written; the
// packed state is advanced so step()/rollout() calls can be mixed.
struct RolloutArgs {
  DevTables tb;
  uint4 *cold;
  uint4 *hot;
  int32_t *status;
  int64_t n;
  int64_t gid0;
  w2a_policy pol;
  int32_t pol_slot;    // table slot of the observed feature (threshold policy)
  int32_t n_steps;
  float *ret_out;      // [n] sum of rewards over the days run by this call
  int32_t *alerts_out; // [n] alerts actually issued by this call
  int32_t *attempts_over_budget;  // [n] alerts attempted while at budget (nullable)
  uint32_t *alert_mask;           // [n][mask_words] bit d = alert issued on day d (nullable)
  int32_t mask_words;
  float *last_return;
};

__global__ __launch_bounds__(BLOCK) void k_rollout(const RolloutArgs a) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = lane & (LANES - 1);
  const int grp = lane / LANES;
  const uint32_t lb = logical_block(blockIdx.x, gridDim.x >> 3);
  const int64_t wave_env0 = ((int64_t)lb * (BLOCK / 64) + wave) * ENVS_PER_WAVE;
  if (wave_env0 >= a.n) return;
  const int64_t env = wave_env0 + grp;
  const bool valid = env < a.n;
  const uint32_t e = (uint32_t)(valid ? env : (a.n - 1));
  const uint4 cold = a.cold[e];
  const uint4 hot = a.hot[e];
  uint32_t t = D0_T(hot.x), used = D0_USED(hot.x), streak = D0_STREAK(hot.x), last = D0_LAST(hot.x);
  uint32_t atb = D0_ATB(hot.x), hist = D1_HIST(hot.y);
  const uint32_t ndays = D1_NDAYS(hot.y);
  const int32_t budget = (int32_t)hot.w;
  bool fin = D1_FIN(hot.y) != 0;
  float ret_total = __uint_as_float(hot.z);
  const uint32_t rows_per_day = (uint32_t)(a.tb.S_w * a.tb.Y);
  const uint32_t wrow = W_COL(cold.y) * (uint32_t)a.tb.n_samples + W_SAMPLE(cold.y);
  // coefficient rows once per launch, kept as fp64 in registers
  double wb[4 * QUADS], we[4 * QUADS];
  {
    const float4 *wp = a.tb.W + wrow * (2 * ROWF / 4) + l * QUADS;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
      const float4 b = wp[q], f = wp[ROWF / 4 + q];
      wb[4 * q] = b.x; wb[4 * q + 1] = b.y; wb[4 * q + 2] = b.z; wb[4 * q + 3] = b.w;
      we[4 * q] = f.x; we[4 * q + 1] = f.y; we[4 * q + 2] = f.z; we[4 * q + 3] = f.w;
    }
  }
  const float *Xf = reinterpret_cast<const float *>(a.tb.X);
  const uint64_t pstream = rng_stream(a.pol.seed ^ 0xA5A5A5A55A5A5A5Aull, (uint64_t)(a.gid0 + e), cold.w);
  float ret = 0.0f;
  int32_t alerts = 0, over = 0;
  uint32_t mask_word = 0, mask_idx = 0xFFFFFFFFu;
  // feature the policy sees on its first day here: row of day max(t-1, 0) (lagging observation, Q6)
  float feat = 0.0f;
  if (a.pol.kind == W2A_POLICY_THRESHOLD)
    feat = Xf[((size_t)((a.pol.obs_lag && t > 0 ? t - 1 : t) * rows_per_day + cold.x)) * ROWF + a.pol_slot];
  bool active = !fin && valid;
  for (int s = 0; s < a.n_steps; ++s) {
    if (!__any(active)) break;
    // ---- policy
    int32_t act = 0;
    const int32_t rem_now = budget - (int32_t)used;
    if (a.pol.kind == W2A_POLICY_ALWAYS) act = 1;
    else if (a.pol.kind == W2A_POLICY_BERNOULLI) {
      const uint32_t u = (uint32_t)(w2a_mix64(pstream + (uint64_t)(t + 1) * 0x9E3779B97F4A7C15ull) >> 32);
      act = ((float)u * 2.3283064365386963e-10f < a.pol.p) ? 1 : 0;
    } else if (a.pol.kind == W2A_POLICY_THRESHOLD) act = (feat > a.pol.threshold) ? 1 : 0;
    else if (a.pol.kind == W2A_POLICY_TABLE) {
      int32_t rr = rem_now < 0 ? 0 : (rem_now >= a.pol.table_R ? a.pol.table_R - 1 : rem_now);
      act = a.pol.table[(size_t)t * a.pol.table_R + rr] ? 1 : 0;
    }
    if (a.pol.require_budget && rem_now <= 0) act = 0;
    // ---- env.py:242-250
    const uint32_t atb_s = ((int32_t)used == budget) ? 1u : 0u;
    const uint32_t actual = (act == 1 && atb_s) ? 0u : (uint32_t)act;
    const uint32_t used2 = used + actual;
    const uint32_t hist2 = ((hist << 1) | actual) & 0x3FFFu;
    const uint32_t day_row = t * rows_per_day + cold.x;
    float4 x[QUADS];
#pragma unroll
    for (int q = 0; q < QUADS; ++q) x[q] = a.tb.X[day_row * (ROWF / 4) + l * QUADS + q];
    const float today = (a.pol.kind == W2A_POLICY_THRESHOLD) ? Xf[(size_t)day_row * ROWF + a.pol_slot] : 0.0f;
    const uint32_t fx = a.tb.fixes;
    const float f_a2w = (float)__popc(hist2);
    if (l == RT_QUAD / QUADS)
      x[RT_QUAD % QUADS] = make_float4((t > 0) ? (float)((fx & W2A_FIX_LAG) ? last : actual) : 0.0f, (float)streak,
                                       (float)(budget - (int32_t)used2), f_a2w);
    if ((fx & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0 && l == a.tb.slot_hist2w / (4 * QUADS))
      set_comp(x, a.tb.slot_hist2w % (4 * QUADS), f_a2w);
    double zb = 0.0, ze = 0.0;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
      const double x0 = (double)x[q].x, x1 = (double)x[q].y, x2 = (double)x[q].z, x3 = (double)x[q].w;
      zb = fma(x0, wb[4 * q], zb); zb = fma(x1, wb[4 * q + 1], zb);
      zb = fma(x2, wb[4 * q + 2], zb); zb = fma(x3, wb[4 * q + 3], zb);
      ze = fma(x0, we[4 * q], ze); ze = fma(x1, we[4 * q + 1], ze);
      ze = fma(x2, we[4 * q + 2], ze); ze = fma(x3, we[4 * q + 3], ze);
    }
    if (l == GATE_QUAD / QUADS && !(x[GATE_QUAD % QUADS].z > 0.5f)) ze = -__builtin_inf();
    zb = group_sum(zb);
    ze = group_sum(ze);
    float r = -(1000.0f / 152.0f) * sigmoid_f32((float)zb) * (1.0f - sigmoid_f32((float)ze) * (float)actual);
    if ((fx & W2A_FIX_PENALTY) && act == 1 && atb_s) r = -1.0f;
    if (active) {
      const bool done = (t + 1 >= ndays);
      ret += r;
      ret_total += r;
      alerts += (int32_t)actual;
      over += (act == 1 && atb_s) ? 1 : 0;
      if (a.alert_mask && actual) {
        const uint32_t wi = t >> 5;
        if (wi != mask_idx) {
          if (mask_idx != 0xFFFFFFFFu && l == 0 && mask_idx < (uint32_t)a.mask_words)
            a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
          mask_idx = wi;
          mask_word = 0;
        }
        mask_word |= 1u << (t & 31);
      }
      used = used2; hist = hist2; last = actual; atb = atb_s;
      if (!done) { streak = actual ? streak + 1 : 0; t = t + 1; }
      else { fin = true; active = false; }
      feat = a.pol.obs_lag ? today : feat;
    }
    // lag 1 (faithful): the next decision sees today's row; lag 0 needs tomorrow's row
    if (a.pol.kind == W2A_POLICY_THRESHOLD && !a.pol.obs_lag && active)
      feat = Xf[(size_t)(t * rows_per_day + cold.x) * ROWF + a.pol_slot];
  }
  if (valid && l == 0) {
    a.hot[e] = make_uint4(pack_d0(t, used, streak, last, atb), pack_d1(hist, ndays, fin ? 1u : 0u),
                          __float_as_uint(ret_total), (uint32_t)budget);
    if (a.ret_out) a.ret_out[e] = ret;
    if (a.alerts_out) a.alerts_out[e] = alerts;
    if (a.attempts_over_budget) a.attempts_over_budget[e] = over;
    if (a.alert_mask && mask_idx != 0xFFFFFFFFu && mask_idx < (uint32_t)a.mask_words)
      a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
    if (fin && a.last_return && !D1_FIN(hot.y)) a.last_return[e] = ret_total;
  }
}

// ----------------------------------------------------------------------------------------
// episode_order="sorted": relabel envs so that neighbours share coefficient / logit rows
// ----------------------------------------------------------------------------------------
__global__ void k_sort_keys(const uint4 *cold, uint64_t *keys, uint32_t *idx, int64_t n, int by_weather_row) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 c = cold[i];
  // gather path: coefficient row (column, draw) major, weather row minor; table path: weather row, then draw
  keys[i] = by_weather_row ? (((uint64_t)c.x << SAMPLE_BITS) | W_SAMPLE(c.y)) : (((uint64_t)c.y << 32) | c.x);
  idx[i] = (uint32_t)i;
}
__global__ void k_permute_state(const uint4 *cold, const uint4 *hot, const uint32_t *idx, uint4 *cold_o, uint4 *hot_o,
                                int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t j = idx[i];
  cold_o[i] = cold[j];
  hot_o[i] = hot[j];
}

// ----------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                                   \
  do {                                                                  \
    hipError_t _e = (expr);                                             \
    if (_e != hipSuccess) return fail(W2A_ERR_HIP, #expr ": %s", hipGetErrorString(_e)); \
  } while (0)

extern "C" {

int w2a_abi_version(void) { return W2A_ABI_VERSION; }
const char *w2a_last_error(void) { return g_err; }

size_t w2a_state_bytes(int64_t num_envs) {
  if (num_envs <= 0) return 0;
  size_t b = HDR_BYTES + (size_t)num_envs * 32;
  return (b + 255) & ~(size_t)255;
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
  if ((t->L != nullptr) != (t->Wendo != nullptr) || (t->L && !t->weather_to_fips))
    return fail(W2A_ERR_ARG, "w2a_create: L, Wendo and weather_to_fips must be given together");
  if (((uintptr_t)t->L & 15) || ((uintptr_t)t->Wendo & 15))
    return fail(W2A_ERR_STATE, "w2a_create: L and Wendo must be 16-B aligned");
  if (t->n_obs <= 0 || t->n_obs > W2A_ROW_FLOATS) return fail(W2A_ERR_SCHEMA, "w2a_create: n_obs must be in 1..32");
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
  h->tb.weather_to_fips = t->weather_to_fips;
  h->tb.sim_ptr = t->sim_ptr;
  h->tb.sim_idx = t->sim_idx;
  h->tb.slot_hist2w = t->slot_alerts_2wks;
  h->tb.fixes = 0;
  h->tb.L = reinterpret_cast<const double2 *>(t->L);
  h->tb.Wendo = reinterpret_cast<const float4 *>(t->Wendo);
  h->tb.T = t->T; h->tb.S_w = t->S_w; h->tb.Y = t->Y; h->tb.S = t->S; h->tb.n_samples = t->n_samples;
  h->tb.n_obs = t->n_obs;
  h->n = num_envs;
  h->gid0 = env_gid0;
  h->slot_obs = reinterpret_cast<const int32_t *>(state);
  h->cold = reinterpret_cast<uint4 *>((char *)state + HDR_BYTES);
  h->hot = h->cold + num_envs;
  h->status = status;
  h->has_autoreset = 0;
  for (int j = 0; j < ROWF; ++j) h->obs_slot_host[j] = j < t->n_obs ? t->obs_slot[j] : -1;
  hipError_t e1 = hipMemcpy(state, slot_obs, sizeof(slot_obs), hipMemcpyHostToDevice);
  hipError_t e2 = hipMemset(status, 0, sizeof(int32_t));
  if (e1 != hipSuccess || e2 != hipSuccess) { delete h; return fail(W2A_ERR_HIP, "w2a_create: header upload failed: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2)); }
  int64_t blocks = (num_envs + 255) / 256;
  hipLaunchKernelGGL(k_init_state, dim3((unsigned)blocks), dim3(256), 0, 0, h->cold, h->hot, num_envs);
  hipError_t e3 = hipDeviceSynchronize();
  if (e3 != hipSuccess) { delete h; return fail(W2A_ERR_HIP, "w2a_create: init kernel failed: %s", hipGetErrorString(e3)); }
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
  a.tb = env->tb; a.slot_obs = env->slot_obs; a.cold = env->cold; a.hot = env->hot;
  a.status = env->status; a.n = env->n; a.gid0 = env->gid0;
  if (a.obs && ((uintptr_t)a.obs & 15)) return fail(W2A_ERR_ARG, "reset: obs must be 16-B aligned");
  hipLaunchKernelGGL(k_reset, dim3(grid_for(env->n)), dim3(BLOCK), 0, (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
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
                         int sample_budget_mode, int sticky, const uint8_t *mask, float *obs, void *stream) {
  if (!env) return fail(W2A_ERR_ARG, "w2a_reset_device_rng: NULL handle");
  ResetArgs a;
  memset(&a, 0, sizeof(a));
  int rc = fill_cfg(env, a.rc, seed, location, augment, budget_kw, sample_budget_mode, sticky);
  if (rc) return rc;
  a.mask = mask; a.obs = obs; a.from_tuples = 0;
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

int w2a_step(w2a_env *env, const void *actions, int action_dtype, float *obs, float *reward, uint8_t *done,
             float *last_return, int flags, void *stream) {
  if (!env || !actions || !reward || !done) return fail(W2A_ERR_ARG, "w2a_step: NULL argument");
  if (action_dtype < W2A_ACT_I32 || action_dtype > W2A_ACT_U8) return fail(W2A_ERR_ARG, "w2a_step: bad action_dtype");
  const bool no_obs = (flags & W2A_STEP_NO_OBS) != 0;
  const bool autoreset = (flags & W2A_STEP_AUTORESET) != 0;
  const bool table = (flags & W2A_STEP_TABLE) != 0;
  if (table && !env->tb.L) return fail(W2A_ERR_ARG, "w2a_step: W2A_STEP_TABLE needs tables with L/Wendo (w2a_build_logit_table)");
  if (table && autoreset && env->autoreset.augment)
    return fail(W2A_ERR_ARG, "w2a_step: the logit-table path cannot serve similar_climate_counties episodes");
  if (!no_obs && !obs) return fail(W2A_ERR_ARG, "w2a_step: obs is NULL (pass W2A_STEP_NO_OBS for reward-only)");
  if (!no_obs && ((uintptr_t)obs & 15)) return fail(W2A_ERR_ARG, "w2a_step: obs must be 16-B aligned");
  if (autoreset && !env->has_autoreset) return fail(W2A_ERR_ARG, "w2a_step: W2A_STEP_AUTORESET needs w2a_set_autoreset first");
  StepArgs a;
  memset(&a, 0, sizeof(a));
  a.tb = env->tb; a.slot_obs = env->slot_obs; a.cold = env->cold; a.hot = env->hot;
  a.actions = actions; a.obs = obs; a.reward = reward; a.done = done; a.last_return = last_return;
  a.status = env->status; a.n = env->n; a.gid0 = env->gid0; a.rc = env->autoreset; a.act_dtype = action_dtype;
  dim3 grid(grid_for(env->n)), block(BLOCK);
  hipStream_t s = (hipStream_t)stream;
#define W2A_LAUNCH(AR, OB, TB) \
  do { if (env->tb.fixes) hipLaunchKernelGGL((k_step<AR, OB, TB, true>), grid, block, 0, s, a); \
       else hipLaunchKernelGGL((k_step<AR, OB, TB, false>), grid, block, 0, s, a); } while (0)
  const int variant = (autoreset ? 4 : 0) | (no_obs ? 0 : 2) | (table ? 1 : 0);
  switch (variant) {
    case 0: W2A_LAUNCH(false, false, false); break;
    case 1: W2A_LAUNCH(false, false, true); break;
    case 2: W2A_LAUNCH(false, true, false); break;
    case 3: W2A_LAUNCH(false, true, true); break;
    case 4: W2A_LAUNCH(true, false, false); break;
    case 5: W2A_LAUNCH(true, false, true); break;
    case 6: W2A_LAUNCH(true, true, false); break;
    default: W2A_LAUNCH(true, true, true); break;
  }
#undef W2A_LAUNCH
  HIP_TRY(hipGetLastError());
  return W2A_OK;
}


size_t w2a_logit_table_bytes(const w2a_tables *t) {
  if (!t || t->T <= 0 || t->S_w <= 0 || t->Y <= 0 || t->n_samples <= 0) return 0;
  return (size_t)t->T * t->S_w * t->Y * t->n_samples * sizeof(double2);
}

size_t w2a_wendo_bytes(const w2a_tables *t) {
  if (!t || t->S <= 0 || t->n_samples <= 0) return 0;
  return (size_t)t->S * t->n_samples * 2 * sizeof(float4);
}

int w2a_build_logit_table(const w2a_tables *t, void *L, size_t L_bytes, void *Wendo, size_t Wendo_bytes, void *stream) {
  if (!t || !L || !Wendo) return fail(W2A_ERR_ARG, "w2a_build_logit_table: NULL argument");
  if (!t->X || !t->W || !t->weather_to_fips) return fail(W2A_ERR_ARG, "w2a_build_logit_table: X, W and weather_to_fips are required");
  if (L_bytes < w2a_logit_table_bytes(t) || Wendo_bytes < w2a_wendo_bytes(t))
    return fail(W2A_ERR_STATE, "w2a_build_logit_table: output buffer too small");
  if (((uintptr_t)L & 15) || ((uintptr_t)Wendo & 15)) return fail(W2A_ERR_STATE, "w2a_build_logit_table: buffers must be 16-B aligned");
  if ((int64_t)t->T * t->S_w * t->Y > 0x7FFFFFFFll) return fail(W2A_ERR_SCHEMA, "w2a_build_logit_table: table too large");
  LogitArgs a;
  memset(&a, 0, sizeof(a));
  a.tb.X = reinterpret_cast<const float4 *>(t->X);
  a.tb.W = reinterpret_cast<const float4 *>(t->W);
  a.tb.weather_to_fips = t->weather_to_fips;
  a.tb.T = t->T; a.tb.S_w = t->S_w; a.tb.Y = t->Y; a.tb.S = t->S; a.tb.n_samples = t->n_samples;
  a.L = reinterpret_cast<double *>(L);
  a.msplit = 4;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_logit_table, dim3((unsigned)t->S_w, (unsigned)a.msplit), dim3(BLOCK), 0, s, a);
  HIP_TRY(hipGetLastError());
  const int64_t rows = (int64_t)t->S * t->n_samples;
  hipLaunchKernelGGL(k_pack_wendo, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const float4 *>(t->W), reinterpret_cast<float4 *>(Wendo), rows);
  HIP_TRY(hipGetLastError());
  return W2A_OK;
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t cub_sort_bytes(int64_t n) {
  size_t b = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                           (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
  return b;
}

size_t w2a_sort_workspace_bytes(int64_t num_envs) {
  if (num_envs <= 0 || num_envs > (1ll << 27)) return 0;
  size_t n = (size_t)num_envs;
  return align256(8 * n) * 2 + align256(4 * n) * 2 + align256(16 * n) * 2 + align256(cub_sort_bytes(num_envs));
}

int w2a_sort_episodes(w2a_env *env, int by_weather_row, void *workspace, size_t workspace_bytes, void *stream) {
  if (!env || !workspace) return fail(W2A_ERR_ARG, "w2a_sort_episodes: NULL argument");
  if (workspace_bytes < w2a_sort_workspace_bytes(env->n)) return fail(W2A_ERR_STATE, "w2a_sort_episodes: workspace too small");
  if ((uintptr_t)workspace & 255) return fail(W2A_ERR_STATE, "w2a_sort_episodes: workspace must be 256-B aligned");
  const size_t n = (size_t)env->n;
  char *p = (char *)workspace;
  uint64_t *k_in = (uint64_t *)p;  p += align256(8 * n);
  uint64_t *k_out = (uint64_t *)p; p += align256(8 * n);
  uint32_t *i_in = (uint32_t *)p;  p += align256(4 * n);
  uint32_t *i_out = (uint32_t *)p; p += align256(4 * n);
  uint4 *cold_t = (uint4 *)p;      p += align256(16 * n);
  uint4 *hot_t = (uint4 *)p;       p += align256(16 * n);
  size_t cub_bytes = cub_sort_bytes(env->n);
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_sort_keys, dim3(blocks), dim3(256), 0, s, env->cold, k_in, i_in, env->n, by_weather_row);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipcub::DeviceRadixSort::SortPairs(p, cub_bytes, k_in, k_out, i_in, i_out, (int)n, 0, 64, s));
  hipLaunchKernelGGL(k_permute_state, dim3(blocks), dim3(256), 0, s, env->cold, env->hot, i_out, cold_t, hot_t, env->n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(env->cold, cold_t, 16 * n, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipMemcpyAsync(env->hot, hot_t, 16 * n, hipMemcpyDeviceToDevice, s));
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
  if ((fixes & W2A_FIX_ALERTS_2WKS) && env->tb.L)
    return fail(W2A_ERR_ARG, "w2a_set_semantics: W2A_FIX_ALERTS_2WKS moves a coefficient out of the precomputed logit "
                             "table; create the handle without L/Wendo (row-gather path)");
  if ((fixes & W2A_FIX_ALERTS_2WKS) && env->tb.slot_hist2w >= W2A_ROW_FLOATS)
    return fail(W2A_ERR_SCHEMA, "w2a_set_semantics: slot_alerts_2wks out of range");
  env->tb.fixes = fixes;
  return W2A_OK;
}

int w2a_rollout(w2a_env *env, const w2a_policy *policy, int32_t n_steps, float *ret_out, int32_t *alerts_out,
                int32_t *attempts_over_budget, uint32_t *alert_mask, int32_t mask_words, float *last_return,
                void *stream) {
  if (!env || !policy) return fail(W2A_ERR_ARG, "w2a_rollout: NULL argument");
  if (n_steps <= 0) return fail(W2A_ERR_ARG, "w2a_rollout: n_steps must be positive");
  if (policy->kind < W2A_POLICY_NEVER || policy->kind > W2A_POLICY_TABLE) return fail(W2A_ERR_ARG, "w2a_rollout: bad policy kind");
  if (policy->kind == W2A_POLICY_TABLE && (!policy->table || policy->table_R <= 0))
    return fail(W2A_ERR_ARG, "w2a_rollout: tabular policy needs table [T][table_R] and table_R > 0");
  if (alert_mask && mask_words * 32 < env->tb.T) return fail(W2A_ERR_ARG, "w2a_rollout: alert_mask needs ceil(T/32) words per env");
  RolloutArgs a;
  memset(&a, 0, sizeof(a));
  a.tb = env->tb; a.cold = env->cold; a.hot = env->hot; a.status = env->status; a.n = env->n; a.gid0 = env->gid0;
  a.pol = *policy;
  a.pol_slot = 0;
  if (policy->kind == W2A_POLICY_THRESHOLD) {
    if (policy->obs_col < 0 || policy->obs_col >= env->tb.n_obs) return fail(W2A_ERR_ARG, "w2a_rollout: obs_col outside the observation");
    int slot = env->obs_slot_host[policy->obs_col];
    if (slot >= 24 && slot <= 27) return fail(W2A_ERR_ARG, "w2a_rollout: threshold policies read table-sourced columns only");
    a.pol_slot = slot;
  }
  a.n_steps = n_steps; a.ret_out = ret_out; a.alerts_out = alerts_out; a.attempts_over_budget = attempts_over_budget;
  a.alert_mask = alert_mask; a.mask_words = mask_words; a.last_return = last_return;
  hipStream_t s = (hipStream_t)stream;
  if (alert_mask) HIP_TRY(hipMemsetAsync(alert_mask, 0, (size_t)env->n * mask_words * sizeof(uint32_t), s));
  hipLaunchKernelGGL(k_rollout, dim3(grid_for(env->n)), dim3(BLOCK), 0, s, a);
  HIP_TRY(hipGetLastError());
  return W2A_OK;
}

int w2a_get_state(w2a_env *env, const w2a_state_view *view, void *stream) {
  if (!env || !view) return fail(W2A_ERR_ARG, "w2a_get_state: NULL argument");
  int64_t blocks = (env->n + 255) / 256;
  hipLaunchKernelGGL(k_get_state, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, env->cold, env->hot,
                     env->n, env->tb.Y, env->tb.n_samples, *view);
  HIP_TRY(hipGetLastError());
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
