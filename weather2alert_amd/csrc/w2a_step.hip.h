// w2a_step.hip.h -- k_step: one day for every env (env.py:238-262)
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_STEP_HIP_H
#define W2A_STEP_HIP_H

// ----------------------------------------------------------------------------------------
// step kernel
// ----------------------------------------------------------------------------------------
struct StepArgs {
  DevTables tb;
  const int32_t *slot_obs;
  StateArrays st;
  const void *actions;
  void *obs;  // f32 [n][n_obs]
  float *reward;
  uint8_t *done;
  float *last_return;
  int32_t *status;
  int64_t n;
  int64_t gid0;
  ResetCfg rc;
  int32_t act_dtype;
  int32_t uni_nd;         // k_step64 packed variant: the episode length shared by every env (a table constant; the day
                          // is read from the mirror's day word, StateArrays::pk_day)
  int32_t skip_finished;  // k_step64: envs whose episode is over are left untouched (reward 0, done 1, no status bit)
  int32_t next_step;      // AUTORESET variants: restart on the call after the terminal step (W2A_STEP_NEXT_STEP)
};

#ifndef W2A_MIN_WAVES
#define W2A_MIN_WAVES 7  // waves/SIMD the plain step variants are compiled for (<= 72 VGPRs): the kernel is
#endif                   // latency-bound and measured faster at full occupancy
// The in-kernel autoreset variants carry the episode draw and would spill at 64 VGPRs (measured 1.4x slower),
// so they keep the compiler's own allocation; lock-step batches use the plain variant + k_reset instead.
// FIXES: compiled-in support for the W2A_FIX_* corrections; the faithful variants carry none of that code.
__device__ __forceinline__ int32_t load_action_raw(const void *actions, int32_t dtype, uint32_t e) {
  if (dtype == W2A_ACT_I32) return reinterpret_cast<const int32_t *>(actions)[e];
  if (dtype == W2A_ACT_I64) return (int32_t) reinterpret_cast<const int64_t *>(actions)[e];
  return reinterpret_cast<const uint8_t *>(actions)[e];
}
__device__ __forceinline__ int32_t load_action(const StepArgs &a, uint32_t e) {
  return load_action_raw(a.actions, a.act_dtype, e);
}

// One tile = the 16 envs of a wave, one day: everything of env.py:238-262 after the per-env state and action
// have been loaded (the callers differ in how they schedule those first-hop loads).
template <bool AUTORESET, bool WRITE_OBS, bool FIXES>
__device__ __forceinline__ void step_tile(const StepArgs &a, float *s_tile_wave, int64_t wave_env0, int lane, int l,
                                          int grp, bool valid, uint32_t e, const uint4 cold, const uint4 hot,
                                          int32_t act) {
  uint32_t st_bits = 0;
  // W2A_STEP_NEXT_STEP: an env whose terminal step ran on the previous call restarts on this one; its action is ignored
  const bool restart_in = AUTORESET && a.next_step && D1_FIN(hot.y);
  if (restart_in) act = 0;
  if (act != 0 && act != 1) { st_bits |= W2A_ST_BAD_ACTION; act = 1; }

  const uint32_t t = D0_T(hot.x), used = D0_USED(hot.x), streak = D0_STREAK(hot.x);
  const uint32_t hist = D1_HIST(hot.y), ndays = D1_NDAYS(hot.y);
  const int32_t budget = (int32_t)hot.w;
  if (D1_FIN(hot.y) && !restart_in) st_bits |= W2A_ST_STEP_AFTER_DONE;

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
    x[q] = a.tb.X[day_row * (ROWF / 4) + l * QUADS + q];
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
  {
    const float4 *wp = a.tb.W + wrow * (2 * ROWF / 4) + l * QUADS;
    float4 wb[QUADS], we[QUADS];
    // The effectiveness logit only enters the reward through eff * actual (env.py:221): without an alert today
    // its coefficient row is not fetched (most env-days: alerts are budget-limited) -- half the coefficient
    // traffic. Branch-free: such groups re-request their baseline row (already in flight) and the resulting logit
    // is multiplied by actual = 0; a conditional load would be waited for at the end of its branch, before the
    // arithmetic of anybody else in the wave could start. The reward is bit-identical.
    const int eff_off = (actual != 0u) ? ROWF / 4 : 0;
#pragma unroll
    for (int q = 0; q < QUADS; ++q) {
      wb[q] = ld_w(wp + q);
      we[q] = ld_w(wp + eff_off + q);
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
  float r = reward_from_logits(zb, ze, actual);  // env.py:211-221
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
    if (a.next_step ? restart_in : done) {
      // same-step autoreset (or, W2A_STEP_NEXT_STEP, the call after the terminal step): draw the next episode, emit
      // its first observation (env.py:162-181)
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
    if (AUTORESET && (a.next_step ? restart_in : done)) store_episode(a.st, e, cold2, hot2);
    else store_hot(a.st, e, hot2);
    a.reward[e] = restart_in ? 0.0f : r;                 // the restarting call is no env step: reward 0, not done
    a.done[e] = (done && !restart_in) ? 1 : 0;
    if (done && !restart_in) {
      if (a.last_return) a.last_return[e] = ret;
    }
    if (st_bits) atomicOr(a.status, (int)st_bits);
  }
  if (WRITE_OBS) {
    store_obs_tile(a.obs, s_tile_wave, wave_env0, a.n, a.tb.n_obs, lane, grp, x, so, write_row);
  }
}

template <bool AUTORESET, bool WRITE_OBS, bool FIXES>
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
  uint4 cold, hot;
  load_step_state(a.st, e, cold, hot);
  if (AUTORESET) {  // the episode draw needs the sticky budget and the episode number
    const uint4 full = load_cold(a.st, e);
    cold.z = full.z;
    cold.w = full.w;
  }
  const int32_t act = load_action(a, e);
  step_tile<AUTORESET, WRITE_OBS, FIXES>(a, s_tile[wave], wave_env0, lane, l, grp, valid, e, cold, hot, act);
}

#endif  // W2A_STEP_HIP_H
