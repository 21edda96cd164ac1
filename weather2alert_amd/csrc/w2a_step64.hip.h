// w2a_step64.hip.h -- k_step64: one day for every env (env.py:238-262), the lean form of k_step for batches with
// faithful semantics (w2a_step picks it from 131 072 envs up; the in-kernel autoreset is a rare per-lane epilogue).
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_STEP64_HIP_H
#define W2A_STEP64_HIP_H

// Why a second geometry. k_step serves an env with a 4-lane group from start to end, so the per-env integer
// work (budget gate, history, termination; env.py:242-260) is executed by four lanes each and a wave has only
// 16 x 28 B of streamed state in flight during its first memory hop. Here a wave owns 64 consecutive envs and
// changes its lane <-> env mapping between phases:
//   A  lane = env          state + action loads (three fully coalesced streams: 768 + 768 + 256 B per wave),
//                          budget gate / history / termination once per env, descriptors to LDS
//   B  8 lanes = one row   two passes of 4 rounds; a round serves 8 envs, lane p of a group owns floats 4p..4p+3
//                          of the env's 128-B feature row and coefficient row(s): every gather instruction covers
//                          8 whole lines, a pass has 8..12 of them in flight per lane. fp64 FMA, DPP all-reduce
//                          over the 8 lanes, logits to LDS; the lanes scatter their row fragment into the packed
//                          observation tile (reference column order), flushed per pass as coalesced 16-B stores
//   C  lane = env          sigmoid, reward, return accumulator; reward / done / state written as whole lines
// What is computed is identical to k_step<false, WRITE_OBS, false, false> up to the order of the fp64 additions
// (4-term chains + 8-lane tree instead of 8-term chains + 4-lane tree: ~1e-16 relative on the logits).
#ifndef W2A_S64_MIN_ENVS
#define W2A_S64_MIN_ENVS 131072     // batch size from which w2a_step picks this kernel by itself
#endif
#define S64_ENVS 64                 // envs per wave
#define S64_WAVES (BLOCK / 64)
#define S64_PASS_ENVS 32            // envs per observation flush (4 rounds of 8)
#define S64_ROUNDS 4

// env.py:242-256 for one env from its packed state and today's action: budget gate, history, termination.
// Shared by k_step64 and k_posterior_mean (which must agree on `actual` and the run-time fields).
struct Day {
  uint32_t t, used, streak, hist, ndays, atb, actual, used2, hist2, st_bits, last;
  int32_t budget, act;
  bool done;
};
__device__ __forceinline__ Day derive_day(const u3 h, const u3 c, int32_t act) {
  Day d;
  d.st_bits = 0;
  if (act != 0 && act != 1) { d.st_bits |= W2A_ST_BAD_ACTION; act = 1; }
  d.act = act;                      // the action as the step uses it (a bad one counts as 1)
  d.last = D0_LAST(h.a);            // yesterday's granted alert
  d.t = D0_T(h.a); d.used = D0_USED(h.a); d.streak = D0_STREAK(h.a);
  d.hist = D1_HIST(h.b); d.ndays = D1_NDAYS(h.b);
  d.budget = (int32_t)c.a;
  if (D1_FIN(h.b)) d.st_bits |= W2A_ST_STEP_AFTER_DONE;
  d.atb = ((int32_t)d.used == d.budget) ? 1u : 0u;          // at_budget BEFORE today's action (env.py:242)
  d.actual = (act == 1 && d.atb) ? 0u : (uint32_t)act;       // env.py:243-246
  d.used2 = d.used + d.actual;
  d.hist2 = ((d.hist << 1) | d.actual) & 0x3FFFu;
  d.done = (d.t + 1 >= d.ndays);                             // env.py:256
  return d;
}
// env.py:190-193 as the row's run-time slots 24..27: alert_lag1 = today's action for t > 0 (Q3), streak before
// today's action (Q4), remaining budget after it, the agent's 14-day count (Q1)
__device__ __forceinline__ float4 runtime_fields(const Day &d, bool fix_lag = false) {
  // W2A_FIX_LAG: alert_lag1 is yesterday's granted alert instead of today's (Q3)
  return make_float4((d.t > 0) ? (float)(fix_lag ? d.last : d.actual) : 0.0f, (float)d.streak,
                     (float)(d.budget - (int32_t)d.used2), (float)__popc(d.hist2));
}
// overwrite component `idx` (0..3) of a float4 without dynamic register indexing
__device__ __forceinline__ void set_comp4(float4 &x, int idx, float v) {
  if (idx == 0) x.x = v;
  if (idx == 1) x.y = v;
  if (idx == 2) x.z = v;
  if (idx == 3) x.w = v;
}

template <bool FIXES>
struct S64WaveT {
  uint2 desc[S64_ENVS];             // {float4 index of the feature row, float4 index of the coefficient rows | need_eff << 31}
  float4 rt[S64_ENVS];              // run-time fields alert_lag1, alert_streak, remaining_budget, alert_2wks (slots 24..27)
  float4 rt2[FIXES ? S64_ENVS : 1]; // FIXES / W2A_FIX_OBS: the same fields as the NEXT day's observation shows them
  float2 z[S64_ENVS];               // {(float) baseline logit, (float) effectiveness logit (-inf: gate closed)}
  float tile[S64_PASS_ENVS * ROWF]; // packed observation rows of one pass (+ scratch words behind them)
};

// Store cache policy (A/B): bit 0 = observation rows, bit 1 = reward / done / state leave with `sc1`
// (write-through: the line is not kept in the XCD's L2, MI355X_MICROARCH.md "stores of each flavour"), so the
// streamed outputs do not evict the coefficient rows the gathers want to find there. 0 = nt observation stores.
#ifndef W2A_S64_SC1
#define W2A_S64_SC1 0
#endif
#ifndef W2A_S64_SKIP_EFF
#define W2A_S64_SKIP_EFF 1  // skip the effectiveness dot product in rounds where no env alerts today (A/B)
#endif
#ifndef W2A_S64_NT_STATE
#define W2A_S64_NT_STATE 0  // A/B: bit 0 = hot3, bit 1 = stepc, bit 2 = the packed mirror's words loaded non-temporally
#endif
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void st16_sc1(void *p, v4f v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st12_sc1(void *p, v3u v) {
  asm volatile("global_store_dwordx3 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st8_sc1(void *p, v2u v) {
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st4_sc1(void *p, uint32_t v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st1_sc1(void *p, uint32_t v) {
  asm volatile("global_store_byte %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

#ifndef W2A_S64_MIN_WAVES
#define W2A_S64_MIN_WAVES 4  // waves/SIMD the kernel is compiled for (<= 128 VGPRs)
#endif
#ifndef W2A_S64_PRIO
#define W2A_S64_PRIO 3  // s_setprio level of a wave between its row gathers and phase C (0 = off; A/B in DESIGN.md section 4)
#endif
#ifndef W2A_S64_TILES
#define W2A_S64_TILES 1  // consecutive 64-env tiles per wave (2 and 4, with the next tile's state loads in flight, measured slower)
#endif
// One 64-env tile of one wave, phases A..C, from the tile's already loaded state words and action.
// REWARD_GIVEN: a.reward already holds today's reward (w2a_posterior_mean_reward ran on the same state and actions):
// no coefficient gather and no logits here, the rest of env.py:238-262 as usual.
// AUTORESET: in-kernel autoreset (batches that are not in lock step: masked resets, ragged episode lengths; loops recorded
// into a hipGraph, where the host cannot launch a reset kernel between two steps): an env whose terminal step has just run
// draws its next episode and gets that episode's first observation, as in k_step<AUTORESET> -- here as a rare per-lane
// epilogue (one env-day in n_days). With PACKED (a lock-step batch: all envs of a tile restart in the same launch) the
// epilogue also writes the mirror's words of the new episode and puts the tile's day word back to 0.
// FIXES: the corrected-semantics flags of a.tb.fixes (W2A_FIX_*: Q1 agent's 14-day count into the historical column,
// Q3 true lag, Q5 live over-budget penalty, Q6 observation of the day the next action applies to), as in k_step<..., FIXES>.
template <bool WRITE_OBS, bool REWARD_GIVEN, bool PACKED, bool AUTORESET = false, bool FIXES = false>
__device__ __forceinline__ void s64_tile(const StepArgs &a, S64WaveT<FIXES> &sw, const int lane, const int64_t wave_env0,
                                         const bool valid, const uint32_t e, const u3 h, const u3 c, const int32_t act,
                                         const int4 so) {
  // ---------------------------------------------------------------- phase A: lane = env
  const Day d = derive_day(h, c, act);
  const uint32_t t = d.t, used2 = d.used2, hist2 = d.hist2, streak = d.streak, ndays = d.ndays;
  const uint32_t actual = d.actual, atb = d.atb, st_bits = d.st_bits;
  const int32_t budget = d.budget;
  const bool done = d.done;
  // W2A_STEP_NEXT_STEP: an env whose terminal step ran on the previous call restarts on this one (epilogue below); the
  // call is no env step for it: reward 0, not done, no status bit, its action ignored
  const bool restart_in = AUTORESET && a.next_step && D1_FIN(h.b);
  const uint32_t fx = FIXES ? a.tb.fixes : 0u;
  const bool fix_obs = FIXES && WRITE_OBS && (fx & W2A_FIX_OBS);
  // The effectiveness logit enters the reward through eff * gate * actual (env.py:218-221): its coefficient row is
  // fetched only for envs that alert today AND whose gate is open. The gate flag of (day, row) comes from a 1 KB
  // per-day bitmap (L2-resident) when the tables carry one; without it every alerting env fetches the row and the
  // closed gate acts through the -inf logit.
  uint32_t need_eff = actual;
  if (a.tb.gate_bits && actual)
    need_eff = (a.tb.gate_bits[t * (uint32_t)a.tb.gate_words + (c.b >> 5)] >> (c.b & 31u)) & 1u;
  {
    // feature row of day t (pre-increment, Q6) and the env's coefficient rows, as float4 indices (32-bit: table
    // sizes are validated in w2a_create). The effectiveness row is fetched only on alert days (k_step).
    const uint32_t day_row = t * (uint32_t)(a.tb.S_w * a.tb.Y) + c.b;
    const uint32_t wrow = W_COL(c.c) * (uint32_t)a.tb.n_samples + W_SAMPLE(c.c);
    // bit 31 of the row index (FIXES only): the observation is the NEXT day's row (W2A_FIX_OBS, not on the terminal step)
    sw.desc[lane] = make_uint2((day_row * (ROWF / 4)) | ((fix_obs && !done) ? 0x80000000u : 0u),
                               (wrow * (2 * ROWF / 4)) | (need_eff << 31));
    sw.rt[lane] = runtime_fields(d, (fx & W2A_FIX_LAG) != 0u);
    if (fix_obs)  // env.py:190-193 after today's action: lag = today's alert, the updated streak
      sw.rt2[FIXES ? lane : 0] = make_float4((float)actual, (float)(actual ? streak + 1 : 0), (float)(budget - (int32_t)used2),
                                 (float)__popc(hist2));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // ---------------------------------------------------------------- phase B: 8 lanes = one 128-B row
  const int p = lane & 7;   // float4 of the row owned by this lane
  const int g = lane >> 3;  // row group: env j = pass * 32 + round * 8 + g of the wave
  const int n_obs = a.tb.n_obs;
  // a finished env keeps its stale observation (env.py:257-262, Q6); W2A_FIX_OBS writes the last row instead
  const bool write_me = valid && (!done || fix_obs) && !restart_in;
  const uint32_t next_day = (uint32_t)(a.tb.S_w * a.tb.Y) * (ROWF / 4);  // float4 index distance to the next day's row
#pragma unroll 1
  for (int pass = 0; pass < S64_ENVS / S64_PASS_ENVS; ++pass) {
    if (REWARD_GIVEN && !WRITE_OBS) break;               // nothing to gather at all
    if (wave_env0 + pass * S64_PASS_ENVS >= a.n) break;  // wave-uniform
    float4 x[S64_ROUNDS], wb[S64_ROUNDS], we[S64_ROUNDS], xo[FIXES ? S64_ROUNDS : 1];
    bool need[S64_ROUNDS], nxt[S64_ROUNDS];
#pragma unroll
    for (int r = 0; r < S64_ROUNDS; ++r) {
      uint2 ds = sw.desc[pass * S64_PASS_ENVS + r * 8 + g];
      need[r] = (ds.y >> 31) != 0u;
      nxt[r] = FIXES && (ds.x >> 31) != 0u;
      if (FIXES) ds.x &= 0x7FFFFFFFu;
      const uint32_t wq = ds.y & 0x7FFFFFFFu;
      x[r] = a.tb.X[ds.x + p];
      // W2A_FIX_OBS: the row the observation shows (the next day's; the same line again where it is today's)
      if (fix_obs) xo[r] = a.tb.X[ds.x + (nxt[r] ? next_day : 0u) + p];
      wb[r] = we[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!REWARD_GIVEN) {
        wb[r] = a.tb.W[wq + p];
        // branch-free: groups that do not need the effectiveness row re-request their baseline line (already in
        // flight, no extra fabric traffic) and the value is dropped below. A conditional load would make the
        // compiler wait for it (s_waitcnt vmcnt(0)) at the end of its branch, before the next round's loads issue:
        // one exposed memory round trip per round with an alert.
        we[r] = a.tb.W[wq + (need[r] ? ROWF / 4 : 0) + p];
      }
    }
#if W2A_S64_PRIO
    // waves whose gathers are in flight or have landed are issued ahead of waves still in phase A: the rows they hold are
    // turned into stores sooner (measured: iid 34.4 -> 34.0 us, sorted 30.9 -> 30.3 us, always-alert 46.3 -> 44.5 us)
    __builtin_amdgcn_s_setprio(W2A_S64_PRIO);
#endif
#pragma unroll
    for (int r = 0; r < S64_ROUNDS; ++r) {
      const int j = pass * S64_PASS_ENVS + r * 8 + g;
      if (p == RT_QUAD) x[r] = sw.rt[j];
      if (FIXES && (fx & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0) {
        // the agent's 14-day count also replaces the historical 'alerts_2wks' column, so it feeds the reward (Q1)
        const float a2w = sw.rt[j].w;
        if (p == (a.tb.slot_hist2w >> 2)) set_comp4(x[r], a.tb.slot_hist2w & 3, a2w);
        if (fix_obs && p == (a.tb.slot_hist2w >> 2)) set_comp4(xo[r], a.tb.slot_hist2w & 3, a2w);
      }
      if (fix_obs && p == RT_QUAD) xo[r] = nxt[r] ? sw.rt2[FIXES ? j : 0] : sw.rt[j];
      if (!REWARD_GIVEN) {
      // env.py:207-217: two 28-term dot products, fp64 accumulation (products of f32 values are exact in fp64)
      const double x0 = (double)x[r].x, x1 = (double)x[r].y, x2 = (double)x[r].z, x3 = (double)x[r].w;
      double zb = x0 * (double)wb[r].x;
      zb = fma(x1, (double)wb[r].y, zb);
      zb = fma(x2, (double)wb[r].z, zb);
      zb = fma(x3, (double)wb[r].w, zb);
      zb += dpp_f64<0xB1>(zb);   // lane ^ 1
      zb += dpp_f64<0x4E>(zb);   // lane ^ 2
      zb += dpp_f64<0x141>(zb);  // row_half_mirror: the other quad of the 8-lane group
      double ze = 0.0;
#if W2A_S64_SKIP_EFF
      if (__any(need[r]))  // wave-uniform: most rounds have no alert today (eff * actual = 0 whatever eff is)
#endif
      {
        // groups without need carry their baseline row here: harmless, their ze is never used (phase C)
        ze = x0 * (double)we[r].x;
        ze = fma(x1, (double)we[r].y, ze);
        ze = fma(x2, (double)we[r].z, ze);
        ze = fma(x3, (double)we[r].w, ze);
        // effectiveness gate heat_qi > 0.5 (env.py:218): slot 30 holds the 0/1 flag with a zero coefficient; a
        // closed gate drives the logit to -inf so that the sigmoid is exactly 0
        if (p == GATE_QUAD && !(x[r].z > 0.5f)) ze = -__builtin_inf();
        ze += dpp_f64<0xB1>(ze);
        ze += dpp_f64<0x4E>(ze);
        ze += dpp_f64<0x141>(ze);
      }
      if (p == 0) sw.z[j] = make_float2((float)zb, (float)ze);
      }
      if (WRITE_OBS) {
        // branch-free scatter into the packed tile; slots that are not observation columns go to scratch words
        const int base = (r * 8 + g) * n_obs;
        const int trash = S64_PASS_ENVS * n_obs + (lane & 31);
        const float4 ov = fix_obs ? xo[FIXES ? r : 0] : x[r];
        sw.tile[so.x >= 0 ? base + so.x : trash] = ov.x;
        sw.tile[so.y >= 0 ? base + so.y : trash] = ov.y;
        sw.tile[so.z >= 0 ? base + so.z : trash] = ov.z;
        sw.tile[so.w >= 0 ? base + so.w : trash] = ov.w;
      }
    }
    if (WRITE_OBS) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int64_t env0 = wave_env0 + pass * S64_PASS_ENVS;
      float *dst = reinterpret_cast<float *>(a.obs) + env0 * n_obs;
      // which envs of this pass write: their phase-A lanes are pass*32 .. pass*32+31
      const unsigned long long wm_all = __ballot(write_me);
      const uint32_t wm = (uint32_t)(wm_all >> (pass * S64_PASS_ENVS));
      const bool full = env0 + S64_PASS_ENVS <= a.n;
      if (full && wm == 0xFFFFFFFFu) {
        const int chunks = (S64_PASS_ENVS * n_obs) >> 2;  // 32 * n_obs floats: a multiple of 4
#pragma unroll
        for (int c0 = 0; c0 < (S64_PASS_ENVS * ROWF) / 4; c0 += 64) {
          const int ch = c0 + lane;
          if (ch < chunks) {
            const v4f v = reinterpret_cast<const v4f *>(sw.tile)[ch];
#if W2A_S64_SC1 & 1
            st16_sc1(reinterpret_cast<v4f *>(dst) + ch, v);
#elif W2A_NT_OBS
            __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(dst) + ch);
#else
            reinterpret_cast<v4f *>(dst)[ch] = v;
#endif
          }
        }
      } else if (wm != 0u) {
        // ragged tail, or some env of the pass keeps its stale row: element-wise, masked
        const int total = S64_PASS_ENVS * n_obs;
        for (int i = lane; i < total; i += 64) {
          const int jj = i / n_obs;
          if ((wm >> jj) & 1u) dst[i] = sw.tile[i];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the tile is rewritten by the next pass
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (!WRITE_OBS) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }

#if W2A_S64_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  // ---------------------------------------------------------------- phase C: lane = env
  if (AUTORESET && valid && restart_in) {
    a.reward[e] = 0.0f;
    a.done[e] = 0;
  } else if (valid && a.skip_finished && D1_FIN(h.b)) {
    // W2A_STEP_SKIP_FINISHED: policy loops over batches that are not in lock step (w2a_policy_actions gave this env
    // action 0): the finished episode's state, return and observation stay as they are
    a.reward[e] = 0.0f;
    a.done[e] = 1;
  } else if (valid) {
  float r;
  if (REWARD_GIVEN) {
    r = a.reward[e];
  } else {
    const float2 z = sw.z[lane];
    const float base = sigmoid_f32(z.x);  // env.py:211-221
    const float eff = need_eff ? sigmoid_f32(z.y) : 0.0f;  // not needed = no alert or closed gate: eff * actual = 0
    r = -(1000.0f / 152.0f) * base * (1.0f - eff);
    if (FIXES && (fx & W2A_FIX_PENALTY) && d.act == 1 && atb) r = -1.0f;  // env.py:223-224 made live (Q5)
  }
  const uint32_t t2 = done ? t : t + 1;
  const uint32_t streak2 = done ? streak : (actual ? streak + 1 : 0);  // env.py:260
  const float ret = __uint_as_float(h.c) + r;
  u3 h2;
  h2.a = pack_d0(t2, used2, streak2, actual, atb);
  h2.b = pack_d1(hist2, ndays, done ? 1u : 0u);
  h2.c = __float_as_uint(ret);
#if W2A_S64_SC1 & 2
  // the packed variant's state word goes to the lock-step mirror here too (the host marks the mirror current)
  if (PACKED) st8_sc1(&a.st.pk_hot[e], v2u{pk_pack_hot(used2, streak2, hist2, done ? 1u : 0u), h2.c});
  else st12_sc1(&a.st.hot3[e], v3u{h2.a, h2.b, h2.c});
  if (!REWARD_GIVEN) st4_sc1(&a.reward[e], __float_as_uint(r));
  st1_sc1(&a.done[e], done ? 1u : 0u);
#else
  if (PACKED) a.st.pk_hot[e] = make_uint2(pk_pack_hot(used2, streak2, hist2, done ? 1u : 0u), h2.c);
  else a.st.hot3[e] = h2;
  if (!REWARD_GIVEN) a.reward[e] = r;
  a.done[e] = done ? 1 : 0;
#endif
  // the tile's day word moves on with its envs (lock step: the same day and length in every lane; the terminal step
  // leaves the day where it is, env.py:256-259)
  if (PACKED && lane == 0) a.st.pk_day[wave_env0 >> 6] = t2;
  if (done && a.last_return) a.last_return[e] = ret;
  if (st_bits) atomicOr(a.status, (int)st_bits);
  }
  if (AUTORESET) {
    // env.py:162-181 for the envs that finished today: next episode from the device RNG (draw_episode, the k_reset /
    // k_step<AUTORESET> code), its state words over the ones phase C has just written, its day-0 row as the returned
    // observation (the pass flush left the finished env's row alone).
    // The first observation is written the way phase B reads rows: 8 lanes = one 128-B row, 8 envs per round, descriptors
    // through LDS. Until round 5 every restarting LANE fetched its own row (8 x 16 B from 64 different lines per
    // instruction) and stored its 29 floats one by one at a 116-B stride: on the day a lock-step batch restarts -- every
    // lane of every wave -- that cost ~600 us per 1 M envs against 56 us for k_reset, i.e. 2.5-4 us per step averaged over
    // an episode (profiles/r03/exp_autoreset_modes.log: 39.8 vs 35.9 us; profiles/r05/bench_graph51.log: 36.6 vs 34.1).
    const bool rs = valid && (a.next_step ? restart_in : done);
    const unsigned long long rs_mask = __ballot(rs);
    if (rs_mask) {  // wave-uniform
      uint2 ds = make_uint2(0xFFFFFFFFu, 0u);
      if (rs) {
        const uint4 cold = load_cold(a.st, e);
        const Episode ep = draw_episode(a.tb, a.rc, (uint64_t)(a.gid0 + e), cold.w + 1, (int32_t)cold.z);
        if (ep.bad) atomicOr(a.status, (int)W2A_ST_BAD_EPISODE);
        store_episode(a.st, e, make_uint4(ep.ep_row, ep.ep_w, (uint32_t)ep.sticky, cold.w + 1),
                      make_uint4(pack_d0(0, 0, 0, 0, 0), pack_d1(0, ep.ndays, 0), __float_as_uint(0.0f), (uint32_t)ep.budget));
        if (PACKED) {  // the mirror's words of the new episode, over the ones phase C has just written (k_pack_state's packing)
          a.st.pk_hot[e] = make_uint2(pk_pack_hot(0u, 0u, 0u, 0u), __float_as_uint(0.0f));
          a.st.pk_c[e] = make_uint2(pk_budget16((uint32_t)ep.budget) | (W_COL(ep.ep_w) << 16),
                                    (ep.ep_row & 0x3FFFFFu) | (W_SAMPLE(ep.ep_w) << 22));
        }
        ds = make_uint2(ep.ep_row * (ROWF / 4), (uint32_t)ep.budget);  // float4 index of the day-0 row, the budget
      }
      if (WRITE_OBS) {
        sw.desc[lane] = ds;  // (phase B is over: the descriptor slots are free)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll 1
        for (int r = 0; r < S64_ENVS / 8; ++r) {
          if (!((rs_mask >> (r * 8)) & 0xFFull)) continue;  // wave-uniform: no env of this round restarts
          const int j = r * 8 + g;
          const uint2 dj = sw.desc[j];
          if (dj.x != 0xFFFFFFFFu) {
            float4 v = a.tb.X[dj.x + p];  // day 0
            if (p == RT_QUAD) v = make_float4(0.0f, 0.0f, (float)(int32_t)dj.y, 0.0f);
            if (FIXES && (fx & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0 && p == (a.tb.slot_hist2w >> 2))
              set_comp4(v, a.tb.slot_hist2w & 3, 0.0f);  // the agent's (empty) history replaces the column
            float *row = reinterpret_cast<float *>(a.obs) + (size_t)(wave_env0 + j) * n_obs;
            if (so.x >= 0) row[so.x] = v.x;
            if (so.y >= 0) row[so.y] = v.y;
            if (so.z >= 0) row[so.z] = v.z;
            if (so.w >= 0) row[so.w] = v.w;
          }
        }
      }
      // packed form = lock step: the envs of a tile finish, and restart, together -- the tile is on day 0 again
      if (PACKED && lane == 0) a.st.pk_day[wave_env0 >> 6] = 0u;
    }
  }
}

// packed variant: 8 + 8 B per env, the uniform day from the tile's day word (device memory: a recorded step finds the right
// day on every replay), the episode length from the kernel arguments; the canonical words are rebuilt in registers so
// that the tile code is shared
__device__ __forceinline__ void s64_load_packed(const StepArgs &a, uint32_t e, uint32_t day, u3 &h, u3 &c, int32_t &act) {
#if W2A_S64_NT_STATE & 4  // A/B: the mirror's words loaded non-temporally (streamed once per step: no reason to keep them in L2)
  const v2u phv = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(&a.st.pk_hot[e]));
  const v2u pcv = __builtin_nontemporal_load(reinterpret_cast<const v2u *>(&a.st.pk_c[e]));
  const uint2 ph = make_uint2(phv.x, phv.y), pc = make_uint2(pcv.x, pcv.y);
#else
  const uint2 ph = a.st.pk_hot[e];
  const uint2 pc = a.st.pk_c[e];
#endif
  h.a = pack_d0(day, PK_USED(ph.x), PK_STREAK(ph.x), PK_HIST(ph.x) & 1u, 0u);
  h.b = pack_d1(PK_HIST(ph.x), (uint32_t)a.uni_nd, PK_FIN(ph.x));
  h.c = ph.y;
  c.a = pc.x & 0xFFFFu;
  // budgets the 16-bit field cannot hold (w2a_common.hip.h, pk_budget16): the lane reads the canonical word, which every
  // reset path writes and nothing changes while an episode runs -- a rare divergent 4-B load (env.py:167-178: a budget
  // keyword of any size, a sticky centred budget that has walked upwards)
  if (c.a == PK_BUDGET_ESCAPE) c.a = a.st.stepc[e].a;
  c.b = pc.y & 0x3FFFFFu;
  c.c = PACK_W(pc.x >> 16, pc.y >> 22);
  act = load_action(a, e);
}
__device__ __forceinline__ void s64_load_state(const StepArgs &a, uint32_t e, u3 &h, u3 &c, int32_t &act) {
#if W2A_S64_NT_STATE & 1
  const v3u hv = __builtin_nontemporal_load(reinterpret_cast<const v3u *>(&a.st.hot3[e]));
  h.a = hv.x; h.b = hv.y; h.c = hv.z;
#else
  h = a.st.hot3[e];
#endif
#if W2A_S64_NT_STATE & 2
  const v3u cv = __builtin_nontemporal_load(reinterpret_cast<const v3u *>(&a.st.stepc[e]));
  c.a = cv.x; c.b = cv.y; c.c = cv.z;
#else
  c = a.st.stepc[e];
#endif
  act = load_action(a, e);
}

// A wave owns W2A_S64_TILES consecutive 64-env tiles. The first memory hop of a tile (its three coalesced state /
// action streams, 28 B per env) has few bytes in flight and a full memory round trip of latency; requesting the
// NEXT tile's words before the current tile's phases run takes that hop off the wave's critical path for every
// tile but the first (7 more VGPRs). Measured: DESIGN.md §4.
template <bool WRITE_OBS, bool REWARD_GIVEN, bool PACKED = false, bool AUTORESET = false, bool FIXES = false>
__global__ __launch_bounds__(BLOCK, FIXES ? 3 : W2A_S64_MIN_WAVES) void k_step64(const StepArgs a) {
  __shared__ __attribute__((aligned(16))) S64WaveT<FIXES> s_w[S64_WAVES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  S64WaveT<FIXES> &sw = s_w[wave];
  const uint32_t lb = logical_block(blockIdx.x, gridDim.x >> 3);  // grid is a multiple of 8 workgroups
  const int64_t wave_env0 = ((int64_t)lb * S64_WAVES + wave) * (S64_ENVS * W2A_S64_TILES);
  if (wave_env0 >= a.n) return;  // whole wave past the end (padding tiles); no workgroup barrier is used below
  // packed variant: the day of this wave's tile(s), one word per 64 envs (wave-uniform address)
  uint32_t dayn = 0;
  if (PACKED) {
    dayn = a.st.pk_day[wave_env0 >> 6];
    // poisoned mirror: the host could not keep the packed form current for a recorded graph (w2a_bookkeeping.h,
    // graph_packed) -- a replay must not step stale state: nothing is touched, the status word says why
    if (dayn == W2A_PK_DAY_POISON) {
      if (lane == 0) atomicOr(a.status, (int)W2A_ST_STALE_GRAPH);
      return;
    }
  }
  u3 hn, cn;
  int32_t an;
  {
    const int64_t env = wave_env0 + lane;
    if (PACKED) s64_load_packed(a, (uint32_t)(env < a.n ? env : a.n - 1), dayn, hn, cn, an);
    else s64_load_state(a, (uint32_t)(env < a.n ? env : a.n - 1), hn, cn, an);
  }
  // slot -> observation column of the 4 row slots this lane owns in the row phase; requested together with the
  // state words so that its latency is not exposed in front of the gathers
  int4 so = make_int4(-1, -1, -1, -1);
  if (WRITE_OBS) so = reinterpret_cast<const int4 *>(a.slot_obs)[lane & 7];
#pragma unroll 1
  for (int i = 0; i < W2A_S64_TILES; ++i) {
    const int64_t env0 = wave_env0 + (int64_t)i * S64_ENVS;
    if (env0 >= a.n) break;  // wave-uniform
    const int64_t env = env0 + lane;
    const bool valid = env < a.n;
    const uint32_t e = (uint32_t)(valid ? env : (a.n - 1));  // clamp: surplus lanes shadow the last env, never store
    const u3 h = hn, c = cn;
    const int32_t act = an;
    if (i + 1 < W2A_S64_TILES && env0 + S64_ENVS < a.n) {
      const int64_t en = env + S64_ENVS;
      if (PACKED) {
        dayn = a.st.pk_day[(env0 + S64_ENVS) >> 6];
        s64_load_packed(a, (uint32_t)(en < a.n ? en : a.n - 1), dayn == W2A_PK_DAY_POISON ? 0u : dayn, hn, cn, an);
      } else s64_load_state(a, (uint32_t)(en < a.n ? en : a.n - 1), hn, cn, an);
    }
    s64_tile<WRITE_OBS, REWARD_GIVEN, PACKED, AUTORESET, FIXES>(a, sw, lane, env0, valid, e, h, c, act, so);
    // the per-wave LDS record is rewritten by the next tile
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// canonical -> lock-step mirror (entering the packed form: once per episode) and back (before anything else reads the
// canonical arrays). The day of a 64-env tile is that of its first env (lock step: the host packs only batches it knows
// to be on one day); the episode length is a table constant the host passes.
__global__ void k_pack_state(StateArrays st, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u3 h = st.hot3[i];
  const u3 c = st.stepc[i];
  st.pk_hot[i] = make_uint2(pk_pack_hot(D0_USED(h.a), D0_STREAK(h.a), D1_HIST(h.b), D1_FIN(h.b)), h.c);
  st.pk_c[i] = make_uint2(pk_budget16(c.a) | (W_COL(c.c) << 16), (c.b & 0x3FFFFFu) | (W_SAMPLE(c.c) << 22));
  if ((i & 63) == 0) st.pk_day[i >> 6] = D0_T(h.a);
}
__global__ void k_unpack_state(StateArrays st, int64_t n, int32_t n_days) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint2 ph = st.pk_hot[i];
  const uint32_t t = st.pk_day[i >> 6];
  const uint32_t used = PK_USED(ph.x), hist = PK_HIST(ph.x), fin = PK_FIN(ph.x), last = hist & 1u;
  // at_budget as the last step left it (env.py:242: decided BEFORE that day's action); False after a reset
  const uint32_t atb = ((t > 0 || fin) && (int32_t)(used - last) == (int32_t)st.stepc[i].a) ? 1u : 0u;
  u3 h;
  h.a = pack_d0(t, used, PK_STREAK(ph.x), last, atb);
  h.b = pack_d1(hist, (uint32_t)n_days, fin);
  h.c = ph.y;
  st.hot3[i] = h;
}
__global__ void k_poison_mirror(StateArrays st, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < ((n + 63) >> 6)) st.pk_day[i] = W2A_PK_DAY_POISON;
}

#endif  // W2A_STEP64_HIP_H
