// w2a_rollout.hip.h -- k_rollout: on-device policy rollout
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_ROLLOUT_HIP_H
#define W2A_ROLLOUT_HIP_H

// ----------------------------------------------------------------------------------------
// on-device policy rollout: many days per launch, coefficients kept in registers
// ----------------------------------------------------------------------------------------
// Same per-day arithmetic as k_step (env.py:238-262) for up to n_steps days or until the episode ends; the
// action comes from a policy evaluated in the kernel on what the reference's agent would see: the lagging
// observation (row of day t-1, Q6), the remaining budget and the day. No observation rows are written; the
// packed state is advanced so step()/rollout() calls can be mixed.
struct RolloutArgs {
  DevTables tb;
  StateArrays st;
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
  uint32_t *attempt_mask;         // [n][mask_words] bit d = alert ATTEMPTED on day d, granted or not (nullable)
  int32_t mask_words;
  float *last_return;
  float *ret_snapshot;            // [n] running episode return after the step that leaves t == n_days - 2 (nullable):
                                  // what the reference's logging callbacks read (callbacks.py:47-48,128-132)
  const uint32_t *order;          // [n] visiting order (a permutation of the env ids; nullable = identity): w2a_rollout_order
                                  // sorts the envs by feature row, so the 16 envs of a wave read one or two table lines
                                  // per day instead of 16 (the day loop is otherwise bound by those reads: 19 GB of
                                  // fabric traffic per 1 M-env episode, profiles/r02/rollout_counters.log)
};

// visiting order = counting sort of the env ids by feature row (cold.x = county_w * Y + year_i). The order inside a row
// is whatever the atomics give -- any permutation is correct, see above.
//   count + rank   cnt[row] = envs of the row, rank[env] = the env's position inside its row: ONE returning atomic per env.
//                  A whole-batch k_reset does this itself when an order workspace is attached (w2a_rollout_order_attach:
//                  it holds the env's row anyway and the atomic hides behind its observation stores); k_order_rank is
//                  the same pass for every other case (first use, masked resets, in-kernel autoresets, restores)
//   scan           start[row] = first position of the row (exclusive scan of cnt, start[rows] = n) and, in the same pass,
//                  tile_start[row] = first 16-env subtile of the row (exclusive scan of ceil(cnt / 16)): the matrix-core
//                  rollout's subtile list needs no pass of its own (k_rm_tiles)
//   place          order[start[row] + rank[env]] = env: no atomics
// (Until round 4: histogram atomics, scan, a second pass of returning atomics for the scatter, and a one-workgroup tile
// list: 47 + 8 + 49 + 5 + 24 us per 1 M-env episode, profiles/r04/kernel_trace_rollout_bench.txt.)
__global__ void k_order_rank(const uint4 *cold, uint32_t *cnt, uint32_t *rank, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rank[i] = atomicAdd(&cnt[cold[i].x], 1u);
}
__global__ __launch_bounds__(1024) void k_order_scan(const uint32_t *cnt, int32_t rows, uint32_t *start, uint32_t *tile_start) {
  __shared__ uint32_t s_sum[1024], s_til[1024];  // one workgroup
  const int tid = threadIdx.x;
  const int chunk = (rows + 1023) / 1024;
  const int r0 = min(rows, tid * chunk), r1 = min(rows, r0 + chunk);
  uint32_t mine = 0, mine_t = 0;
  for (int r = r0; r < r1; ++r) { const uint32_t c = cnt[r]; mine += c; mine_t += (c + 15u) >> 4; }
  s_sum[tid] = mine; s_til[tid] = mine_t;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t v = tid >= d ? s_sum[tid - d] : 0u, w = tid >= d ? s_til[tid - d] : 0u;
    __syncthreads();
    s_sum[tid] += v; s_til[tid] += w;
    __syncthreads();
  }
  uint32_t run = s_sum[tid] - mine, run_t = s_til[tid] - mine_t;
  for (int r = r0; r < r1; ++r) {
    const uint32_t c = cnt[r];
    start[r] = run; tile_start[r] = run_t;
    run += c; run_t += (c + 15u) >> 4;
  }
  if (tid == 1023) { start[rows] = s_sum[1023]; tile_start[rows] = s_til[1023]; }
}
__global__ void k_order_place(const uint4 *cold, const uint32_t *start, const uint32_t *rank, uint32_t *order, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[start[cold[i].x] + rank[i]] = (uint32_t)i;
}

// the built-in policies on what the reference's agent would see (shared by k_rollout and k_policy_actions)
__device__ __forceinline__ int32_t policy_action(int32_t kind, const w2a_policy &pol, uint64_t pstream, uint32_t t,
                                                 int32_t rem_now, float feat) {
  int32_t act = 0;
  if (kind == W2A_POLICY_ALWAYS) act = 1;
  else if (kind == W2A_POLICY_BERNOULLI) {
    const uint32_t u = (uint32_t)(w2a_mix64(pstream + (uint64_t)(t + 1) * 0x9E3779B97F4A7C15ull) >> 32);
    act = ((float)u * 2.3283064365386963e-10f < pol.p) ? 1 : 0;
  } else if (kind == W2A_POLICY_THRESHOLD) act = (feat > pol.threshold) ? 1 : 0;
  else if (kind == W2A_POLICY_TABLE) {
    int32_t rr = rem_now < 0 ? 0 : (rem_now >= pol.table_R ? pol.table_R - 1 : rem_now);
    act = pol.table[(size_t)t * pol.table_R + rr] ? 1 : 0;
  }
  if (pol.require_budget && rem_now <= 0) act = 0;
  return act;
}

// Compiled per policy kind, with / without the day bitmaps, with / without corrected-semantics flags: the day loop is
// bound by instruction issue, and these wave-uniform choices otherwise cost branches and register moves every day.
template <int KIND, bool MASKS, bool FIXES>
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
  const uint32_t slot = (uint32_t)(valid ? env : (a.n - 1));
  const uint32_t e = a.order ? a.order[slot] : slot;  // the env this lane group serves
  uint4 c2, hot;
  load_step_state(a.st, e, c2, hot);
  const uint4 cold = load_cold(a.st, e);
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
  uint32_t att_word = 0, att_idx = 0xFFFFFFFFu;
  float snap = 0.0f;
  bool snapped = false;
  // feature the policy sees on its first day here: row of day max(t-1, 0) (lagging observation, Q6)
  float feat = 0.0f;
  if (KIND == W2A_POLICY_THRESHOLD)
    feat = Xf[((size_t)((a.pol.obs_lag && t > 0 ? t - 1 : t) * rows_per_day + cold.x)) * ROWF + a.pol_slot];
  bool active = !fin && valid;
  for (int s = 0; s < a.n_steps; ++s) {
    if (!__any(active)) break;
    // ---- policy
    const int32_t act = policy_action(KIND, a.pol, pstream, t, budget - (int32_t)used, feat);
    // ---- env.py:242-250
    const uint32_t atb_s = ((int32_t)used == budget) ? 1u : 0u;
    const uint32_t actual = (act == 1 && atb_s) ? 0u : (uint32_t)act;
    const uint32_t used2 = used + actual;
    const uint32_t hist2 = ((hist << 1) | actual) & 0x3FFFu;
    const uint32_t day_row = t * rows_per_day + cold.x;
    float4 x[QUADS];
#pragma unroll
    for (int q = 0; q < QUADS; ++q) x[q] = a.tb.X[day_row * (ROWF / 4) + l * QUADS + q];
    const float today = (KIND == W2A_POLICY_THRESHOLD) ? Xf[(size_t)day_row * ROWF + a.pol_slot] : 0.0f;
    const uint32_t fx = FIXES ? a.tb.fixes : 0u;
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
    float r = reward_from_logits(zb, ze, actual);
    if ((fx & W2A_FIX_PENALTY) && act == 1 && atb_s) r = -1.0f;
    if (active) {
      const bool done = (t + 1 >= ndays);
      ret += r;
      ret_total += r;
      alerts += (int32_t)actual;
      over += (act == 1 && atb_s) ? 1 : 0;
      if (MASKS && a.alert_mask && actual) {
        const uint32_t wi = t >> 5;
        if (wi != mask_idx) {
          if (mask_idx != 0xFFFFFFFFu && l == 0 && mask_idx < (uint32_t)a.mask_words)
            a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
          mask_idx = wi;
          mask_word = 0;
        }
        mask_word |= 1u << (t & 31);
      }
      if (MASKS && a.attempt_mask && act == 1) {
        const uint32_t wi = t >> 5;
        if (wi != att_idx) {
          if (att_idx != 0xFFFFFFFFu && l == 0 && att_idx < (uint32_t)a.mask_words)
            a.attempt_mask[(size_t)e * a.mask_words + att_idx] |= att_word;
          att_idx = wi;
          att_word = 0;
        }
        att_word |= 1u << (t & 31);
      }
      if (MASKS && (done ? t : t + 1) + 2 == ndays) { snap = ret_total; snapped = true; }
      used = used2; hist = hist2; last = actual; atb = atb_s;
      if (!done) { streak = actual ? streak + 1 : 0; t = t + 1; }
      else { fin = true; active = false; }
      feat = a.pol.obs_lag ? today : feat;
    }
    // lag 1 (faithful): the next decision sees today's row; lag 0 needs tomorrow's row
    if (KIND == W2A_POLICY_THRESHOLD && !a.pol.obs_lag && active)
      feat = Xf[(size_t)(t * rows_per_day + cold.x) * ROWF + a.pol_slot];
  }
  if (valid && l == 0) {
    store_hot(a.st, e, make_uint4(pack_d0(t, used, streak, last, atb), pack_d1(hist, ndays, fin ? 1u : 0u),
                                  __float_as_uint(ret_total), (uint32_t)budget));
    if (a.ret_out) a.ret_out[e] = ret;
    if (a.alerts_out) a.alerts_out[e] = alerts;
    if (a.attempts_over_budget) a.attempts_over_budget[e] = over;
    if (MASKS && a.alert_mask && mask_idx != 0xFFFFFFFFu && mask_idx < (uint32_t)a.mask_words)
      a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
    if (MASKS && a.attempt_mask && att_idx != 0xFFFFFFFFu && att_idx < (uint32_t)a.mask_words)
      a.attempt_mask[(size_t)e * a.mask_words + att_idx] |= att_word;
    if (MASKS && a.ret_snapshot && snapped) a.ret_snapshot[e] = snap;
    if (fin && a.last_return && !D1_FIN(hot.y)) a.last_return[e] = ret_total;
  }
}

// ----------------------------------------------------------------------------------------
// lane = env form of the day loop
// ----------------------------------------------------------------------------------------
// With the visiting order the envs of a wave share one or two feature rows, so a lane can afford to read its env's
// whole row every day (64 lanes, one or two distinct lines per load instruction). One lane per env then removes what
// the 4-lanes-per-env form pays per day: the cross-lane reductions and the state / policy arithmetic repeated on the
// three other lanes. The coefficients stay in registers as f32 (the converts are redone every day, W2A_RO64_F32COEF = 1)
// or as fp64 (twice the registers, half the converts).
#ifndef W2A_RO64_F32COEF
#define W2A_RO64_F32COEF 1
#endif
#define RO64_SLOTS 30  // slots 0..29 enter the logits (30 carries the gate flag with a zero coefficient, 31 is zero)

template <int KIND, bool MASKS, bool FIXES>
__global__ __launch_bounds__(BLOCK) void k_rollout64(const RolloutArgs a) {
  const int64_t slot64 = (int64_t)logical_block(blockIdx.x, gridDim.x >> 3) * BLOCK + threadIdx.x;
  if (slot64 - (threadIdx.x & 63) >= a.n) return;  // whole wave past the end
  const bool valid = slot64 < a.n;
  const uint32_t slot = (uint32_t)(valid ? slot64 : (a.n - 1));
  const uint32_t e = a.order ? a.order[slot] : slot;  // the env this lane serves
  uint4 c2, hot;
  load_step_state(a.st, e, c2, hot);
  const uint4 cold = load_cold(a.st, e);
  uint32_t t = D0_T(hot.x), used = D0_USED(hot.x), streak = D0_STREAK(hot.x), last = D0_LAST(hot.x);
  uint32_t atb = D0_ATB(hot.x), hist = D1_HIST(hot.y);
  const uint32_t ndays = D1_NDAYS(hot.y);
  const int32_t budget = (int32_t)hot.w;
  bool fin = D1_FIN(hot.y) != 0;
  float ret_total = __uint_as_float(hot.z);
  const uint32_t rows_per_day = (uint32_t)(a.tb.S_w * a.tb.Y);
  const uint32_t wrow = W_COL(cold.y) * (uint32_t)a.tb.n_samples + W_SAMPLE(cold.y);
  // the env's two coefficient rows, once per launch
#if W2A_RO64_F32COEF
  float wb[32], we[32];
#else
  double wb[32], we[32];
#endif
  {
    const float4 *wp = a.tb.W + (size_t)wrow * (2 * ROWF / 4);
#pragma unroll
    for (int q = 0; q < ROWF / 4; ++q) {
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
  uint32_t att_word = 0, att_idx = 0xFFFFFFFFu;
  float snap = 0.0f;
  bool snapped = false;
  float feat = 0.0f;
  if (KIND == W2A_POLICY_THRESHOLD)
    feat = Xf[((size_t)((a.pol.obs_lag && t > 0 ? t - 1 : t) * rows_per_day + cold.x)) * ROWF + a.pol_slot];
  bool active = !fin && valid;
  for (int s = 0; s < a.n_steps; ++s) {
    if (!__any(active)) break;
    const int32_t act = policy_action(KIND, a.pol, pstream, t, budget - (int32_t)used, feat);
    // ---- env.py:242-250
    const uint32_t atb_s = ((int32_t)used == budget) ? 1u : 0u;
    const uint32_t actual = (act == 1 && atb_s) ? 0u : (uint32_t)act;
    const uint32_t used2 = used + actual;
    const uint32_t hist2 = ((hist << 1) | actual) & 0x3FFFu;
    const uint32_t day_row = t * rows_per_day + cold.x;
    float xv[32];
    {
      const float4 *xp = a.tb.X + (size_t)day_row * (ROWF / 4);
#pragma unroll
      for (int q = 0; q < ROWF / 4; ++q) {
        if (q == RT_QUAD) continue;  // slots 24..27 are run-time fields
        const float4 v = xp[q];
        xv[4 * q] = v.x; xv[4 * q + 1] = v.y; xv[4 * q + 2] = v.z; xv[4 * q + 3] = v.w;
      }
    }
    const float today = (KIND == W2A_POLICY_THRESHOLD) ? Xf[(size_t)day_row * ROWF + a.pol_slot] : 0.0f;
    const uint32_t fx = FIXES ? a.tb.fixes : 0u;
    const float f_a2w = (float)__popc(hist2);
    xv[4 * RT_QUAD] = (t > 0) ? (float)((fx & W2A_FIX_LAG) ? last : actual) : 0.0f;
    xv[4 * RT_QUAD + 1] = (float)streak;
    xv[4 * RT_QUAD + 2] = (float)(budget - (int32_t)used2);
    xv[4 * RT_QUAD + 3] = f_a2w;
    if (FIXES && (fx & W2A_FIX_ALERTS_2WKS) && a.tb.slot_hist2w >= 0) {
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (k == a.tb.slot_hist2w) xv[k] = f_a2w;
    }
    double zb = 0.0, ze = 0.0;
#pragma unroll
    for (int k = 0; k < RO64_SLOTS; ++k) {
#if W2A_RO64_F32COEF
      asm volatile("" : "+v"(wb[k]), "+v"(we[k]));  // keep the coefficients f32: the converts are redone every day
#endif
      const double xk = (double)xv[k];
      zb = fma(xk, (double)wb[k], zb);
      ze = fma(xk, (double)we[k], ze);
    }
    if (!(xv[30] > 0.5f)) ze = -__builtin_inf();
    float r = reward_from_logits(zb, ze, actual);
    if ((fx & W2A_FIX_PENALTY) && act == 1 && atb_s) r = -1.0f;
    if (active) {
      const bool done = (t + 1 >= ndays);
      ret += r;
      ret_total += r;
      alerts += (int32_t)actual;
      over += (act == 1 && atb_s) ? 1 : 0;
      if (MASKS && a.alert_mask && actual) {
        const uint32_t wi = t >> 5;
        if (wi != mask_idx) {
          if (mask_idx != 0xFFFFFFFFu && mask_idx < (uint32_t)a.mask_words)
            a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
          mask_idx = wi;
          mask_word = 0;
        }
        mask_word |= 1u << (t & 31);
      }
      if (MASKS && a.attempt_mask && act == 1) {
        const uint32_t wi = t >> 5;
        if (wi != att_idx) {
          if (att_idx != 0xFFFFFFFFu && att_idx < (uint32_t)a.mask_words)
            a.attempt_mask[(size_t)e * a.mask_words + att_idx] |= att_word;
          att_idx = wi;
          att_word = 0;
        }
        att_word |= 1u << (t & 31);
      }
      if (MASKS && (done ? t : t + 1) + 2 == ndays) { snap = ret_total; snapped = true; }
      used = used2; hist = hist2; last = actual; atb = atb_s;
      if (!done) { streak = actual ? streak + 1 : 0; t = t + 1; }
      else { fin = true; active = false; }
      feat = a.pol.obs_lag ? today : feat;
    }
    if (KIND == W2A_POLICY_THRESHOLD && !a.pol.obs_lag && active)
      feat = Xf[(size_t)(t * rows_per_day + cold.x) * ROWF + a.pol_slot];
  }
  if (valid) {
    store_hot(a.st, e, make_uint4(pack_d0(t, used, streak, last, atb), pack_d1(hist, ndays, fin ? 1u : 0u),
                                  __float_as_uint(ret_total), (uint32_t)budget));
    if (a.ret_out) a.ret_out[e] = ret;
    if (a.alerts_out) a.alerts_out[e] = alerts;
    if (a.attempts_over_budget) a.attempts_over_budget[e] = over;
    if (MASKS && a.alert_mask && mask_idx != 0xFFFFFFFFu && mask_idx < (uint32_t)a.mask_words)
      a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
    if (MASKS && a.attempt_mask && att_idx != 0xFFFFFFFFu && att_idx < (uint32_t)a.mask_words)
      a.attempt_mask[(size_t)e * a.mask_words + att_idx] |= att_word;
    if (MASKS && a.ret_snapshot && snapped) a.ret_snapshot[e] = snap;
    if (fin && a.last_return && !D1_FIN(hot.y)) a.last_return[e] = ret_total;
  }
}

// ----------------------------------------------------------------------------------------
// whole-episode rollout with the posterior-mean reward: k_rollout64's day loop around k_posterior_mean_v's reward
// ----------------------------------------------------------------------------------------
// One workgroup = one tile of the posterior-mean kernel (<= PMV_THREADS envs of ONE coefficient column, lane = env).
// The column's coefficient block is staged in LDS once per launch instead of once per day, the per-day pre-pass and
// the separate step kernel disappear, state stays in registers. Per day: policy, budget gate, run-time fields, the
// env's feature row, the baseline pass over all draws, the effectiveness phase shared by all waves (see
// k_posterior_mean_v), reward, state update. Needs n_samples <= W2A_PMV_NPAD (the host loop of w2a_policy_actions +
// w2a_posterior_mean_reward + w2a_step serves larger draw counts).
struct PmRolloutArgs {
  RolloutArgs r;
  const uint32_t *perm;   // env ids sorted by coefficient column
  const uint4 *tiles;
  const uint32_t *n_tiles;
  const double *wd;
};

template <int KS, int KIND, bool MASKS>
__global__ __launch_bounds__(PMV_THREADS, 4) void k_pm_rollout(const PmRolloutArgs pa) {
  const RolloutArgs &a = pa.r;
  constexpr int WAVES = PMV_THREADS / 64;
  __shared__ uint32_t s_wga[WAVES];
  __shared__ double sW[W2A_PMV_NPAD][2][ROWF];
  __shared__ float4 s_ax[64][KS];
  __shared__ double s_part[WAVES][64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  const uint32_t n_tiles = *pa.n_tiles, per_xcd = (n_tiles + 7u) >> 3;
  const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tile >= n_tiles) return;
  const uint4 tl = pa.tiles[tile];
  const uint32_t col = __builtin_amdgcn_readfirstlane(tl.z);
  const int rows = (int)__builtin_amdgcn_readfirstlane(tl.y);
  const int n_samples = a.tb.n_samples;
  const bool valid = tid < rows;
  const uint32_t e = pa.perm[tl.x + (valid ? tid : 0)];
  uint4 c2, hot;
  load_step_state(a.st, e, c2, hot);
  const uint4 cold = load_cold(a.st, e);
  uint32_t t = D0_T(hot.x), used = D0_USED(hot.x), streak = D0_STREAK(hot.x), last = D0_LAST(hot.x);
  uint32_t atb = D0_ATB(hot.x), hist = D1_HIST(hot.y);
  const uint32_t ndays = D1_NDAYS(hot.y);
  const int32_t budget = (int32_t)hot.w;
  bool fin = D1_FIN(hot.y) != 0;
  float ret_total = __uint_as_float(hot.z);
  const uint32_t rows_per_day = (uint32_t)(a.tb.S_w * a.tb.Y);
  // the column's coefficient block, once per launch
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(pa.wd + (size_t)col * n_samples * (2 * ROWF));
    uint4 *dst = reinterpret_cast<uint4 *>(&sW[0][0][0]);
    for (int idx = tid; idx < n_samples * (2 * ROWF / 2); idx += PMV_THREADS) dst[idx] = src[idx];
  }
  const float *Xf = reinterpret_cast<const float *>(a.tb.X);
  const uint64_t pstream = rng_stream(a.pol.seed ^ 0xA5A5A5A55A5A5A5Aull, (uint64_t)(a.gid0 + e), cold.w);
  float ret = 0.0f;
  int32_t alerts = 0, over = 0;
  uint32_t mask_word = 0, mask_idx = 0xFFFFFFFFu;
  uint32_t att_word = 0, att_idx = 0xFFFFFFFFu;
  float snap = 0.0f;
  bool snapped = false;
  float feat = 0.0f;
  if (KIND == W2A_POLICY_THRESHOLD)
    feat = Xf[((size_t)((a.pol.obs_lag && t > 0 ? t - 1 : t) * rows_per_day + cold.x)) * ROWF + a.pol_slot];
  bool active = !fin && valid;
  for (int s = 0; s < a.n_steps; ++s) {
    if (!__syncthreads_or(active ? 1 : 0)) break;  // also: sW is staged, last day's LDS reads are done
    const int32_t act = policy_action(KIND, a.pol, pstream, t, budget - (int32_t)used, feat);
    // ---- env.py:242-250
    const uint32_t atb_s = ((int32_t)used == budget) ? 1u : 0u;
    const uint32_t actual = (act == 1 && atb_s) ? 0u : (uint32_t)act;
    const uint32_t used2 = used + actual;
    const uint32_t hist2 = ((hist << 1) | actual) & 0x3FFFu;
    const uint32_t day_row = t * rows_per_day + cold.x;
    float4 xf[KS];
    {
      const float4 *xp = a.tb.X + (size_t)day_row * (ROWF / 4);
#pragma unroll
      for (int q = 0; q < KS; ++q) xf[q] = xp[q];
      xf[RT_QUAD] = make_float4((t > 0) ? (float)actual : 0.0f, (float)streak, (float)(budget - (int32_t)used2),
                                (float)__popc(hist2));
    }
    const float today = (KIND == W2A_POLICY_THRESHOLD) ? Xf[(size_t)day_row * ROWF + a.pol_slot] : 0.0f;
    // effectiveness enters through eff * gate * actual (env.py:218-221); inactive rows take no part
    const uint32_t ga = (active && actual && Xf[(size_t)day_row * ROWF + 30] > 0.5f) ? 1u : 0u;
    const uint64_t bal = __ballot(ga != 0);
    if (lane == 0) s_wga[wave] = (uint32_t)__popcll(bal);
    double contrib = 0.0;
    if (64 * wave < rows) {  // baseline pass: the wave's own rows, every draw
      double ax[4 * KS];
      pmv_widen<KS>(xf, ax);
      contrib = pmv_draws<KS>(sW, n_samples, l15, ax);
    }
    __syncthreads();
    uint32_t G = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    int n_eff = 0;
    for (int w = 0; w < WAVES; ++w) {
      const uint32_t c = s_wga[w];
      G += w < wave ? c : 0u;
      n_eff += (int)c;
    }
    n_eff = __builtin_amdgcn_readfirstlane(n_eff);
    for (int grp = 0; grp < n_eff; grp += 64) {  // effectiveness phase, as in k_posterior_mean_v
      const int cnt = min(64, n_eff - grp);
      const int j_own = (int)G - grp;
      const bool own = ga && j_own >= 0 && j_own < 64;
      if (own) {
#pragma unroll
        for (int q = 0; q < KS; ++q) s_ax[j_own][q] = xf[q];
      }
      __syncthreads();
      const int R = (cnt + 15) >> 4, P = R == 1 ? 4 : (R == 2 ? 2 : 1);
      const int drow = lane >> 4;
      const int j = l15 + 16 * (drow % R);
      const int sub = drow / R;
      const int d0 = wave * n_samples / WAVES, d1 = (wave + 1) * n_samples / WAVES;
      double part = 0.0;
      {
        float4 ef[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) ef[q] = s_ax[j < cnt ? j : 0][q];
        double ax[4 * KS];
        pmv_widen<KS>(ef, ax);
#pragma unroll 1
        for (int dd = d0; dd < d1; dd += P) {
          const int sl = dd + sub;
          const float tt = pmv_term<KS, true>(sW, sl < d1 ? sl : d1 - 1, l15, ax);
          part += (sl < d1 && sub < P) ? (double)tt : 0.0;
        }
      }
      s_part[wave][lane] = part;
      __syncthreads();
      if (own) {
        double tsum = 0.0;
#pragma unroll 1
        for (int w = 0; w < WAVES; ++w)
#pragma unroll 1
          for (int p = 0; p < P; ++p) tsum += s_part[w][(j_own & 15) + 16 * ((j_own >> 4) + R * p)];
        contrib = tsum;
      }
    }
    const float r = (float)(-(1000.0 / 152.0) * contrib / (double)n_samples);
    if (active) {
      const bool done = (t + 1 >= ndays);
      ret += r;
      ret_total += r;
      alerts += (int32_t)actual;
      over += (act == 1 && atb_s) ? 1 : 0;
      if (MASKS && a.alert_mask && actual) {
        const uint32_t wi = t >> 5;
        if (wi != mask_idx) {
          if (mask_idx != 0xFFFFFFFFu && mask_idx < (uint32_t)a.mask_words)
            a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
          mask_idx = wi;
          mask_word = 0;
        }
        mask_word |= 1u << (t & 31);
      }
      if (MASKS && a.attempt_mask && act == 1) {
        const uint32_t wi = t >> 5;
        if (wi != att_idx) {
          if (att_idx != 0xFFFFFFFFu && att_idx < (uint32_t)a.mask_words)
            a.attempt_mask[(size_t)e * a.mask_words + att_idx] |= att_word;
          att_idx = wi;
          att_word = 0;
        }
        att_word |= 1u << (t & 31);
      }
      if (MASKS && (done ? t : t + 1) + 2 == ndays) { snap = ret_total; snapped = true; }
      used = used2; hist = hist2; last = actual; atb = atb_s;
      if (!done) { streak = actual ? streak + 1 : 0; t = t + 1; }
      else { fin = true; active = false; }
      feat = a.pol.obs_lag ? today : feat;
    }
    if (KIND == W2A_POLICY_THRESHOLD && !a.pol.obs_lag && active)
      feat = Xf[(size_t)(t * rows_per_day + cold.x) * ROWF + a.pol_slot];
  }
  if (valid) {
    store_hot(a.st, e, make_uint4(pack_d0(t, used, streak, last, atb), pack_d1(hist, ndays, fin ? 1u : 0u),
                                  __float_as_uint(ret_total), (uint32_t)budget));
    if (a.ret_out) a.ret_out[e] = ret;
    if (a.alerts_out) a.alerts_out[e] = alerts;
    if (a.attempts_over_budget) a.attempts_over_budget[e] = over;
    if (MASKS && a.alert_mask && mask_idx != 0xFFFFFFFFu && mask_idx < (uint32_t)a.mask_words)
      a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
    if (MASKS && a.attempt_mask && att_idx != 0xFFFFFFFFu && att_idx < (uint32_t)a.mask_words)
      a.attempt_mask[(size_t)e * a.mask_words + att_idx] |= att_word;
    if (MASKS && a.ret_snapshot && snapped) a.ret_snapshot[e] = snap;
    if (fin && a.last_return && !D1_FIN(hot.y)) a.last_return[e] = ret_total;
  }
}

// ----------------------------------------------------------------------------------------
// one day of a built-in policy: the action of every env from its pre-step state (for policy loops whose step is
// not k_rollout's: reward_mode = "posterior_mean")
// ----------------------------------------------------------------------------------------
// Same policy, same lagging observation, same budget gate (env.py:242-246) as k_rollout; one thread per env. Also keeps
// the per-env rollout counters: alerts issued, alerts attempted at budget, the alert / attempt day bitmaps.
struct PolicyArgs {
  DevTables tb;
  StateArrays st;
  int64_t n;
  int64_t gid0;
  w2a_policy pol;
  int32_t pol_slot;
  int32_t *actions;               // [n] out
  int32_t *alerts;                // [n] += alert issued today (nullable)
  int32_t *attempts_over_budget;  // [n] += alert attempted at budget (nullable)
  uint32_t *alert_mask;           // [n][mask_words] |= bit t (nullable)
  uint32_t *attempt_mask;
  int32_t mask_words;
};

__global__ void k_policy_actions(const PolicyArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t e = (uint32_t)i;
  uint4 c2, hot;
  load_step_state(a.st, e, c2, hot);
  const uint4 cold = load_cold(a.st, e);
  const uint32_t t = D0_T(hot.x), used = D0_USED(hot.x);
  const int32_t budget = (int32_t)hot.w;
  int32_t act = 0;
  if (!D1_FIN(hot.y)) {
    float feat = 0.0f;
    if (a.pol.kind == W2A_POLICY_THRESHOLD) {
      const uint32_t rows_per_day = (uint32_t)(a.tb.S_w * a.tb.Y);
      const uint32_t tt = (a.pol.obs_lag && t > 0) ? t - 1 : t;  // the lagging observation (Q6)
      feat = reinterpret_cast<const float *>(a.tb.X)[((size_t)tt * rows_per_day + cold.x) * ROWF + a.pol_slot];
    }
    const uint64_t pstream = rng_stream(a.pol.seed ^ 0xA5A5A5A55A5A5A5Aull, (uint64_t)(a.gid0 + e), cold.w);
    act = policy_action(a.pol.kind, a.pol, pstream, t, budget - (int32_t)used, feat);
    const bool atb = (int32_t)used == budget;
    const bool actual = act == 1 && !atb;
    if (a.alerts && actual) a.alerts[e] += 1;
    if (a.attempts_over_budget && act == 1 && atb) a.attempts_over_budget[e] += 1;
    if ((t >> 5) < (uint32_t)a.mask_words) {
      if (a.alert_mask && actual) a.alert_mask[(size_t)e * a.mask_words + (t >> 5)] |= 1u << (t & 31);
      if (a.attempt_mask && act == 1) a.attempt_mask[(size_t)e * a.mask_words + (t >> 5)] |= 1u << (t & 31);
    }
  }
  a.actions[e] = act;
}

#endif  // W2A_ROLLOUT_HIP_H
