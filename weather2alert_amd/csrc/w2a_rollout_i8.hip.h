// w2a_rollout_i8.hip.h -- k_pm_rollout_i8: whole-episode policy rollout with the posterior-mean reward on the int8 matrix
// cores (k_pm_rollout's day loop around k_posterior_mean_i8's contraction).
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_ROLLOUT_I8_HIP_H
#define W2A_ROLLOUT_I8_HIP_H

// One workgroup = one tile of the int8 kernel (<= 256 envs of ONE coefficient column, lane = env) for the whole episode:
// the column's int8 coefficient block and its scales are staged in LDS ONCE per launch and stay; per-env state lives in
// registers; the per-day pre-pass (k_pm_prep), the per-day restaging of the block and the separate step kernel of the
// host loop disappear. Per day: policy, budget gate and run-time fields per lane (as k_rollout64), the env's feature row
// -> fixed-point digits -> the X image in LDS (rows with an open-gate alert first), A fragments, the tile loop and the
// DPP reduction of k_posterior_mean_i8, reward, state update. Here the X image and the coefficient block are both alive
// all the time (70 KB of LDS: two workgroups per CU). Needs n_samples <= PI8_NPAD; tiles of columns outside the
// fixed-point range run the exact fp64 sum per day.
struct PmI8RolloutArgs {
  RolloutArgs r;
  const uint32_t *perm;      // env ids sorted by coefficient column
  const uint4 *tiles;        // tile list with PI8_ROWS positions per tile
  const uint32_t *n_tiles;
  const uint32_t *wq;
  const float *wscale;
  const uint32_t *colflag;
  const float *xs;
};

__global__ __launch_bounds__(PI8_THREADS, 2) void k_pm_rollout_i8(const PmI8RolloutArgs pa) {
  const RolloutArgs &a = pa.r;
  __shared__ __attribute__((aligned(16))) uint32_t sX[PI8_ROWS][PI8_XSTRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t sW[PI8_NPAD][PI8_WSTRIDE];
  __shared__ float sScale[PI8_NPAD * 2];
  __shared__ __attribute__((aligned(16))) float sGa[PI8_ROWS];
  __shared__ float sSum[PI8_ROWS];
  __shared__ uint32_t s_wga[PI8_WAVES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t n_tiles = *pa.n_tiles, per_xcd = (n_tiles + 7u) >> 3;
  const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tile >= n_tiles) return;
  const uint4 tl = pa.tiles[tile];
  const uint32_t col = __builtin_amdgcn_readfirstlane(tl.z);
  const int rows = (int)__builtin_amdgcn_readfirstlane(tl.y);
  const int n_samples = a.tb.n_samples;  // <= PI8_NPAD (checked by the host)
  const int ntiles = (n_samples + 15) >> 4;
  const bool exact = pa.colflag[col] != 0u;  // workgroup-uniform
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
  if (!exact) {  // the column's int8 block and scales, once per launch
    const uint4 *src = reinterpret_cast<const uint4 *>(pa.wq + (size_t)col * n_samples * (2 * ROWF));
    for (int idx = tid; idx < ntiles * 16 * 16; idx += PI8_THREADS)
      *reinterpret_cast<uint4 *>(&sW[idx >> 4][4 * (idx & 15)]) = idx < n_samples * 16 ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
    for (int idx = tid; idx < ntiles * 32; idx += PI8_THREADS)
      sScale[idx] = idx < n_samples * 2 ? pa.wscale[(size_t)col * n_samples * 2 + idx] : 0.0f;
  }
  const float *Xf = reinterpret_cast<const float *>(a.tb.X);
  const uint64_t pstream = rng_stream(a.pol.seed ^ 0xA5A5A5A55A5A5A5Aull, (uint64_t)(a.gid0 + e), cold.w);
  const int32_t kind = a.pol.kind;
  float ret = 0.0f;
  int32_t alerts = 0, over = 0;
  uint32_t mask_word = 0, mask_idx = 0xFFFFFFFFu;
  uint32_t att_word = 0, att_idx = 0xFFFFFFFFu;
  float snap = 0.0f;
  bool snapped = false;
  float feat = 0.0f;
  if (kind == W2A_POLICY_THRESHOLD)
    feat = Xf[((size_t)((a.pol.obs_lag && t > 0 ? t - 1 : t) * rows_per_day + cold.x)) * ROWF + a.pol_slot];
  bool active = !fin && valid;
  for (int s = 0; s < a.n_steps; ++s) {
    if (!__syncthreads_or(active ? 1 : 0)) break;  // also: the block is staged; yesterday's LDS reads are done
    const int32_t act = policy_action(kind, a.pol, pstream, t, budget - (int32_t)used, feat);
    // ---- env.py:242-250
    const uint32_t atb_s = ((int32_t)used == budget) ? 1u : 0u;
    const uint32_t actual = (act == 1 && atb_s) ? 0u : (uint32_t)act;
    const uint32_t used2 = used + actual;
    const uint32_t hist2 = ((hist << 1) | actual) & 0x3FFFu;
    const uint32_t day_row = t * rows_per_day + cold.x;
    float4 xf[ROWF / 4];
    {
      const float4 *xp = a.tb.X + (size_t)day_row * (ROWF / 4);
#pragma unroll
      for (int q = 0; q < ROWF / 4; ++q) xf[q] = xp[q];
    }
    const float today = (kind == W2A_POLICY_THRESHOLD) ? Xf[(size_t)day_row * ROWF + a.pol_slot] : 0.0f;
    xf[RT_QUAD] = make_float4((t > 0) ? (float)actual : 0.0f, (float)streak, (float)(budget - (int32_t)used2),
                              (float)__popc(hist2));
    // effectiveness enters through eff * gate * actual (env.py:218-221); inactive rows take no part
    const uint32_t ga = (active && actual && xf[GATE_QUAD].z > 0.5f) ? 1u : 0u;
    double contrib;
    if (exact) {
      contrib = active ? pi8_exact_row(xf, a.tb.W, col, n_samples, ga) : 0.0;
    } else {
      // rows with gate * actual = 1 first (stable partition inside the workgroup), as in k_posterior_mean_i8
      const uint64_t bal = __ballot(ga != 0);
      if (lane == 0) s_wga[wave] = (uint32_t)__popcll(bal);
      __syncthreads();
      uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull)), n_eff = 0;
      for (int w = 0; w < PI8_WAVES; ++w) {
        const uint32_t c = s_wga[w];
        before += w < wave ? c : 0u;
        n_eff += c;
      }
      n_eff = __builtin_amdgcn_readfirstlane(n_eff);
      const uint32_t pos = ga ? before : n_eff + ((uint32_t)tid - before);
      sGa[pos] = (float)ga;
      pi8_store_row(xf, pa.xs, sX[pos]);
      __syncthreads();  // the X image of the day is complete
      pi8_v4i P[PI8_MT_PER_WAVE], Q[PI8_MT_PER_WAVE];
      pi8_load_a(sX, wave, lane, P, Q);
      float rs[PI8_MT_PER_WAVE][4];
#pragma unroll
      for (int i = 0; i < PI8_MT_PER_WAVE; ++i) rs[i][0] = rs[i][1] = rs[i][2] = rs[i][3] = 0.0f;
      pi8_accumulate(sW, sScale, sGa, P, Q, rs, ntiles, n_samples, rows, (int)((n_eff + 15u) >> 4), wave, lane);
      pi8_reduce(rs, sSum, wave, lane);
      __syncthreads();
      contrib = (double)sSum[pos];
    }
    const float r = (float)(-(1000.0 / 152.0) * contrib / (double)n_samples);
    if (active) {
      const bool done = (t + 1 >= ndays);
      ret += r;
      ret_total += r;
      alerts += (int32_t)actual;
      over += (act == 1 && atb_s) ? 1 : 0;
      if (a.alert_mask && actual) {
        const uint32_t wi = t >> 5;
        if (wi != mask_idx) {
          if (mask_idx != 0xFFFFFFFFu && mask_idx < (uint32_t)a.mask_words)
            a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
          mask_idx = wi;
          mask_word = 0;
        }
        mask_word |= 1u << (t & 31);
      }
      if (a.attempt_mask && act == 1) {
        const uint32_t wi = t >> 5;
        if (wi != att_idx) {
          if (att_idx != 0xFFFFFFFFu && att_idx < (uint32_t)a.mask_words)
            a.attempt_mask[(size_t)e * a.mask_words + att_idx] |= att_word;
          att_idx = wi;
          att_word = 0;
        }
        att_word |= 1u << (t & 31);
      }
      if ((done ? t : t + 1) + 2 == ndays) { snap = ret_total; snapped = true; }
      used = used2; hist = hist2; last = actual; atb = atb_s;
      if (!done) { streak = actual ? streak + 1 : 0; t = t + 1; }
      else { fin = true; active = false; }
      feat = a.pol.obs_lag ? today : feat;
    }
    if (kind == W2A_POLICY_THRESHOLD && !a.pol.obs_lag && active)
      feat = Xf[(size_t)(t * rows_per_day + cold.x) * ROWF + a.pol_slot];
  }
  if (valid) {
    store_hot(a.st, e, make_uint4(pack_d0(t, used, streak, last, atb), pack_d1(hist, ndays, fin ? 1u : 0u),
                                  __float_as_uint(ret_total), (uint32_t)budget));
    if (a.ret_out) a.ret_out[e] = ret;
    if (a.alerts_out) a.alerts_out[e] = alerts;
    if (a.attempts_over_budget) a.attempts_over_budget[e] = over;
    if (a.alert_mask && mask_idx != 0xFFFFFFFFu && mask_idx < (uint32_t)a.mask_words)
      a.alert_mask[(size_t)e * a.mask_words + mask_idx] |= mask_word;
    if (a.attempt_mask && att_idx != 0xFFFFFFFFu && att_idx < (uint32_t)a.mask_words)
      a.attempt_mask[(size_t)e * a.mask_words + att_idx] |= att_word;
    if (a.ret_snapshot && snapped) a.ret_snapshot[e] = snap;
    if (fin && a.last_return && !D1_FIN(hot.y)) a.last_return[e] = ret_total;
  }
}

#endif  // W2A_ROLLOUT_I8_HIP_H
