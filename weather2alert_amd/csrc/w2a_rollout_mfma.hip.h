// w2a_rollout_mfma.hip.h -- k_rollout_mfma: the on-device policy rollout (sampled reward, env.py:238-262 per day) with the
// table-sourced part of both logits on the int8 matrix cores.
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_ROLLOUT_MFMA_HIP_H
#define W2A_ROLLOUT_MFMA_HIP_H

// k_rollout64 spends three quarters of its vector instructions on the two 30-term fp64 dot products of every env-day.
// 27 of the 30 terms do not depend on what the agent did: for the envs of one (county, year) -- which the visiting order
// (w2a_rollout_order) already puts side by side -- they are
//      Z[env][day][head] = sum_{k table-sourced, bias} W[env's (column, draw)][head][k] * X[(county, year)][day][k],
// a GEMM with M = envs of the feature row, N = days, K = 32 slots. Only alert_lag1, alert_streak and remaining_budget
// (slots 24..26) follow the agent's actions; they are added per day in fp64. The GEMM runs on v_mfma_i32_16x16x64_i8 with
// the exact fixed-point digits of w2a_posterior_i8.hip.h -- here the ENV rows carry the coefficient digits (A operands
// P = (W0 | W1), Q = (W2 | W3), gathered once per launch from a digit table of W built once per handle; the run-time slots
// carry no digits -- they do not inflate the row's scale, and their words hold the row's scale, exact flag and run-time
// coefficients instead, see k_rm_wq) and the DAY columns carry the feature digits (B_m = (X_m | X_{m-1}), converted per
// 16-day chunk from the float32 rows of the (county, year)). Same six MFMAs per 16 x 16 tile, same a-priori bound
// |dz| <= 1.5 * 2^(ew - 23.4); coefficient rows outside the fixed-point range (a negative scale) make their env compute the
// plain fp64 dot product per day instead (per lane, rare).
//
// One wave = FOUR SUBTILES of <= 16 envs, each of ONE feature row (subtile list of w2a_rollout_mfma_prepare: the envs of a
// (county, year) in runs of 16, the last one partial) = the four 16-row tiles of the GEMM's M dimension; lock step required
// (every env on the same day: the handle's bookkeeping, else k_rollout64 serves the call). Until round 6 a wave was one
// tile of <= 64 envs of one feature row: at 1 M envs over 8 206 (county, year) rows -- 128 envs per row on average -- that
// left a third of the lanes idle (24 592 waves for 16 384 x 64 envs) in a kernel whose cost is per wave-day; padding to
// 16 instead of 64 wastes 11 % (18 440 waves). Per 16-day chunk and subtile: the feature digits of its row's 16 days -> the wave's LDS
// (128 (day, slot group) conversions; skipped when the subtile continues the previous one's row, which is the common
// case: a row's ~8 subtiles sit in consecutive waves), B operands from there, 2 heads x 6 MFMAs, the int32 sums -> f32
// logit parts -> LDS [env][day][head]; then 16 days of k_rollout64's day loop with 3 + 3 fp64 FMAs in place of 30 + 30.
// Outputs, RNG streams and state are those of k_rollout64 (indexed by env id).
#ifndef RM_WAVES
#define RM_WAVES 1  // waves (= tiles) per workgroup: no workgroup-level synchronisation is used, and single-wave workgroups
                    // leave the dispatcher the finest grain (measured 0.87 / 0.88 / 0.90 ms per episode for 1 / 2 / 4)
#endif
#define RM_ZSTRIDE 34  // floats per env of the logit-part image: 16 days x 2 heads + 2 pad (8-B aligned pairs; lane = env reads of a
                       // {baseline, effectiveness} pair are bank-conflict free)

struct RmArgs {
  RolloutArgs r;
  const uint4 *tiles;       // (first position in the visiting order, envs, feature row, 0)
  const uint32_t *n_tiles;
  const uint32_t *wq;       // [S * n_samples * 2][32] int8 digit planes of W; the words of the run-time slots carry the row's
                            // scale, exact-path flag and run-time coefficients (k_rm_wq)
  const float *xs;          // [64] slot scales
};

// coefficient rows -> digit planes for this kernel, scale in natural units. The run-time slots 24..27 take no part in the
// GEMM (their features follow the agent's actions; the kernel zeroes the feature digits of that quad), so word 6 of each
// of a row's four planes is free -- and carries what the env's lane needs of the row besides the digits:
//   plane 0: the row's scale 2^(ew - 20) as f32 bits, NEGATIVE if the row takes the exact path (a head outside the
//            fixed-point range, or a coefficient on slot 27)          planes 1..3: the f32 coefficients of slots 24, 25, 26
// They arrive with the A-operand loads the wave does anyway and are handed to the env's lane by lane shuffles. Until round
// 6 the lane fetched them itself -- scale, flag and the two 16-B pieces of the float rows: four more cache lines per env,
// by env id, in a launch whose fixed cost is exactly such lines (profiles/r06/exp_rollout_nsteps.log: 200 of 470 us).
__global__ void k_rm_wq(const float *W, const float *xs, int64_t rows, uint32_t *wq) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float w[ROWF];
  float m = 0.0f;
#pragma unroll
  for (int k = 0; k < ROWF; ++k) {
    w[k] = (k >= 24 && k <= 27) ? 0.0f : W[r * ROWF + k] * xs[ROWF + k];
    m = fmaxf(m, fabsf(w[k]));
  }
  int ew = 0;
  if (m > 0.0f) (void)frexpf(m, &ew);
  // a coefficient on slot 27 (the agent's 14-day count: none in the faithful semantics, Q1 -- alert_2wks is an appended
  // observation key, not a reward feature) is honoured by the exact path, so that the common path carries three run-time
  // terms per head instead of four
  const bool exact = ew > W2A_PI8_EW_MAX || W[r * ROWF + 27] != 0.0f;
  const float s = ldexpf(1.0f, 30 - ew);
  uint32_t planes[4][8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    uint32_t d[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) d[c] = pi8_digits(__float2int_rn(w[4 * g + c] * s));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int sh = 8 * (3 - p);
      planes[p][g] = ((d[0] >> sh) & 255u) | (((d[1] >> sh) & 255u) << 8) | (((d[2] >> sh) & 255u) << 16) | (((d[3] >> sh) & 255u) << 24);
    }
  }
  const float scale = ldexpf(1.0f, ew - 20);  // z = scale * (A0 2^8 + A1 + (A2 2^8 + A3) 2^-16)
  planes[0][RT_QUAD] = __float_as_uint(exact ? -scale : scale);
  planes[1][RT_QUAD] = __float_as_uint(W[r * ROWF + 24]);
  planes[2][RT_QUAD] = __float_as_uint(W[r * ROWF + 25]);
  planes[3][RT_QUAD] = __float_as_uint(W[r * ROWF + 26]);
  uint4 *dst = reinterpret_cast<uint4 *>(wq + r * ROWF);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    dst[2 * p] = make_uint4(planes[p][0], planes[p][1], planes[p][2], planes[p][3]);
    dst[2 * p + 1] = make_uint4(planes[p][4], planes[p][5], planes[p][6], planes[p][7]);
  }
}
// the subtile list: <= 16 consecutive positions of the visiting order that share one feature row; one thread per row, from
// the row starts and subtile starts the order's scan left (k_order_scan). tiles[j] = (first position, envs, feature row, 0)
#define RM_SUB 16u
__global__ void k_rm_tiles(const uint32_t *start, const uint32_t *tile_start, int32_t rows, uint4 *tiles, uint32_t *n_tiles) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r == 0) *n_tiles = tile_start[rows];
  if (r >= rows) return;
  const uint32_t s0 = start[r], s1 = start[r + 1];
  uint32_t j = tile_start[r];
  for (uint32_t st = s0; st < s1; st += RM_SUB) tiles[j++] = make_uint4(st, min(RM_SUB, s1 - st), (uint32_t)r, 0u);
}

// Compiled per policy kind and with / without the day bitmaps + return snapshot (like k_rollout64): with two waves per
// SIMD resident the day loop is bound by the latency of its own instruction chain, and those wave-uniform choices
// otherwise cost a dozen scalar branches per day.
#ifndef RM_MIN_WAVES
#define RM_MIN_WAVES 3  // waves per SIMD the kernel is compiled for (<= 168 VGPRs: 156-168 by policy kind, no spills)
#endif
template <int KIND, bool MASKS>
__global__ __launch_bounds__(64 * RM_WAVES, RM_MIN_WAVES) void k_rollout_mfma(const RmArgs ra) {
  const RolloutArgs &a = ra.r;
  __shared__ __attribute__((aligned(16))) uint32_t sXd[RM_WAVES][16][PI8_XSTRIDE];  // feature digits of the chunk's days
  __shared__ __attribute__((aligned(16))) float sZ[RM_WAVES][64][RM_ZSTRIDE];                                      // logit parts [env][day * 2 + head]
  __shared__ float sDay[RM_WAVES][4][16][2];  // per subtile and day of the chunk: gate flag (slot 30), the threshold policy's feature
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, c16 = lane & 15;
  // The grid is sized for the worst case (n / 64 + feature rows tiles); the tiles there are get an eighth per XCD, dealt
  // from the actual count -- with the grid's eighths the surplus workgroups would all fall on the last XCDs and leave
  // them idle (at 1 M envs / 8 206 rows: 20.5 K tiles of 24.6 K, XCD 7 empty and XCD 6 a third full: 1.03 -> 0.90 ms)
  const uint32_t n_sub = *ra.n_tiles;           // subtiles of <= 16 envs
  const uint32_t n_tiles = (n_sub + 3u) >> 2;   // waves
  const uint32_t n_wgs = (n_tiles + RM_WAVES - 1) / RM_WAVES, per_xcd = (n_wgs + 7u) >> 3;
  if ((blockIdx.x >> 3) >= per_xcd) return;
  const uint32_t tile = ((blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3)) * RM_WAVES + wave;
  if (tile >= n_tiles) return;  // whole wave; only wave-level synchronisation below
  const int subs = (int)min(4u, n_sub - 4u * tile);  // subtiles of this wave (4 but for the last wave); wave-uniform
  // lane (q, c16) = env c16 of subtile q; lanes past a subtile's end (and past the last subtile) shadow a valid env
  const uint4 tl = ra.tiles[min(4u * tile + (uint32_t)q, n_sub - 1u)];
  const uint32_t frow = tl.z;  // this lane's (county, year): up to four different ones per wave
  const bool valid = q < subs && (uint32_t)c16 < tl.y;
  const uint32_t e = a.order[tl.x + (valid ? (uint32_t)c16 : 0u)];
  uint4 c2, hot;
  load_step_state(a.st, e, c2, hot);  // c2 = {ep_row, ep_w}: the copies in stepc
  // the episode number keys the Bernoulli policy's stream; no other policy needs the env's cold record (one line per env)
  const uint32_t episode_no = KIND == W2A_POLICY_BERNOULLI ? load_cold(a.st, e).w : 0u;
  uint32_t t = D0_T(hot.x), used = D0_USED(hot.x), streak = D0_STREAK(hot.x), last = D0_LAST(hot.x);
  uint32_t atb = D0_ATB(hot.x), hist = D1_HIST(hot.y);
  const uint32_t ndays = D1_NDAYS(hot.y);
  const int32_t budget = (int32_t)hot.w;
  bool fin = D1_FIN(hot.y) != 0;
  float ret_total = __uint_as_float(hot.z);
  const uint32_t rows_per_day = (uint32_t)(a.tb.S_w * a.tb.Y);
  const uint32_t wrow = W_COL(c2.y) * (uint32_t)a.tb.n_samples + W_SAMPLE(c2.y);
  const float *Wf = reinterpret_cast<const float *>(a.tb.W) + (size_t)wrow * (2 * ROWF);  // (read by the exact path only)
  // A operands: row tile m, lane (c16, q) holds 16 slots of one plane of env 16 m + c16 (its coefficient digits): 64
  // registers for the whole launch
  pi8_v4i P[4][2], Q[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const uint32_t wr = (uint32_t)__shfl((int)wrow, 16 * m + c16);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t *row = ra.wq + ((size_t)wr * 2 + h) * ROWF;
      P[m][h] = *reinterpret_cast<const pi8_v4i *>(row + 4 * q);
      Q[m][h] = *reinterpret_cast<const pi8_v4i *>(row + 16 + 4 * q);
    }
  }
  // What rides in the run-time slot words of the digit rows (k_rm_wq): word 6 of plane 0 / 1 sits in component 2 of the P
  // fragment of lane group q = 1 / 3, word 6 of plane 2 / 3 in component 2 of the Q fragment of lane group 1 / 3 -- of the
  // lane with the env's c16. Each env's own lane (row tile m = its q) takes them by lane shuffles:
  //   the two row scales 2^(ew - 20), powers of two, applied by the env's lane in the day loop (exact), so that the
  //   matrix-core section needs no per-row scale; negative = the row takes the exact path; the three run-time
  //   coefficients of both heads (slots 24, 25, 26) as doubles for the whole launch
  float sc_b = 0.0f, sc_e = 0.0f;
  double wl_b = 0.0, ws_b = 0.0, wr_b = 0.0, wl_e = 0.0, ws_e = 0.0, wr_e = 0.0;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float s0 = __int_as_float(__shfl(P[m][0][2], 16 + c16)), s1 = __int_as_float(__shfl(P[m][1][2], 16 + c16));
    const float l0 = __int_as_float(__shfl(P[m][0][2], 48 + c16)), l1 = __int_as_float(__shfl(P[m][1][2], 48 + c16));
    const float k0 = __int_as_float(__shfl(Q[m][0][2], 16 + c16)), k1 = __int_as_float(__shfl(Q[m][1][2], 16 + c16));
    const float r0 = __int_as_float(__shfl(Q[m][0][2], 48 + c16)), r1 = __int_as_float(__shfl(Q[m][1][2], 48 + c16));
    if (q == m) { sc_b = s0; sc_e = s1; wl_b = l0; wl_e = l1; ws_b = k0; ws_e = k1; wr_b = r0; wr_e = r1; }
  }
  const bool exact = sc_b < 0.0f || sc_e < 0.0f;  // this env's coefficient row is outside the fixed-point range (or uses slot 27)
  const bool any_exact = __any(exact) != 0;       // wave-uniform: the common case never enters the exact path's control flow
  // The day loop works on -log2(e) x logit, what v_exp_f32 wants (as k_posterior_mean_i8 does): the factor rides in the row
  // scales (a power of two times the constant: one rounding, where the plain form rounds the product with the constant)
  // and in the fp64 run-time coefficients, instead of two multiplies per env-day
  constexpr float NL2E = -1.44269504088896340736f;
  constexpr double NL2E_D = -1.44269504088896340736;
  sc_b = fabsf(sc_b) * NL2E; sc_e = fabsf(sc_e) * NL2E;
  wl_b *= NL2E_D; ws_b *= NL2E_D; wr_b *= NL2E_D; wl_e *= NL2E_D; ws_e *= NL2E_D; wr_e *= NL2E_D;
  const float *Xf = reinterpret_cast<const float *>(a.tb.X);
  const uint64_t pstream = rng_stream(a.pol.seed ^ 0xA5A5A5A55A5A5A5Aull, (uint64_t)(a.gid0 + e), episode_no);
  constexpr int32_t kind = KIND;
  float ret = 0.0f;
  int32_t over = 0;
  uint32_t mask_word = 0, att_word = 0;
  float snap = 0.0f;
  bool snapped = false;
  // Lock step (the handle's bookkeeping; else k_rollout64 serves the call): every env of the batch is on the same day of an
  // episode of the one length there is, finished or not together. So the day, the number of days this call runs, "today
  // is the terminal day" and the word of the day bitmaps are WAVE-UNIFORM: they live in scalar registers, and the day loop
  // has a fixed trip count and no per-lane control flow (round 6; until then every day re-derived them per lane and the
  // compiler kept three nested exec-mask conditions alive through the loop: ~35 scalar mask instructions and 8 branches
  // per day beside ~75 vector ones).
  const uint32_t t_first = __builtin_amdgcn_readfirstlane(t);
  const uint32_t nd_u = __builtin_amdgcn_readfirstlane(ndays);
  const bool fin_u = __builtin_amdgcn_readfirstlane((uint32_t)fin) != 0u;
  int total = fin_u ? 0 : min(a.n_steps, (int)(nd_u - t_first));  // days this call runs
  const bool ends = !fin_u && total == (int)(nd_u - t_first);     // ... the terminal one among them
  uint32_t tu = t_first;                                          // today
  // threshold policy: what the agent sees on day d is the row of day d - 1 (the lagging observation, Q6; day 0 and
  // obs_lag = 0: today's). The chunk's per-day LDS entries hold THAT row's feature, so the day loop carries nothing over
  const uint32_t lag = (kind == W2A_POLICY_THRESHOLD && a.pol.obs_lag) ? 1u : 0u;  // wave-uniform
  const bool req = a.pol.require_budget != 0;                                       // wave-uniform
  const uint32_t used0 = used;
  int32_t rem = budget - (int32_t)used;  // what remains of the budget (negative for a negative budget: never "at budget")
  const bool ran = total > 0;
  for (uint32_t c0 = t_first; total > 0; c0 += 16) {
    // ---- what the day loop reads of the float rows themselves -- the gate flag (slot 30) and the threshold policy's
    // feature --: lane (q, c16) fetches day c0 + c16 of ITS subtile's row (the subtile's envs share the row); requested
    // here, stored to LDS behind the matrix-core section, whose work covers their latency
    float pd_gate, pd_feat = 0.0f;
    {
      const uint32_t d = min(c0 + (uint32_t)c16, (uint32_t)a.tb.T - 1u);
      pd_gate = Xf[((size_t)d * rows_per_day + frow) * ROWF + 4 * GATE_QUAD + 2];
      if (kind == W2A_POLICY_THRESHOLD) pd_feat = Xf[((size_t)(d - min(d, lag)) * rows_per_day + frow) * ROWF + a.pol_slot];
    }
    // ---- per subtile m (= 16-row tile of the GEMM): the B operands of its row's 16 days, then 2 heads x 6 MFMAs
    pi8_v4i B[4];
    uint32_t row_in_lds = 0xFFFFFFFFu;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (m >= subs) continue;  // wave-uniform
      const uint32_t fr = (uint32_t)__builtin_amdgcn_readlane((int)frow, 16 * m);
      if (fr != row_in_lds) {  // wave-uniform: another (county, year) than the previous subtile's
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // (the previous subtile's B operands have been read)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // feature digits of days c0 .. c0 + 15 of that row: lane = (day, two slot groups). (Requesting the first row a
        // chunk ahead, in front of the previous chunk's day loop, was measured: 516 / 498 us against 480 / 468 without --
        // its 8 registers cost more than the hidden latency gains; profiles/r06/ab_rollout_subtiles.log)
        const int j = lane >> 2, g0 = (lane & 3) * 2;
        const uint32_t day = min(c0 + (uint32_t)j, (uint32_t)a.tb.T - 1u);
        const float4 *xp = a.tb.X + ((size_t)day * rows_per_day + fr) * (ROWF / 4);
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
          const int g = g0 + gg;
          const float4 v = xp[g];
          uint32_t d[4], o[4];
          d[0] = pi8_digits((int32_t)(v.x * ra.xs[4 * g]));
          d[1] = pi8_digits((int32_t)(v.y * ra.xs[4 * g + 1]));
          d[2] = pi8_digits((int32_t)(v.z * ra.xs[4 * g + 2]));
          d[3] = pi8_digits((int32_t)(v.w * ra.xs[4 * g + 3]));
          pi8_planes(d, o);
          // (the run-time slots' quad: no feature digits -- the coefficient rows carry other data in those words)
#pragma unroll
          for (int p = 0; p < 4; ++p) sXd[wave][j][8 * p + g] = g == RT_QUAD ? 0u : o[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // B operands (day column c16): B_i = (X_i | X_{i-1})
        const uint32_t *x = sXd[wave][c16];
        const int half = 4 * (q & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int plane = q < 2 ? i : i - 1;
          const pi8_v4i v = *reinterpret_cast<const pi8_v4i *>(x + 8 * max(plane, 0) + half);
          B[i] = plane >= 0 ? v : pi8_v4i{0, 0, 0, 0};
        }
        row_in_lds = fr;
      }
      float z0[4];  // head 0's logit parts of the four accumulator rows, kept until head 1's are there: one 8-B store per row
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const pi8_v4i zero = {0, 0, 0, 0};
        const pi8_v4i Pm = P[m][h], Qm = Q[m][h];
        pi8_v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Pm, B[0], zero, 0, 0, 0);
        pi8_v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Pm, B[1], zero, 0, 0, 0);
        pi8_v4i a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Pm, B[2], zero, 0, 0, 0);
        pi8_v4i a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Pm, B[3], zero, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qm, B[0], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qm, B[1], a3, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {  // accumulator row 4 q + j of the row tile = env, column c16 = day
          const int er = 16 * m + 4 * q + j;
          const float u = (float)(a0[j] * 256 + a1[j]);
          const float v = (float)(a2[j] * 256 + a3[j]);
          const float z = fmaf(v, 1.52587890625e-05f, u);  // x the row's scale (a power of two: exact) in the day loop
          if (h == 0) z0[j] = z;
          else *reinterpret_cast<float2 *>(&sZ[wave][er][c16 * 2]) = make_float2(z0[j], z);
        }
      }
    }
    sDay[wave][q][c16][0] = pd_gate;
    if (kind == W2A_POLICY_THRESHOLD) sDay[wave][q][c16][1] = pd_feat;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- the chunk's days, lane = env (k_rollout64's day loop with the table part of the logits taken from sZ).
    // The body exists four times, chosen by two wave-uniform facts, so that none of them costs vector instructions per day:
    //   REQ    the policy never alerts without budget (require_budget: act = 0 when nothing remains). Then an attempt at
    //          budget cannot happen: actual = act, no "at budget" compare, no over-budget count (it stays 0 exactly);
    //   FIRST  day 0 of the episode (only the first day of the first chunk can be): alert_lag1 = 0 whatever the action.
    // The budget is carried as what REMAINS (remaining_budget is a reward feature; "at budget" is remaining == 0).
    const int chunk = min(16, total);
    auto day = [&](auto req_c, auto first_c, const int dd) {
      constexpr bool REQ = decltype(req_c)::value, FIRST = decltype(first_c)::value;
      const float seen = sDay[wave][q][dd][1], gate = sDay[wave][q][dd][0];
      const int32_t act = policy_action(kind, a.pol, pstream, tu, rem, seen);
      uint32_t atb_s = 0u, actual = (uint32_t)act;
      if (!REQ) {
        atb_s = (rem == 0) ? 1u : 0u;
        actual = (act == 1 && atb_s) ? 0u : (uint32_t)act;
      }
      const int32_t rem2 = rem - (int32_t)actual;
      const uint32_t hist2 = ((hist << 1) | actual) & 0x3FFFu;
      const double f_lag = FIRST ? 0.0 : (double)actual, f_streak = (double)streak;
      const double f_rem = (double)rem2;
      const float2 zt = *reinterpret_cast<const float2 *>(&sZ[wave][lane][dd * 2]);
      double zb = (double)(zt.x * sc_b), ze = (double)(zt.y * sc_e);
      if (any_exact) {   // wave-uniform; rare
        if (exact) {     // plain fp64 dot products for a coefficient row outside the fixed-point range (or with a slot-27 term)
          zb = 0.0; ze = 0.0;
          uint32_t tt = tu;
          asm volatile("" : "+s"(tt));  // (keeps the row index of this rare path out of the day loop's induction variables)
          const float4 *xr = a.tb.X + (size_t)(tt * rows_per_day + frow) * (ROWF / 4);
          const float4 *wr4 = reinterpret_cast<const float4 *>(Wf);
          // one quad of slots per trip, NOT unrolled: unrolled, this rare path held 16 float4 loads in flight and set the
          // register allocation of the whole kernel (the source of its spills until round 6)
#pragma unroll 1
          for (int qd = 0; qd < ROWF / 4; ++qd) {
            if (qd == RT_QUAD) continue;  // slots 24..27: the run-time fields, added below
            const float4 xv = xr[qd], wb = wr4[qd], we = wr4[ROWF / 4 + qd];
            zb = fma((double)xv.x, (double)wb.x, zb); ze = fma((double)xv.x, (double)we.x, ze);
            zb = fma((double)xv.y, (double)wb.y, zb); ze = fma((double)xv.y, (double)we.y, ze);
            zb = fma((double)xv.z, (double)wb.z, zb); ze = fma((double)xv.z, (double)we.z, ze);
            zb = fma((double)xv.w, (double)wb.w, zb); ze = fma((double)xv.w, (double)we.w, ze);
          }
          const double f_a2w = (double)__popc(hist2);
          zb = fma(f_a2w, (double)Wf[27], zb);
          ze = fma(f_a2w, (double)Wf[ROWF + 27], ze);
          zb *= NL2E_D; ze *= NL2E_D;
        }
      }
      if (!FIRST) { zb = fma(f_lag, wl_b, zb); ze = fma(f_lag, wl_e, ze); }
      zb = fma(f_streak, ws_b, zb); zb = fma(f_rem, wr_b, zb);
      ze = fma(f_streak, ws_e, ze); ze = fma(f_rem, wr_e, ze);
      // env.py:211-221 on the negated, log2-scaled logits: sigmoid(z) = 1 / (1 + 2^(-log2(e) z)); a closed gate is
      // logit -inf = +inf here: 2^inf = inf, 1 / inf = 0 exactly (reward_from_logits, w2a_common.hip.h, minus its two multiplies)
      // The effectiveness counts on alert days only (eff x actual, actual in {0, 1}): "no alert" closes the gate as well,
      // which gives the same 1 - 0 bit for bit and spares the env's action as a float.
      float eb = (float)zb, ee = (float)ze;
      if (!(gate > 0.5f && actual)) ee = __builtin_inff();
      const float cb = -(1000.0f / 152.0f) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(eb));
      const float eff = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ee));
      const float r = fmaf(-cb, eff, cb);  // = c x base x (1 - eff), one rounding less
      const bool done = tu + 1 >= nd_u;  // wave-uniform: the last day this call runs, if it runs to the end
      ret += r;
      ret_total += r;
      if (!REQ) over += (act == 1 && atb_s) ? 1 : 0;
      if (MASKS) {
        mask_word |= actual << (tu & 31u);
        att_word |= (uint32_t)(act == 1) << (tu & 31u);
        if ((tu & 31u) == 31u || dd + 1 == total) {  // wave-uniform: the bitmap word of these 32 days is complete (or the call ends)
          const uint32_t wi = tu >> 5;
          if (valid && wi < (uint32_t)a.mask_words) {
            if (a.alert_mask && mask_word) a.alert_mask[(size_t)e * a.mask_words + wi] |= mask_word;
            if (a.attempt_mask && att_word) a.attempt_mask[(size_t)e * a.mask_words + wi] |= att_word;
          }
          mask_word = 0; att_word = 0;
        }
        if ((done ? tu : tu + 1) + 2 == nd_u) { snap = ret_total; snapped = true; }
      }
      rem = rem2; hist = hist2; last = actual;
      if (!REQ) atb = atb_s;
      {  // streak = actual ? streak + 1 : 0 as actual x streak + actual: one instruction (written as a select the compiler
         // emits two); env.py:256-260: the terminal step leaves day and streak as they are
        uint32_t s2;
        asm("v_mad_u32_u24 %0, %1, %2, %1" : "=v"(s2) : "v"(actual), "v"(streak));
        streak = done ? streak : s2;
      }
      ++tu;
    };
    {
      const std::true_type yes;
      const std::false_type no;
      int dd = 0;
      if (tu == 0u) {
        if (req) day(yes, yes, 0); else day(no, yes, 0);
        dd = 1;
      }
      if (req) for (; dd < chunk; ++dd) day(yes, no, dd);
      else for (; dd < chunk; ++dd) day(no, no, dd);
    }
    total -= chunk;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();  // sXd / sZ are rewritten by the next chunk
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (!fin_u) { t = ends ? nd_u - 1u : tu; fin = ends; }
  used = (uint32_t)(budget - rem);
  if (req && ran) atb = (rem + (int32_t)last == 0) ? 1u : 0u;  // "at budget" as the last day's step found it (env.py:243)
  if (valid) {
    store_hot(a.st, e, make_uint4(pack_d0(t, used, streak, last, atb), pack_d1(hist, ndays, fin ? 1u : 0u),
                                  __float_as_uint(ret_total), (uint32_t)budget));
    if (a.ret_out) a.ret_out[e] = ret;
    if (a.alerts_out) a.alerts_out[e] = (int32_t)(used - used0);  // every alert issued is one unit of budget used (env.py:246-250)
    if (a.attempts_over_budget) a.attempts_over_budget[e] = over;
    if (MASKS && a.ret_snapshot && snapped) a.ret_snapshot[e] = snap;
    if (fin && a.last_return && !D1_FIN(hot.y)) a.last_return[e] = ret_total;
  }
}

#endif  // W2A_ROLLOUT_MFMA_HIP_H
