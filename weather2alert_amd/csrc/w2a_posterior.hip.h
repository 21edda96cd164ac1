// w2a_posterior.hip.h -- k_posterior_mean: today's reward averaged over ALL posterior draws of the env's
// coefficient column (the legacy env's eval mode, _deprecated/env.py:332-342, on today's linear-logistic form
// env.py:197-226), as a grouped fp64-MFMA GEMM per coefficient column; k_group_keys feeds the grouping sort.
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_POSTERIOR_HIP_H
#define W2A_POSTERIOR_HIP_H

// For env e with coefficient column c:   reward_e = mean_s  -(1000/152) * sigmoid(zb_s) * (1 - sigmoid(ze_s) * gate * actual)
//     zb_s = sum_k x_k * Wb[c][s][k],   ze_s = sum_k x_k * We[c][s][k]          (k over the 32 row slots)
// i.e. per column c one dense contraction  D_c [N_c envs][2 * n_samples] = A_c [N_c][32] * B_c [32][2 * n_samples]
// followed by a sigmoid / product / mean epilogue. Envs are served in the order of `perm` (env ids sorted by
// column, built once per episode by w2a_group_by_column), so a 256-row tile spans one column, sometimes two:
//   * the workgroup stages B_c (both heads, all draws: 25.6 KB of f32) in LDS once per column segment;
//   * each of its 4 waves owns 4 row tiles of 16: A fragments = the env's feature row of the day with the run-time slots
//     patched in (the same derive_day()/runtime_fields() as the step kernel), converted f32 -> f64 in registers;
//   * per 16-draw tile 8 + 8 v_mfma_f64_16x16x4_f64 (K = 32 slots; products of f32 values are exact in fp64, so
//     the logits carry ~1e-16 relative error, as in the step kernels);
//   * epilogue in the accumulator layout (lane = draw column, 4 rows per lane): f32 sigmoids as in the step
//     kernels, closed gate = -inf logit, per-row sums over draws in fp64, 16-lane DPP all-reduce, one f32 per env.
// The step kernel then runs with W2A_STEP_REWARD_GIVEN and does everything else of env.py:238-262.
#define PM_ROWS 256                // sorted positions per workgroup: 16 row tiles of 16, 4 per wave
static_assert(PM_ROWS == BLOCK, "one thread per row in the set-up phase");
#define PM_TILES_PER_WAVE (PM_ROWS / 16 / (BLOCK / 64))
#define PM_NPAD 112                // draws per staging pass (7 MFMA column tiles)
#ifndef W2A_PM_EXPERIMENT
#define W2A_PM_EXPERIMENT 0
#endif
typedef double pm_double4 __attribute__((ext_vector_type(4)));

struct PosteriorArgs {
  DevTables tb;
  StateArrays st;
  const uint32_t *perm;  // [n] env ids sorted by coefficient column
  uint4 *prep;           // [n] per-env record of the day, env order (k_pm_prep -> k_posterior_mean)
  const void *actions;
  int32_t act_dtype;
  float *reward;
  int32_t *status;
  int64_t n;
};

// Per-env record of the day in ENV order (coalesced state / action reads, like phase A of k_step64), so that the
// GEMM kernel, which walks the envs in column order, gathers ONE 16-B record per env instead of three state words
// and a feature-row word:  x = float index of the feature row, y = run-time fields packed (alert_lag1 bit 0,
// alert_streak bits 1..10, alert_2wks bits 11..14, gate * actual bit 15, remaining_budget bits 16..31: budgets
// up to 65535, checked by the host class), z = coefficient column.
__global__ void k_pm_prep(const PosteriorArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t e = (uint32_t)i;
  const u3 h = a.st.hot3[e];
  const u3 c = a.st.stepc[e];
  const Day d = derive_day(h, c, load_action_raw(a.actions, a.act_dtype, e));
  const uint32_t xrow = (d.t * (uint32_t)(a.tb.S_w * a.tb.Y) + c.b) * ROWF;
  const float4 rt = runtime_fields(d);
  // effectiveness enters through eff * gate * actual (env.py:218-221): slot 30 of the row is the 0/1 gate flag
  const uint32_t ga = (d.actual && reinterpret_cast<const float *>(a.tb.X)[xrow + 30] > 0.5f) ? 1u : 0u;
  const uint32_t rem = (uint32_t)min(max((int32_t)rt.z, 0), 65535);
  const uint32_t pk = (uint32_t)rt.x | ((uint32_t)rt.y << 1) | ((uint32_t)rt.w << 11) | (ga << 15) | (rem << 16);
  a.prep[e] = make_uint4(xrow, pk, W_COL(c.c), 0u);
  if (d.st_bits) atomicOr(a.status, (int)d.st_bits);
}

__global__ void k_group_keys(const u3 *stepc, uint32_t *keys, uint32_t *idx, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = W_COL(stepc[i].c);
  idx[i] = (uint32_t)i;
}

template <int CTRL>
__device__ __forceinline__ double pm_add_dpp(double v) { return v + dpp_f64<CTRL>(v); }

// the 8 B-operand values of one (head, 16-draw tile) for this lane: slots 4 ks + q, ks = 0..7
struct PmB { float v[ROWF / 4]; };
__device__ __forceinline__ PmB pm_load_b(const float (*sBh)[PM_NPAD], int q, int nn) {
  PmB b;
#pragma unroll
  for (int ks = 0; ks < ROWF / 4; ++ks) b.v[ks] = sBh[4 * ks + q][nn];
  return b;
}

__global__ __launch_bounds__(BLOCK, 4) void k_posterior_mean(const PosteriorArgs a) {
  __shared__ float sB[2][ROWF][PM_NPAD];         // [head][slot][draw]
  __shared__ uint32_t s_col[PM_ROWS];            // coefficient column per row (0xFFFFFFFF: row past the end)
  __shared__ float4 s_rt[PM_ROWS];               // run-time slots 24..27
  __shared__ float s_ga[PM_ROWS];                // gate * actual (0 or 1)
  __shared__ uint32_t s_xrow[PM_ROWS];           // float index of the row's feature row
  __shared__ double s_sum[PM_ROWS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t pos0 = (int64_t)blockIdx.x * PM_ROWS;
  const float *Xf = reinterpret_cast<const float *>(a.tb.X);
  const float *Wf = reinterpret_cast<const float *>(a.tb.W);
  // ---- per-row set-up: thread r owns sorted position pos0 + r (PM_ROWS == BLOCK)
  uint32_t my_env = 0;
  {
    const int64_t pos = pos0 + tid;
    uint32_t col = 0xFFFFFFFFu, xrow = 0;
    float4 rt = make_float4(0.f, 0.f, 0.f, 0.f);
    float ga = 0.0f;
    if (pos < a.n) {
      my_env = a.perm[pos];
      const uint4 pr = a.prep[my_env];
      xrow = pr.x;
      col = pr.z;
      rt = make_float4((float)(pr.y & 1u), (float)((pr.y >> 1) & 1023u), (float)(pr.y >> 16),
                       (float)((pr.y >> 11) & 15u));
      ga = (float)((pr.y >> 15) & 1u);
    }
    s_col[tid] = col; s_xrow[tid] = xrow; s_rt[tid] = rt; s_ga[tid] = ga; s_sum[tid] = 0.0;
  }
  __syncthreads();
  // ---- A fragment of one row tile: lane l holds x[row = l & 15][slot = 4 ks + (l >> 4)], f32 until used
  const int q = lane >> 4;
  auto load_a = [&](int tile, float *af) {
    const int row = tile * 16 + (lane & 15);
    const uint32_t xr = s_xrow[row];  // rows past the end carry xrow = 0 (a valid address) and zero run-time fields:
                                      // loaded unconditionally, their results are never stored
#pragma unroll
    for (int ks = 0; ks < ROWF / 4; ++ks) af[ks] = Xf[xr + 4 * ks + q];
    const float4 rt = s_rt[row];  // k-step 6 = slots 24..27: the run-time fields replace the table's zeros
    af[RT_QUAD] = q == 0 ? rt.x : q == 1 ? rt.y : q == 2 ? rt.z : rt.w;
  };
  const int n_samples = a.tb.n_samples;
  // ---- column segments of the tile (rows are sorted by column: a segment is a contiguous run)
  int seg = 0;
  while (seg < PM_ROWS) {
    const uint32_t col = s_col[seg];  // uniform
    if (col == 0xFFFFFFFFu) break;
    int lo = seg + 1, hi = PM_ROWS;   // seg_end = first row whose column is larger (binary search, uniform)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_col[mid] > col) hi = mid; else lo = mid + 1;
    }
    const int seg_end = lo;
    for (int n0 = 0; n0 < n_samples; n0 += PM_NPAD) {
      __syncthreads();  // previous users of sB are done
      // stage B_col: W[(col * n_samples + s)][head][slot] -> sB[head][slot][s - n0]; 16-B global loads
      for (int idx = tid; idx < PM_NPAD * 2 * (ROWF / 4); idx += BLOCK) {
        const int s = idx >> 4, rem = idx & 15;
        const int head = rem >> 3, k4 = rem & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + s < n_samples)
          v = reinterpret_cast<const float4 *>(Wf)[((size_t)col * n_samples + n0 + s) * (2 * ROWF / 4) + head * (ROWF / 4) + k4];
        sB[head][4 * k4][s] = v.x; sB[head][4 * k4 + 1][s] = v.y; sB[head][4 * k4 + 2][s] = v.z; sB[head][4 * k4 + 3][s] = v.w;
      }
      __syncthreads();
      const int tiles = (min(PM_NPAD, n_samples - n0) + 15) >> 4;
      float afn[ROWF / 4];  // the next row tile's A fragment is requested while the current one computes
      load_a(wave * PM_TILES_PER_WAVE, afn);
#pragma unroll 1
      for (int i = 0; i < PM_TILES_PER_WAVE; ++i) {
        const int w_lo = (wave * PM_TILES_PER_WAVE + i) * 16, w_hi = w_lo + 16;
        double ad[ROWF / 4];
#pragma unroll
        for (int ks = 0; ks < ROWF / 4; ++ks) ad[ks] = (double)afn[ks];
        if (i + 1 < PM_TILES_PER_WAVE) load_a(wave * PM_TILES_PER_WAVE + i + 1, afn);
        if (!(seg < w_hi && seg_end > w_lo)) continue;  // no row of this tile in the segment (wave-uniform)
        float rs[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // per accumulator row (q + 4 j): sum over this lane's <= 7 draws (each
                                                 // term in [0, 1]: f32 is ample; the cross-lane sum is fp64)
        float ga[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ga[j] = s_ga[w_lo + q + 4 * j];
        // effectiveness enters only through eff * gate * actual: when no row of this tile has an open gate AND an
        // alert today (most tiles: alerts are budget-limited), its half of the GEMM and its sigmoids are skipped
        const bool any_eff = __any(ga[0] != 0.0f || ga[1] != 0.0f || ga[2] != 0.0f || ga[3] != 0.0f);
        // one 16-draw tile: 8 (+ 8) MFMAs and the epilogue. The effectiveness fragment is read from LDS before the
        // baseline MFMAs are issued, the NEXT tile's baseline fragment (into `nb`) before the effectiveness ones:
        // LDS latency sits under 512 cycles of MFMA. Two named buffers alternate (the loop is unrolled by two), so
        // no fragment is ever copied between registers.
        auto do_tile = [&](int nt, const PmB &cb, PmB &nb) {
          const int nn = nt * 16 + (lane & 15);
          PmB ce;
          if (any_eff) ce = pm_load_b(sB[1], q, nn);
          else if (nt + 1 < tiles) nb = pm_load_b(sB[0], q, nn + 16);
          pm_double4 accb = {0.0, 0.0, 0.0, 0.0}, acce = {0.0, 0.0, 0.0, 0.0};
#if W2A_PM_EXPERIMENT == 2  // timing experiment: VALU FMA instead of MFMA (results wrong)
#pragma unroll
          for (int ks = 0; ks < ROWF / 4; ++ks) accb[ks & 3] = fma(ad[ks], (double)cb.v[ks], accb[ks & 3]);
          if (any_eff) {
            if (nt + 1 < tiles) nb = pm_load_b(sB[0], q, nn + 16);
#pragma unroll
            for (int ks = 0; ks < ROWF / 4; ++ks) acce[ks & 3] = fma(ad[ks], (double)ce.v[ks], acce[ks & 3]);
          }
#else
          if (any_eff) {  // wave-uniform: two independent accumulation chains, interleaved
            if (nt + 1 < tiles) nb = pm_load_b(sB[0], q, nn + 16);
#pragma unroll
            for (int ks = 0; ks < ROWF / 4; ++ks) {
              accb = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[ks], (double)cb.v[ks], accb, 0, 0, 0);
              acce = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[ks], (double)ce.v[ks], acce, 0, 0, 0);
            }
          } else {
#pragma unroll
            for (int ks = 0; ks < ROWF / 4; ++ks)
              accb = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[ks], (double)cb.v[ks], accb, 0, 0, 0);
          }
#endif
          if (n0 + nn < n_samples) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // D[row = q + 4 j][col = lane & 15]
#if W2A_PM_EXPERIMENT == 1  // timing experiment: no sigmoid epilogue (results wrong)
              rs[j] += (float)accb[j] + (float)acce[j];
#else
              const float base = sigmoid_f32((float)accb[j]);
              float keep = 1.0f;
              if (any_eff) keep = 1.0f - sigmoid_f32((float)acce[j]) * ga[j];
              rs[j] += base * keep;
#endif
            }
          }
        };
        PmB b0 = pm_load_b(sB[0], q, lane & 15), b1;
        for (int nt = 0; nt < tiles; nt += 2) {
          do_tile(nt, b0, b1);
          if (nt + 1 < tiles) do_tile(nt + 1, b1, b0);
        }
        // sum over the 16 lanes that share q (one DPP row): xor 1, xor 2, half mirror, mirror
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double v = (double)rs[j];
          v = pm_add_dpp<0xB1>(v);
          v = pm_add_dpp<0x4E>(v);
          v = pm_add_dpp<0x141>(v);
          v = pm_add_dpp<0x140>(v);
          const int r = w_lo + q + 4 * j;
          if ((lane & 15) == 0 && r >= seg && r < seg_end) s_sum[r] += v;  // rows of other segments: A x other B, dropped
        }
      }
    }
    seg = seg_end;
  }
  __syncthreads();
  if (s_col[tid] != 0xFFFFFFFFu)
    a.reward[my_env] = (float)(-(1000.0 / 152.0) * s_sum[tid] / (double)n_samples);
}

#endif  // W2A_POSTERIOR_HIP_H
