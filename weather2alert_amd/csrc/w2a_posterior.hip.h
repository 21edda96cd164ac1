// w2a_posterior.hip.h -- today's reward averaged over ALL posterior draws of the env's coefficient column (the
// legacy env's eval mode, _deprecated/env.py:332-342, on today's linear-logistic form env.py:197-226): the pre-pass
// k_pm_prep, the kernel k_posterior_mean_v (fp64 vector FMAs, DPP-broadcast coefficients; second half of this file)
// and the fp64-MFMA form k_posterior_mean; both are in the library and w2a_set_posterior_kernel selects one at run time
// (W2A_PM_VECTOR / W2A_PM_MATRIX_F64). k_group_keys / k_group_inverse / k_pm_wd feed the once-per-episode grouping.
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_POSTERIOR_HIP_H
#define W2A_POSTERIOR_HIP_H

#include <type_traits>

// For env e with coefficient column c:   reward_e = mean_s  -(1000/152) * sigmoid(zb_s) * (1 - sigmoid(ze_s) * gate * actual)
//     zb_s = sum_k x_k * Wb[c][s][k],   ze_s = sum_k x_k * We[c][s][k]          (k over the 32 row slots)
// i.e. per column c one dense contraction  D_c [N_c envs][2 * n_samples] = A_c [N_c][32] * B_c [32][2 * n_samples]
// followed by a sigmoid / product / mean epilogue. Envs are served in the order of `perm` (env ids sorted by
// column, built once per episode by w2a_group_by_column). The MATRIX form (k_posterior_mean): a 256-row tile spans one
// column, sometimes two:
//   * the workgroup stages B_c (both heads, all draws: 25.6 KB of f32) in LDS once per column segment;
//   * each of its 4 waves owns 4 row tiles of 16: A fragments = the env's feature row of the day with the run-time slots
//     patched in (the same derive_day()/runtime_fields() as the step kernel), converted f32 -> f64 in registers;
//   * the rows of a column segment are re-ordered inside the workgroup so that the few with an open gate AND an alert
//     today (the only ones whose reward depends on the effectiveness head, env.py:218-221) sit in its first row
//     tiles: every other tile runs the baseline head alone;
//   * per 16-draw tile 7 (+ 7) v_mfma_f64_16x16x4_f64 over slots 0..27, the accumulators starting from the bias
//     coefficient of the draw (slot 29 holds 1.0 in every row); an 8th k-step (slots 28..31) only for schemas with a
//     25th table column. Products of f32 values are exact in fp64, so the logits carry ~1e-16 relative error, as in
//     the step kernels;
//   * epilogue in the accumulator layout (lane = draw column, 4 rows per lane): f32 sigmoids as in the step
//     kernels, closed gate = -inf logit, per-row sums over draws in fp64, 16-lane DPP all-reduce, one f32 per env.
// The step kernel then runs with W2A_STEP_REWARD_GIVEN and does everything else of env.py:238-262.
#define PM_ROWS 256                // sorted positions per workgroup: 16 row tiles of 16, 4 per wave
static_assert(PM_ROWS == BLOCK, "one thread per row in the set-up phase");
#define PM_TILES_PER_WAVE (PM_ROWS / 16 / (BLOCK / 64))
#define PM_NPAD 112                // draws per staging pass (7 MFMA column tiles)
typedef double pm_double4 __attribute__((ext_vector_type(4)));

struct PosteriorArgs {
  DevTables tb;
  StateArrays st;
  const uint32_t *inv;   // [n] sorted position of every env (inverse of the env ids sorted by coefficient column)
  const uint32_t *perm;  // [n] env id at every sorted position
  uint4 *prep;           // [n] per-env record of the day (k_pm_prep -> k_posterior_mean*), see W2A_PM_PREP_SCATTER
  const double *wd;      // [S * n_samples][2][32] coefficient rows as -log2(e) * W in fp64 (k_pm_wd)
  const uint4 *tiles;    // tile list of k_posterior_mean_v (k_tile_list)
  const uint32_t *n_tiles;
  const void *actions;
  int32_t act_dtype;
  float *reward;
  int32_t *status;
  int64_t n;
};

// Per-env record of the day, computed in ENV order (coalesced state / action reads, like phase A of k_step64).
// W2A_PM_PREP_SCATTER = 1 (default): written to the env's position in the column order (one scattered 16-B store per
// env: stores do not stall), so the reward kernels, which walk the envs in column order, read their records coalesced
// with no dependent gather. 0: written in env order (coalesced) and fetched as prep[perm[p]], a 16-B gather behind the
// coalesced read of perm -- measured a wash at 1 M envs (pre-pass + reward kernel: int8 102.5 vs 101.1 us, vector
// 171 vs 176 us; profiles/r03/pm_i8_variants.log): the gather costs what the scatter saved.
#ifndef W2A_PM_PREP_SCATTER
#define W2A_PM_PREP_SCATTER 1
#endif
// the record of sorted position `pos`
#define PM_REC(a, pos) (W2A_PM_PREP_SCATTER ? (a).prep[(pos)] : (a).prep[(a).perm[(pos)]])
// Record layout:
// x = float index of the feature row, y = run-time fields packed (alert_lag1 bit 0, alert_streak bits 1..10,
// alert_2wks bits 11..14, gate * actual bit 15, remaining_budget bits 16..31: budgets up to 65535, checked by the
// host class), z = coefficient column, w = env id.
__global__ void k_pm_prep(const PosteriorArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const uint32_t e = (uint32_t)i;
  const u3 h = a.st.hot3[e];
  const u3 c = a.st.stepc[e];
  const Day d = derive_day(h, c, load_action_raw(a.actions, a.act_dtype, e));
  const uint32_t xrow = (d.t * (uint32_t)(a.tb.S_w * a.tb.Y) + c.b) * ROWF;
  const float4 rt = runtime_fields(d);
  // effectiveness enters through eff * gate * actual (env.py:218-221): slot 30 of the row is the 0/1 gate flag, also
  // available as a bitmap of the day (1 KB, cache-resident) when the tables carry one
  uint32_t ga = 0u;
  if (d.actual)
    ga = a.tb.gate_bits ? (a.tb.gate_bits[(size_t)d.t * a.tb.gate_words + (c.b >> 5)] >> (c.b & 31u)) & 1u
                        : (reinterpret_cast<const float *>(a.tb.X)[xrow + 30] > 0.5f ? 1u : 0u);
  const uint32_t rem = (uint32_t)min(max((int32_t)rt.z, 0), 65535);
  const uint32_t pk = (uint32_t)rt.x | ((uint32_t)rt.y << 1) | ((uint32_t)rt.w << 11) | (ga << 15) | (rem << 16);
  a.prep[W2A_PM_PREP_SCATTER ? a.inv[e] : e] = make_uint4(xrow, pk, W_COL(c.c), e);
  // a finished env is no error HERE: in policy loops over batches that are not in lock step the step that follows runs
  // with W2A_STEP_SKIP_FINISHED and leaves such envs alone; without that flag the step itself raises the bit
  // (found by tools/sequence_fuzz.py: a whole-episode rollout through the per-day calls left W2A_ST_STEP_AFTER_DONE set)
  const uint32_t bits = d.st_bits & ~(uint32_t)W2A_ST_STEP_AFTER_DONE;
  if (bits) atomicOr(a.status, (int)bits);
}

__global__ void k_group_inverse(const uint32_t *perm, uint32_t *inv, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[perm[i]] = (uint32_t)i;
}

__global__ void k_group_keys(const u3 *stepc, uint32_t *keys, uint32_t *idx, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = W_COL(stepc[i].c);
  idx[i] = (uint32_t)i;
}

// Run once by w2a_create: does any coefficient row use slots 28, 30 or 31? (The reference schema's 25th table column,
// 'significance', is not a reward feature: its slot has no coefficient, and k_posterior_mean then contracts slots
// 0..27 only.) One thread per 32-float coefficient row.
__global__ void k_scan_tail_slots(const float4 *W, int64_t rows, int32_t *flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  const float4 v = W[i * (ROWF / 4) + ROWF / 4 - 1];  // slots 28..31; 29 is the bias
  if (v.x != 0.0f || v.z != 0.0f || v.w != 0.0f) atomicOr(flag, 1);
}

template <int CTRL>
__device__ __forceinline__ double pm_add_dpp(double v) { return v + dpp_f64<CTRL>(v); }

// the B-operand values of one (head, 16-draw tile) for this lane: slots 4 ks + q, ks = 0..KS-1, and the draw's bias
template <int KS>
struct PmB { float v[KS]; float bias; };
template <int KS>
__device__ __forceinline__ PmB<KS> pm_load_b(const float (*sBh)[PM_NPAD], int q, int nn) {
  PmB<KS> b;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) b.v[ks] = sBh[4 * ks + q][nn];
  b.bias = sBh[29][nn];
  return b;
}

// KS = 7: slots 0..27 through the MFMAs, bias as the accumulators' start value (slot 28 unused by the schema, slot 30
// has a zero coefficient, slot 31 is zero: include/w2a.h). KS = 8: all 32 slots through the MFMAs, start value 0.
template <int KS>
__global__ __launch_bounds__(BLOCK, 4) void k_posterior_mean(const PosteriorArgs a) {
  __shared__ float sB[2][ROWF][PM_NPAD];         // [head][slot][draw]
  __shared__ uint32_t s_col[PM_ROWS];            // coefficient column per row (0xFFFFFFFF: row past the end)
  __shared__ float4 s_rt[PM_ROWS];               // run-time slots 24..27
  __shared__ float s_ga[PM_ROWS];                // gate * actual (0 or 1)
  __shared__ uint32_t s_xrow[PM_ROWS];           // float index of the row's feature row
  __shared__ uint32_t s_env[PM_ROWS];            // env id of the row (rows are re-ordered inside the workgroup)
  __shared__ double s_sum[PM_ROWS];
  __shared__ uint32_t s_wga[BLOCK / 64];
  uint32_t *s_G = reinterpret_cast<uint32_t *>(s_sum);  // [PM_ROWS + 1] during set-up only
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int64_t pos0 = (int64_t)blockIdx.x * PM_ROWS;
  const float *Xf = reinterpret_cast<const float *>(a.tb.X);
  const float *Wf = reinterpret_cast<const float *>(a.tb.W);
  // ---- per-row set-up: thread r loads sorted position pos0 + r (PM_ROWS == BLOCK) ...
  {
    const int64_t pos = pos0 + tid;
    uint32_t col = 0xFFFFFFFFu, xrow = 0, my_env = 0;
    float4 rt = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t ga = 0;
    if (pos < a.n) {
      const uint4 pr = PM_REC(a, pos);
      my_env = pr.w;
      xrow = pr.x;
      col = pr.z;
      rt = make_float4((float)(pr.y & 1u), (float)((pr.y >> 1) & 1023u), (float)(pr.y >> 16),
                       (float)((pr.y >> 11) & 15u));
      ga = (pr.y >> 15) & 1u;
    }
    // ... and moves it to row p: inside each column segment the rows with gate * actual = 1 first (a stable partition;
    // s_col is unchanged by it). G[r] = number of such rows before row r.
    const uint64_t bal = __ballot(ga != 0);
    const uint32_t before = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    s_col[tid] = col;
    if (lane == 0) s_wga[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t G = before;
    for (int w = 0; w < wave; ++w) G += s_wga[w];
    s_G[tid] = G;
    if (tid == BLOCK - 1) s_G[PM_ROWS] = G + ga;
    int s_lo = 0, s_hi = tid;            // segment start: first row with this column
    while (s_lo < s_hi) {
      const int mid = (s_lo + s_hi) >> 1;
      if (s_col[mid] < col) s_lo = mid + 1; else s_hi = mid;
    }
    int e_lo = tid + 1, e_hi = PM_ROWS;  // segment end: first row with a larger column
    while (e_lo < e_hi) {
      const int mid = (e_lo + e_hi) >> 1;
      if (s_col[mid] > col) e_hi = mid; else e_lo = mid + 1;
    }
    __syncthreads();
    const uint32_t g0 = s_G[s_lo], g_seg = s_G[e_lo] - g0, g_me = G - g0;
    const uint32_t p = (uint32_t)s_lo + (ga ? g_me : g_seg + ((uint32_t)(tid - s_lo) - g_me));
    __syncthreads();                     // s_G is s_sum's memory
    s_xrow[p] = xrow; s_rt[p] = rt; s_ga[p] = (float)ga; s_env[p] = my_env; s_sum[tid] = 0.0;
  }
  __syncthreads();
  // ---- A fragment of one row tile: lane l holds x[row = l & 15][slot = 4 ks + (l >> 4)], f32 until used
  const int q = lane >> 4;
  auto load_a = [&](int tile, float *af) {
    const int row = tile * 16 + (lane & 15);
    const uint32_t xr = s_xrow[row];  // rows past the end carry xrow = 0 (a valid address) and zero run-time fields:
                                      // loaded unconditionally, their results are never stored
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = Xf[xr + 4 * ks + q];
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
      // stage B_col: W[(col * n_samples + s)][head][slot] -> sB[head][slot][s - n0]; 16-B global loads, consecutive
      // lanes take consecutive draws so the four LDS stores of a lane group fall into distinct banks
      for (int idx = tid; idx < PM_NPAD * 2 * (ROWF / 4); idx += BLOCK) {
        const int s = idx % PM_NPAD, rem = idx / PM_NPAD;
        const int head = rem >> 3, k4 = rem & 7;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n0 + s < n_samples)
          v = reinterpret_cast<const float4 *>(Wf)[((size_t)col * n_samples + n0 + s) * (2 * ROWF / 4) + head * (ROWF / 4) + k4];
        sB[head][4 * k4][s] = v.x; sB[head][4 * k4 + 1][s] = v.y; sB[head][4 * k4 + 2][s] = v.z; sB[head][4 * k4 + 3][s] = v.w;
      }
      __syncthreads();
      const int tiles = (min(PM_NPAD, n_samples - n0) + 15) >> 4;
      float afn[KS];  // the next row tile's A fragment is requested while the current one computes
      load_a(wave * PM_TILES_PER_WAVE, afn);
#pragma unroll 1
      for (int i = 0; i < PM_TILES_PER_WAVE; ++i) {
        const int w_lo = (wave * PM_TILES_PER_WAVE + i) * 16, w_hi = w_lo + 16;
        double ad[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ad[ks] = (double)afn[ks];
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
        // one 16-draw tile: KS (+ KS) MFMAs and the epilogue, compiled twice (with / without the effectiveness head:
        // a wave-uniform choice per row tile). The effectiveness fragment is read from LDS before the baseline MFMAs
        // are issued, the NEXT tile's baseline fragment (into `nb`) before the effectiveness ones: LDS latency sits
        // under the MFMAs. Two named buffers alternate (the loop is unrolled by two).
        auto do_tile = [&](auto eff_c, int nt, const PmB<KS> &cb, PmB<KS> &nb) {
          constexpr bool EFF = decltype(eff_c)::value;
          const int nn = nt * 16 + (lane & 15);
          PmB<KS> ce;
          if (EFF) ce = pm_load_b<KS>(sB[1], q, nn);
          else if (nt + 1 < tiles) nb = pm_load_b<KS>(sB[0], q, nn + 16);
          const double zb0 = KS == ROWF / 4 ? 0.0 : (double)cb.bias;
          pm_double4 accb = {zb0, zb0, zb0, zb0};
          if (EFF) {  // two independent accumulation chains, interleaved
            const double ze0 = KS == ROWF / 4 ? 0.0 : (double)ce.bias;
            pm_double4 acce = {ze0, ze0, ze0, ze0};
            if (nt + 1 < tiles) nb = pm_load_b<KS>(sB[0], q, nn + 16);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
              accb = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[ks], (double)cb.v[ks], accb, 0, 0, 0);
              acce = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[ks], (double)ce.v[ks], acce, 0, 0, 0);
            }
            if (n0 + nn < n_samples) {
#pragma unroll
              for (int j = 0; j < 4; ++j)  // D[row = q + 4 j][col = lane & 15]
                rs[j] += sigmoid_f32((float)accb[j]) * (1.0f - sigmoid_f32((float)acce[j]) * ga[j]);
            }
          } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
              accb = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[ks], (double)cb.v[ks], accb, 0, 0, 0);
            if (n0 + nn < n_samples) {
#pragma unroll
              for (int j = 0; j < 4; ++j) rs[j] += sigmoid_f32((float)accb[j]);
            }
          }
        };
        PmB<KS> b0 = pm_load_b<KS>(sB[0], q, lane & 15), b1;
        if (any_eff) {
          for (int nt = 0; nt < tiles; nt += 2) {
            do_tile(std::true_type{}, nt, b0, b1);
            if (nt + 1 < tiles) do_tile(std::true_type{}, nt + 1, b1, b0);
          }
        } else {
          for (int nt = 0; nt < tiles; nt += 2) {
            do_tile(std::false_type{}, nt, b0, b1);
            if (nt + 1 < tiles) do_tile(std::false_type{}, nt + 1, b1, b0);
          }
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
    a.reward[s_env[tid]] = (float)(-(1000.0 / 152.0) * s_sum[tid] / (double)n_samples);
}

// ----------------------------------------------------------------------------------------
// lane = env form (default): fp64 FMAs on the vector ALU, coefficients broadcast by DPP
// ----------------------------------------------------------------------------------------
// On MI355X the fp64 matrix rate equals the fp64 vector rate (v_mfma_f64_16x16x4 = 64 cycles for 1024 FMAs,
// v_fma_f64 = 4 cycles for 64) and fp64 MFMAs do not overlap ANY other vector work (tools/mfma_overlap_probe.hip:
// MFMAs and f32 sigmoids interleaved in one wave take the sum of their times): the matrix form buys no arithmetic and
// pays for its fragments (B through LDS with a convert per use, 112-for-100 draw padding, a cross-lane reduction).
// What the matrix unit does provide is operand delivery, and the vector unit has a second way to get it: gfx90a+
// DPP on 64-bit operations, row_newbcast:n = lane n of every 16-lane row broadcast to the row. So:
//   * a workgroup serves one tile = <= 512 envs of ONE coefficient column (k_tile_list), lane = env; the env's feature
//     row of the day (run-time slots patched in) sits in registers as 28 (32) doubles;
//   * the workgroup stages the column's block of a fp64 copy of W that already carries the factor -log2(e) (k_pm_wd,
//     once per episode) in LDS; per draw every lane reads TWO doubles of the draw's coefficient row, slots (lane & 15)
//     and 16 + (lane & 15): 8 LDS cycles per wave and draw;
//   * v_fmac_f64_dpp acc, coef row_newbcast:k, x[k] -- 28 FMAs in one chain started from the broadcast bias -- then
//     v_cvt_f32_f64, v_exp_f32, add, v_rcp_f32; four terms are added in f32, the blocks in fp64: exactly n_samples
//     draws, the sum over draws lane-local. 33 vector instructions per env-wave and draw.
// The effectiveness head matters for the few rows with gate * actual = 1 only. After the baseline pass their owners
// publish these rows in LDS, <= 64 at a time, and EVERY wave of the workgroup takes the same group and an eighth of
// the draws with both heads; the row's owner adds the partial sums in a fixed order: the extra work is spread
// evenly instead of making one wave a straggler.
// Every lane of a wave stays active in the draw loops (DPP reads its source lane whatever that lane's own row is).
#ifndef PMV_THREADS
#define PMV_THREADS 512
#endif
#ifndef W2A_PMV_NPAD
#define W2A_PMV_NPAD 112           // draws staged in LDS per pass
#endif
__global__ void k_pm_wd(const float *W, double *wd, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) wd[i] = -1.4426950408889634 * (double)W[i];
}

// z += sum_k coef[k] * x[k] for the slots held by `b` (lane n of each 16-lane row holds slot base + n): one chain of
// dependent FMAs, which the SIMD's other waves interleave with theirs (two chains cost two more instructions per draw
// and measured 9 % slower). The leading s_nop covers the VALU-write -> DPP-read hazard should the compiler have
// just moved `b` between registers.
__device__ __forceinline__ void pmv_fma16(double &z0, double b, const double *x) {
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(z0)
      : "v"(b), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]),
        "v"(x[9]), "v"(x[10]), "v"(x[11]), "v"(x[12]), "v"(x[13]), "v"(x[14]), "v"(x[15]));
}
__device__ __forceinline__ void pmv_fma12(double &z0, double b, const double *x) {
  asm("s_nop 1\n\t"
      "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
      : "+v"(z0)
      : "v"(b), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(x[8]),
        "v"(x[9]), "v"(x[10]), "v"(x[11]));
}
// lane 13 of every row holds slot 29, the bias
__device__ __forceinline__ double pmv_bias(double b) {
  double r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:13 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(b));
  return r;
}
// one head's logit (times -log2 e) of the lane's env for the draw whose coefficient row is sWd = [slot]
template <int KS>
__device__ __forceinline__ float pmv_logit(const double *sWd, int l15, const double (&ax)[4 * KS]) {
  const double b0 = sWd[l15], b1 = sWd[16 + l15];
  double z = KS == ROWF / 4 ? 0.0 : pmv_bias(b1);  // all 32 slots (slot 29 of the row holds 1.0) / 0..27 + bias
  pmv_fma16(z, b0, &ax[0]);
  if (KS == ROWF / 4) pmv_fma16(z, b1, &ax[16]);
  else pmv_fma12(z, b1, &ax[16]);
  return (float)z;
}

// sigmoid(zb) [* (1 - sigmoid(ze))] of the lane's env for staged draw s; sW = [draw][head][slot]
template <int KS, bool EFF>
__device__ __forceinline__ float pmv_term(const double (*sW)[2][ROWF], int s, int l15, const double (&ax)[4 * KS]) {
  float t = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pmv_logit<KS>(sW[s][0], l15, ax)));
  if (EFF) t *= 1.0f - __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pmv_logit<KS>(sW[s][1], l15, ax)));
  return t;
}

// baseline pass: sum over `draws` staged draws of sigmoid(zb). Four terms (each in [0, 1]) are added in f32 (error
// <= 3 ulp(4) = 7e-7 per block), the blocks in fp64.
template <int KS>
__device__ __forceinline__ double pmv_draws(const double (*sW)[2][ROWF], int draws, int l15, const double (&ax)[4 * KS]) {
  double sum = 0.0;
#ifdef W2A_PMV_DEBUG_DRAWS  // timing experiment: cap the draw loop (results wrong)
  draws = min(draws, W2A_PMV_DEBUG_DRAWS);
#endif
  int s = 0;
#pragma unroll 1
  for (; s + 4 <= draws; s += 4)
    sum += (double)((pmv_term<KS, false>(sW, s, l15, ax) + pmv_term<KS, false>(sW, s + 1, l15, ax)) +
                    (pmv_term<KS, false>(sW, s + 2, l15, ax) + pmv_term<KS, false>(sW, s + 3, l15, ax)));
#pragma unroll 1
  for (; s < draws; ++s) sum += (double)pmv_term<KS, false>(sW, s, l15, ax);
  return sum;
}

// f32 row (as loaded / staged) -> the doubles of the FMA blocks. The empty asm makes the f32 values opaque at this
// point, so the 28 converts stay where they are written instead of being hoisted out of the segment loop (which
// would keep the f32 AND the fp64 row live for the whole kernel and halve the occupancy).
template <int KS>
__device__ __forceinline__ void pmv_widen(float4 (&xf)[KS], double (&ax)[4 * KS]) {
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    asm volatile("" : "+v"(xf[q].x), "+v"(xf[q].y), "+v"(xf[q].z), "+v"(xf[q].w));
    ax[4 * q] = xf[q].x; ax[4 * q + 1] = xf[q].y; ax[4 * q + 2] = xf[q].z; ax[4 * q + 3] = xf[q].w;
  }
}

// Tiles of the lane = env kernel: <= PMV_THREADS consecutive sorted positions of ONE coefficient column, listed once
// per episode by w2a_group_by_column (k_tile_bounds + k_tile_list). A workgroup then stages exactly one coefficient
// block and runs one pass; with fixed 512-position ranges a third of the workgroups straddled two columns and ran
// the two passes one after the other with half their waves idle.
__global__ void k_tile_bounds(const uint32_t *keys_sorted, uint32_t *col_start, uint32_t *col_end, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = keys_sorted[i];
  if (i == 0 || keys_sorted[i - 1] != k) col_start[k] = (uint32_t)i;
  if (i == n - 1 || keys_sorted[i + 1] != k) col_end[k] = (uint32_t)i + 1u;
}

// one workgroup: tiles of <= tile_rows positions of every column in column order; tiles[j] = (first position, rows,
// column, 0)
__global__ __launch_bounds__(1024) void k_tile_list(const uint32_t *col_start, const uint32_t *col_end, int32_t S,
                                                    uint4 *tiles, uint32_t *n_tiles, uint32_t tile_rows) {
  __shared__ uint32_t s_cnt[1024];
  const int tid = threadIdx.x;
  const int chunk = (S + 1023) / 1024;
  const int c0 = min(S, tid * chunk), c1 = min(S, c0 + chunk);
  uint32_t mine = 0;
  for (int c = c0; c < c1; ++c) mine += (col_end[c] - col_start[c] + tile_rows - 1) / tile_rows;
  s_cnt[tid] = mine;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // inclusive scan
    const uint32_t v = tid >= d ? s_cnt[tid - d] : 0u;
    __syncthreads();
    s_cnt[tid] += v;
    __syncthreads();
  }
  uint32_t j = s_cnt[tid] - mine;
  for (int c = c0; c < c1; ++c)
    for (uint32_t st = col_start[c]; st < col_end[c]; st += tile_rows)
      tiles[j++] = make_uint4(st, min(tile_rows, col_end[c] - st), (uint32_t)c, 0u);
  if (tid == 1023) *n_tiles = s_cnt[1023];
}

template <int KS>
__global__ __launch_bounds__(PMV_THREADS, 4) void k_posterior_mean_v(const PosteriorArgs a) {
  constexpr int WAVES = PMV_THREADS / 64;
  __shared__ uint32_t s_wga[WAVES];                 // per wave: rows with gate * actual = 1
  __shared__ double sW[W2A_PMV_NPAD][2][ROWF];      // the column's coefficient block
  __shared__ float4 s_ax[64][KS];                   // effectiveness phase: the rows of one group, as f32
  __shared__ double s_part[WAVES][64];              // effectiveness phase: per-wave, per-lane partial sums
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15;
  // XCD k walks the k-th contiguous eighth of the tile list: the 2-3 tiles of a coefficient column run on one XCD and
  // its block comes out of that XCD's L2 after the first of them has staged it
  const uint32_t n_tiles = *a.n_tiles, per_xcd = (n_tiles + 7u) >> 3;
  const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tile >= n_tiles) return;  // the grid covers the largest possible tile count
  const uint4 tl = a.tiles[tile];  // first sorted position, rows, column
  const uint32_t col = __builtin_amdgcn_readfirstlane(tl.z);
  const int rows = (int)__builtin_amdgcn_readfirstlane(tl.y);
  const int n_samples = a.tb.n_samples;
  // ---- the row's record (coalesced) and its feature row of the day (requested at once, kept as f32 for the whole
  // kernel); rows past the end of the tile: feature row 0 (a valid address), never stored
  uint4 rec = make_uint4(0u, 0u, 0u, 0u);  // x feature row, y run-time fields, z column, w env id
  if (tid < rows) rec = PM_REC(a, tl.x + tid);
  float4 xf[KS];
  {
    const float4 *xp = a.tb.X + (rec.x >> 2);
#pragma unroll
    for (int q = 0; q < KS; ++q) xf[q] = xp[q];
    xf[RT_QUAD] = make_float4((float)(rec.y & 1u), (float)((rec.y >> 1) & 1023u), (float)(rec.y >> 16),
                              (float)((rec.y >> 11) & 15u));  // slots 24..27: the run-time fields of k_pm_prep
  }
  const uint32_t ga = (rec.y >> 15) & 1u;
  const uint64_t bal = __ballot(ga != 0);
  if (lane == 0) s_wga[wave] = (uint32_t)__popcll(bal);
  double sum = 0.0;
  uint32_t G = 0;      // rows with gate * actual = 1 before this one
  int n_eff = 0;       // ... in the whole tile
  for (int n0 = 0; n0 < n_samples; n0 += W2A_PMV_NPAD) {
    const int draws = min(W2A_PMV_NPAD, n_samples - n0);
    __syncthreads();  // previous users of sW are done
    // stage wd[(col * n_samples + n0 + s)][head][slot] -> sW[s][head][slot]: 16-B loads and stores, coalesced
    {
      const uint4 *src = reinterpret_cast<const uint4 *>(a.wd + ((size_t)col * n_samples + n0) * (2 * ROWF));
      uint4 *dst = reinterpret_cast<uint4 *>(&sW[0][0][0]);
      for (int idx = tid; idx < draws * (2 * ROWF / 2); idx += PMV_THREADS) dst[idx] = src[idx];
    }
    __syncthreads();
    if (n0 == 0) {
      G = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = s_wga[w];
        G += w < wave ? c : 0u;
        n_eff += (int)c;
      }
      n_eff = __builtin_amdgcn_readfirstlane(n_eff);
    }
    double contrib = 0.0;
    // baseline pass: the wave's own rows, every draw
    if (64 * wave < rows) {
      double ax[4 * KS];
      pmv_widen<KS>(xf, ax);
      contrib = pmv_draws<KS>(sW, draws, l15, ax);
    }
    // effectiveness phase, groups of <= 64 rows: their owners publish the rows, then EVERY wave takes an eighth of
    // the draws with both heads. A group of cnt rows fills R = ceil(cnt / 16) DPP rows of a wave; the wave's other
    // DPP rows work on other draws of its eighth at the same time (P = 4 / R draws in flight).
    for (int grp = 0; grp < n_eff; grp += 64) {
      const int cnt = min(64, n_eff - grp);
      const int j_own = (int)G - grp;  // this thread's row is row j_own of the group (if it is one)
      const bool own = ga && j_own >= 0 && j_own < 64;
      if (own) {
#pragma unroll
        for (int q = 0; q < KS; ++q) s_ax[j_own][q] = xf[q];
      }
      __syncthreads();
      const int R = (cnt + 15) >> 4, P = R == 1 ? 4 : (R == 2 ? 2 : 1);
      const int drow = lane >> 4;                // DPP row of the lane
      const int j = l15 + 16 * (drow % R);       // group row served by the lane
      const int sub = drow / R;                  // which of the P draws in flight
      const int d0 = wave * draws / WAVES, d1 = (wave + 1) * draws / WAVES;
      double part = 0.0;
      {
        float4 ef[KS];
#pragma unroll
        for (int q = 0; q < KS; ++q) ef[q] = s_ax[j < cnt ? j : 0][q];
        double ax[4 * KS];
        pmv_widen<KS>(ef, ax);
#pragma unroll 1
        for (int s = d0; s < d1; s += P) {       // uniform trip count; lanes past the slice redo its last draw
          const int sl = s + sub;
          const float t = pmv_term<KS, true>(sW, sl < d1 ? sl : d1 - 1, l15, ax);
          part += (sl < d1 && sub < P) ? (double)t : 0.0;
        }
      }
      s_part[wave][lane] = part;
      __syncthreads();
      if (own) {
        double t = 0.0;
#pragma unroll 1
        for (int w = 0; w < WAVES; ++w)
#pragma unroll 1
          for (int p = 0; p < P; ++p) t += s_part[w][(j_own & 15) + 16 * ((j_own >> 4) + R * p)];
        contrib = t;  // replaces the baseline-only value
      }
    }
    sum += contrib;
  }
  if (tid < rows) a.reward[rec.w] = (float)(-(1000.0 / 152.0) * sum / (double)n_samples);
}

#endif  // W2A_POSTERIOR_HIP_H
