// w2a_posterior_i8.hip.h -- the posterior-mean reward contraction on the int8 matrix cores (W2A_PM_MATRIX_I8).
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_POSTERIOR_I8_HIP_H
#define W2A_POSTERIOR_I8_HIP_H

// Why int8. The contraction D_c [envs x 2*draws] = A_c [envs x 32 slots] * B_c [32 x 2*draws] (w2a_posterior.hip.h) needs
// logits good to ~1e-6, i.e. more than f32 accumulation of 28 terms guarantees, and on MI355X the fp64 matrix rate equals
// the fp64 vector rate (profiles/r02/mfma_overlap_probe.log). The only matrix rate above the vector ALU is at <= 16-bit
// inputs -- and v_mfma_i32_16x16x64_i8 accumulates EXACTLY (int32). So both operands become fixed point:
//
//   x_k = X_k * 2^(ex_k - 30)      X_k int, |X_k| < 2^30; 2^ex_k > the largest |x_k| the slot can hold (table maximum
//                                  per slot, scanned once; run-time slots from their ranges: lag <= 1, remaining_budget
//                                  <= the episode's largest budget B, streak <= min(T, B), alert_2wks <= min(14, B):
//                                  both count GRANTED alerts, and no more than B are granted). A float32 whose
//                                  exponent lies within 2^7 of the slot's range converts exactly; smaller ones are
//                                  truncated at 2^(ex_k - 30).
//   w_k * 2^ex_k = W_k * 2^(ew - 30)   W_k int, |W_k| <= 2^30, rounded to nearest; 2^ew > max_k |w_k| 2^ex_k of THIS
//                                  (column, draw, head) coefficient row -- every output column of the GEMM has its own
//                                  scale, applied in the epilogue.
//   z = sum_k x_k w_k = 2^(ew - 60) * sum_k X_k W_k  up to  |dz| <= 48 * 2^(ew - 30)  (rounding of W, truncation of X).
//
// X_k and W_k are written in four balanced base-256 digits (each an int8 in [-128, 127]; X = sum_i dx_i 2^(24 - 8i)) and
// the product is expanded by digit pairs. Pairs (i, j) with i + j = s share the weight 2^(48 - 8s); the pairs with
// s <= 3 are kept (10 of 16; what is dropped is below 96 * 2^(ew - 30) in z, typically 2^(ew - 28)). K = 64 of one MFMA
// holds two pairs x 32 slots, so with the A operands P = (X0 | X1), Q = (X2 | X3) and the B operands B_m = (W_m | W_{m-1}):
//      A0 = P.B0          A1 = P.B1          A2 = P.B2 + Q.B0          A3 = P.B3 + Q.B1          6 MFMAs per 16 x 16 tile
//      z  = 2^(ew - 12) (A0 + A1 2^-8 + A2 2^-16 + A3 2^-24)
// -- every partial sum an exact int32 (|A_s| < 2^21). The f32 epilogue (two shifts-and-adds in int32, two converts, one
// multiply, one FMA, then v_exp_f32 / v_rcp_f32 as in the other kernels) rounds z to float32 like `(float) zb` does there.
// A-priori bound per coefficient row: |dz| <= 1.5 * 2^(ew - 23.4); rows with ew > W2A_PI8_EW_MAX (max |w_k| 2^ex_k >= 16:
// |dz| could exceed ~2e-6) flag their column, and tiles of flagged columns take an exact fp64 path inside the same
// kernel (slow, never wrong) -- `pm_kernel="auto"` then measures the vector kernel to be the faster one for such data.
//
// Geometry: one workgroup = one tile of <= 256 sorted positions of ONE coefficient column (a second tile list of
// w2a_group_by_column), 4 waves. Lane = row in the set-up: record, feature row of the day, fixed-point digits -> LDS
// image [row][4 planes][32 slots] (rows with gate * actual = 1 moved to the front: only their 16-row tiles run the
// effectiveness head). Each wave then owns the row tiles m = wave, wave + 4, ...: A fragments in registers for the whole
// kernel, B fragments from the staged int8 block of the column (25.6 KB for 100 draws, both heads), lane = draw column
// in the accumulators, sums over draws by DPP. The X image and the coefficient block share one LDS buffer (40 KB per
// workgroup): four workgroups per CU.
#ifndef W2A_PI8_EW_MAX
#define W2A_PI8_EW_MAX 4
#endif
#define PI8_ROWS 256
#define PI8_THREADS 256
#define PI8_WAVES (PI8_THREADS / 64)
#define PI8_MT (PI8_ROWS / 16)            // 16-row tiles per workgroup
#define PI8_MT_PER_WAVE (PI8_MT / PI8_WAVES)
#define PI8_NPAD 112                       // draws staged per pass (7 column tiles of 16)
#define PI8_XSTRIDE 36                     // uint32 per row of the X image: 32 (4 planes x 32 slots) + 4 pad
#define PI8_WSTRIDE 68                     // uint32 per DRAW of the staged coefficient block: 2 heads x 32 + 4 pad. The 16
                                           // lanes of a b128 read take 16 consecutive draws: 68 = 4 (mod 64) spreads them over
                                           // all 64 banks (an unpadded 64-word stride put all 16 on the same four: 16-way)
typedef int pi8_v4i __attribute__((ext_vector_type(4)));

// ---- once per handle: largest |x| per slot over the whole feature table (float bits compare like unsigned ints)
__global__ void k_pi8_slot_max(const float4 *X, int64_t n_quads, uint32_t *xmax_bits) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;  // a multiple of 8: a thread always sees the same quad of the row
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int q = (int)(i & 7);
  for (; i < n_quads; i += stride) {
    const float4 v = X[i];
    m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
  }
  atomicMax(&xmax_bits[4 * q], __float_as_uint(m.x));
  atomicMax(&xmax_bits[4 * q + 1], __float_as_uint(m.y));
  atomicMax(&xmax_bits[4 * q + 2], __float_as_uint(m.z));
  atomicMax(&xmax_bits[4 * q + 3], __float_as_uint(m.w));
}
// ---- once per episode: the largest budget of the batch (remaining_budget never exceeds it)
__global__ void k_pi8_budget_max(const u3 *stepc, int64_t n, uint32_t *bmax) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int32_t b = i < n ? (int32_t)stepc[i].a : 0;
  if (b < 0) b = 0;
  for (int off = 32; off; off >>= 1) b = max(b, __shfl_xor(b, off));
  if ((threadIdx.x & 63) == 0) atomicMax(bmax, (uint32_t)b);
}
// ---- slot scales: xs[k] = 2^(30 - ex_k) (feature -> fixed point), xs[32 + k] = 2^ex_k (coefficient pre-scale)
__global__ void k_pi8_scales(const uint32_t *xmax_bits, const uint32_t *bmax, int32_t T, float *xs) {
  const int k = threadIdx.x;
  if (k >= ROWF) return;
  float m = __uint_as_float(xmax_bits[k]);
  const float b = (float)(*bmax);
  if (k == 24) m = 1.0f;                    // alert_lag1
  if (k == 25) m = fminf((float)T, b);      // alert_streak: consecutive GRANTED alerts <= alerts granted <= budget, < T
  if (k == 26) m = b;                       // remaining_budget <= largest budget
  if (k == 27) m = fminf(14.0f, b);         // the agent's 14-day count of granted alerts
  int e = 0;
  if (m > 0.0f) (void)frexpf(m, &e);        // m = f * 2^e, 0.5 <= f < 1: 2^e > m
  xs[k] = ldexpf(1.0f, 30 - e);
  xs[ROWF + k] = ldexpf(1.0f, e);
}
__device__ __forceinline__ uint32_t pi8_digits(int32_t v) {  // balanced base-256 digits of v, packed: byte b = digit of 2^(8b)
  return ((uint32_t)v + 0x80808080u) ^ 0x80808080u;
}
// ---- once per episode: coefficient rows -> int8 digit planes [row][plane d][32 slots] + the row's epilogue scale
__global__ void k_pi8_wq(const float *W, const float *xs, int64_t rows, int32_t n_samples, uint32_t *wq, float *wscale,
                         uint32_t *colflag) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float w[ROWF];
  float m = 0.0f;
#pragma unroll
  for (int k = 0; k < ROWF; ++k) {
    w[k] = W[r * ROWF + k] * xs[ROWF + k];  // exact: a power of two
    m = fmaxf(m, fabsf(w[k]));
  }
  int ew = 0;
  if (m > 0.0f) (void)frexpf(m, &ew);
  if (ew > W2A_PI8_EW_MAX) atomicOr(&colflag[r / (2 * (int64_t)n_samples)], 1u);
  const float s = ldexpf(1.0f, 30 - ew);
  uint32_t planes[4][8];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    uint32_t d[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) d[c] = pi8_digits(__float2int_rn(w[4 * g + c] * s));
#pragma unroll
    for (int p = 0; p < 4; ++p) {  // plane p = digit of 2^(24 - 8p) = byte 3 - p
      const int sh = 8 * (3 - p);
      planes[p][g] = ((d[0] >> sh) & 255u) | (((d[1] >> sh) & 255u) << 8) | (((d[2] >> sh) & 255u) << 16) | (((d[3] >> sh) & 255u) << 24);
    }
  }
  uint4 *dst = reinterpret_cast<uint4 *>(wq + r * ROWF);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    dst[2 * p] = make_uint4(planes[p][0], planes[p][1], planes[p][2], planes[p][3]);
    dst[2 * p + 1] = make_uint4(planes[p][4], planes[p][5], planes[p][6], planes[p][7]);
  }
  // z * (-log2 e) = wscale * (A0 2^8 + A1 + (A2 2^8 + A3) 2^-16):  2^(ew - 12) * 2^-8 * (-log2 e)
  wscale[r] = -1.4426950408889634f * ldexpf(1.0f, ew - 20);
}

struct PmI8Args {
  PosteriorArgs p;
  const uint4 *tiles;        // tile list with PI8_ROWS positions per tile
  const uint32_t *n_tiles;
  const uint32_t *wq;        // [S * n_samples * 2][32] int8 digit planes
  const float *wscale;       // [S * n_samples * 2]
  const uint32_t *colflag;   // [S] 1 = some coefficient row of the column is outside the fixed-point range
  const float *xs;           // [64] slot scales (k_pi8_scales)
};

// 4 x 4 byte transpose: d[c] = packed digits of slot 4g + c  ->  plane words (plane p = byte 3 - p of every slot)
__device__ __forceinline__ void pi8_planes(const uint32_t d[4], uint32_t out[4]) {
  // v_perm_b32(hi, lo, sel): bytes 0..3 of lo, 4..7 of hi
  const uint32_t l01 = __builtin_amdgcn_perm(d[1], d[0], 0x05010400u);  // d0.b0 d1.b0 d0.b1 d1.b1
  const uint32_t h01 = __builtin_amdgcn_perm(d[1], d[0], 0x07030602u);  // d0.b2 d1.b2 d0.b3 d1.b3
  const uint32_t l23 = __builtin_amdgcn_perm(d[3], d[2], 0x05010400u);
  const uint32_t h23 = __builtin_amdgcn_perm(d[3], d[2], 0x07030602u);
  out[3] = __builtin_amdgcn_perm(l23, l01, 0x05040100u);  // byte 0 of d0..d3: digit of 2^0  = plane 3
  out[2] = __builtin_amdgcn_perm(l23, l01, 0x07060302u);  // byte 1                           = plane 2
  out[1] = __builtin_amdgcn_perm(h23, h01, 0x05040100u);  // byte 2                           = plane 1
  out[0] = __builtin_amdgcn_perm(h23, h01, 0x07060302u);  // byte 3: digit of 2^24            = plane 0
}
template <int CTRL>
__device__ __forceinline__ float pi8_add_dpp(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// ---- pieces shared by k_posterior_mean_i8 (one day per launch) and k_pm_rollout_i8 (a whole episode per launch) ------
// a row's 32 floats -> fixed-point digits -> its 128 B of the X image (4 planes x 32 slots; 8 x 16-B stores, consecutive
// rows sit 4 banks apart)
__device__ __forceinline__ void pi8_store_row(const float4 (&xf)[ROWF / 4], const float *xs, uint32_t *row) {
  uint32_t pl[4][ROWF / 4];
#pragma unroll
  for (int g = 0; g < ROWF / 4; ++g) {
    uint32_t d[4], o[4];
    d[0] = pi8_digits((int32_t)(xf[g].x * xs[4 * g]));      // truncating convert: |x| < 2^ex_k by construction
    d[1] = pi8_digits((int32_t)(xf[g].y * xs[4 * g + 1]));
    d[2] = pi8_digits((int32_t)(xf[g].z * xs[4 * g + 2]));
    d[3] = pi8_digits((int32_t)(xf[g].w * xs[4 * g + 3]));
    pi8_planes(d, o);
#pragma unroll
    for (int p = 0; p < 4; ++p) pl[p][g] = o[p];
  }
  uint4 *dst = reinterpret_cast<uint4 *>(row);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    dst[2 * p] = make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
    dst[2 * p + 1] = make_uint4(pl[p][4], pl[p][5], pl[p][6], pl[p][7]);
  }
}
// the wave's A fragments: row tiles wave, wave + 4, ...; lane (c16, q) holds 16 slots of one plane of row c16
__device__ __forceinline__ void pi8_load_a(const uint32_t (*sX)[PI8_XSTRIDE], int wave, int lane, pi8_v4i *P, pi8_v4i *Q) {
  const int q = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int i = 0; i < PI8_MT_PER_WAVE; ++i) {
    const uint32_t *r = sX[(wave + i * PI8_WAVES) * 16 + c16];
    P[i] = *reinterpret_cast<const pi8_v4i *>(r + 4 * q);        // (X0 | X1): 16 slots of one plane per lane group
    Q[i] = *reinterpret_cast<const pi8_v4i *>(r + 16 + 4 * q);   // (X2 | X3)
  }
}
// all column tiles of the staged draws for the wave's row tiles: rs[i][j] += sum over this lane's draw columns of
// sigmoid(zb) [* (1 - sigmoid(ze) * ga)] for accumulator row 4 q + j of row tile wave + 4 i
__device__ __forceinline__ void pi8_accumulate(const uint32_t (*sW)[PI8_WSTRIDE], const float *sScale, const float *sGa,
                                               const pi8_v4i *P, const pi8_v4i *Q, float (*rs)[4], int ntiles, int draws,
                                               int rows, int eff_tiles, int wave, int lane) {
  const int q = lane >> 4, c16 = lane & 15;
#pragma unroll 1
  for (int nt = 0; nt < ntiles; ++nt) {
    const int dcol = nt * 16 + c16;        // this lane's draw (accumulator column)
    const int drow = dcol * 2;             // its scale entries: drow + head
    const float dvalid = dcol < draws ? 1.0f : 0.0f;  // draws past the end (zero digits -> sigmoid 0.5) count 0
    // B_m = (W_m | W_{m-1}): lane groups 0,1 read plane m, groups 2,3 plane m-1 (zero for m = 0)
    auto load_b = [&](int head, pi8_v4i *B) {
      const uint32_t *w = &sW[dcol][head * ROWF];
      const int half = 4 * (q & 1);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int plane = q < 2 ? m : m - 1;
        const pi8_v4i v = *reinterpret_cast<const pi8_v4i *>(w + 8 * max(plane, 0) + half);
        B[m] = plane >= 0 ? v : pi8_v4i{0, 0, 0, 0};
      }
    };
    // (packed f32 forms of this epilogue -- v_pk_mul / v_pk_fma / v_pk_add on register pairs -- measured 4 % slower)
    auto logits = [&](const pi8_v4i &Pm, const pi8_v4i &Qm, const pi8_v4i *B, float sc, float *z) {
      const pi8_v4i zero = {0, 0, 0, 0};
      pi8_v4i a0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Pm, B[0], zero, 0, 0, 0);
      pi8_v4i a1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Pm, B[1], zero, 0, 0, 0);
      pi8_v4i a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Pm, B[2], zero, 0, 0, 0);
      pi8_v4i a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Pm, B[3], zero, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qm, B[0], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(Qm, B[1], a3, 0, 0, 0);
      const float sc16 = sc * 1.52587890625e-05f;  // 2^-16
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = (float)(a0[j] * 256 + a1[j]);
        const float v = (float)(a2[j] * 256 + a3[j]);
        z[j] = fmaf(v, sc16, u * sc);  // = -log2(e) * logit
      }
    };
    pi8_v4i B0[4], B1[4];
    load_b(0, B0);
    const float sc0 = sScale[drow];
#pragma unroll
    for (int i = 0; i < PI8_MT_PER_WAVE; ++i) {
      const int mt = wave + i * PI8_WAVES;
      if (mt * 16 >= rows) continue;  // wave-uniform: no row of the tile in this row tile
      float zb[4], t[4];
      logits(P[i], Q[i], B0, sc0, zb);
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zb[j]));
      if (mt < eff_tiles) {  // wave-uniform: rows with an open gate AND an alert today sit in the first row tiles
        load_b(1, B1);
        float ze[4];
        logits(P[i], Q[i], B1, sScale[drow + 1], ze);
        const float4 ga4 = *reinterpret_cast<const float4 *>(&sGa[mt * 16 + 4 * q]);  // accumulator row = 4 q + j
        const float ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] *= 1.0f - __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ze[j])) * ga[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) rs[i][j] = fmaf(t[j], dvalid, rs[i][j]);
    }
  }
}
// sum over the 16 lanes (draw columns) of each lane group, one value per row of the image -> sSum
__device__ __forceinline__ void pi8_reduce(const float (*rs)[4], float *sSum, int wave, int lane) {
  const int q = lane >> 4, c16 = lane & 15;
#pragma unroll
  for (int i = 0; i < PI8_MT_PER_WAVE; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = rs[i][j];
      v = pi8_add_dpp<0xB1>(v);   // lane ^ 1
      v = pi8_add_dpp<0x4E>(v);   // lane ^ 2
      v = pi8_add_dpp<0x141>(v);  // row_half_mirror
      v = pi8_add_dpp<0x140>(v);  // row_mirror
      if (c16 == 0) sSum[(wave + i * PI8_WAVES) * 16 + 4 * q + j] = v;
    }
  }
}
// exact fp64 reward sum of one row for a column outside the fixed-point range (uniform addresses into W)
__device__ __forceinline__ double pi8_exact_row(const float4 (&xf)[ROWF / 4], const float4 *W, uint32_t col, int n_samples,
                                                uint32_t ga) {
  const float *xr = reinterpret_cast<const float *>(xf);
  double sum = 0.0;
  for (int s = 0; s < n_samples; ++s) {
    const float *wr = reinterpret_cast<const float *>(W) + ((size_t)col * n_samples + s) * (2 * ROWF);
    double zb = 0.0, ze = 0.0;
#pragma unroll
    for (int k = 0; k < ROWF; ++k) zb = fma((double)xr[k], (double)wr[k], zb);
    float t = sigmoid_f32((float)zb);
    if (ga) {
#pragma unroll
      for (int k = 0; k < ROWF; ++k) ze = fma((double)xr[k], (double)wr[ROWF + k], ze);
      t *= 1.0f - sigmoid_f32((float)ze);
    }
    sum += (double)t;
  }
  return sum;
}

#ifndef W2A_PI8_MIN_WAVES
#define W2A_PI8_MIN_WAVES 4  // waves/SIMD the kernel is compiled for: 4 workgroups of 40 KB LDS per CU, <= 128 VGPRs
#endif
__global__ __launch_bounds__(PI8_THREADS, W2A_PI8_MIN_WAVES) void k_posterior_mean_i8(const PmI8Args a) {
  // ONE buffer, two lives: first the X image [row][4 planes x 32 slots] (36 KB), and -- once every wave holds its A
  // fragments in registers -- the column's coefficient block [draw][head][4 planes x 32 slots] (30 KB). 40 KB per
  // workgroup instead of 71: four workgroups per CU, so that one wave's MFMAs run under another's epilogue and a
  // workgroup in its set-up (two dependent global loads) leaves the SIMDs busy
  static_assert(PI8_NPAD * PI8_WSTRIDE <= PI8_ROWS * PI8_XSTRIDE, "the coefficient block must fit the X image's buffer");
  __shared__ __attribute__((aligned(16))) uint32_t s_buf[PI8_ROWS * PI8_XSTRIDE];
  uint32_t (*sX)[PI8_XSTRIDE] = reinterpret_cast<uint32_t (*)[PI8_XSTRIDE]>(s_buf);
  uint32_t (*sW)[PI8_WSTRIDE] = reinterpret_cast<uint32_t (*)[PI8_WSTRIDE]>(s_buf);
  __shared__ float sScale[PI8_NPAD * 2];
  __shared__ __attribute__((aligned(16))) float sGa[PI8_ROWS];
  __shared__ float sSum[PI8_ROWS];
  __shared__ uint32_t sEnv[PI8_ROWS];
  __shared__ uint32_t s_wga[PI8_WAVES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t n_tiles = *a.n_tiles, per_xcd = (n_tiles + 7u) >> 3;
  const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per_xcd || tile >= n_tiles) return;
  const uint4 tl = a.tiles[tile];  // first sorted position, rows, column
  const uint32_t col = __builtin_amdgcn_readfirstlane(tl.z);
  const int rows = (int)__builtin_amdgcn_readfirstlane(tl.y);
  const int n_samples = a.p.tb.n_samples;
  // ---- the column's int8 block, 16 uint4 per draw, through registers (7 x 16 B per thread) into LDS
  constexpr int WREG = PI8_NPAD * 16 / PI8_THREADS;  // 7 x 16 B per thread
  uint4 wreg[WREG];
  float screg = 0.0f;
  auto stage_load = [&](int n0, int draws) {
    const uint4 *src = reinterpret_cast<const uint4 *>(a.wq + ((size_t)col * n_samples + n0) * (2 * ROWF));
#pragma unroll
    for (int i = 0; i < WREG; ++i) {
      const int idx = tid + i * PI8_THREADS;
      wreg[i] = idx < draws * 16 ? src[idx] : make_uint4(0u, 0u, 0u, 0u);  // draws past the end: zeros
    }
    screg = tid < draws * 2 ? a.wscale[((size_t)col * n_samples + n0) * 2 + tid] : 0.0f;
  };
#ifndef W2A_PI8_PREFETCH_W
#define W2A_PI8_PREFETCH_W 0  // 1: request the block before the record -> feature row chain (28 more live VGPRs)
#endif
  if (W2A_PI8_PREFETCH_W) stage_load(0, min(PI8_NPAD, n_samples));
  // ---- lane = row: record and feature row of the day
  uint4 rec = make_uint4(0u, 0u, 0u, 0xFFFFFFFFu);
  if (tid < rows) rec = PM_REC(a.p, tl.x + tid);
  float4 xf[ROWF / 4];
  {
    const float4 *xp = a.p.tb.X + (rec.x >> 2);
#pragma unroll
    for (int q = 0; q < ROWF / 4; ++q) xf[q] = xp[q];
    xf[RT_QUAD] = make_float4((float)(rec.y & 1u), (float)((rec.y >> 1) & 1023u), (float)(rec.y >> 16),
                              (float)((rec.y >> 11) & 15u));
  }
  const uint32_t ga = (tid < rows) ? (rec.y >> 15) & 1u : 0u;
  if (a.colflag[col]) {
    // ---- exact path for a column outside the fixed-point range: fp64 dot products straight from W (uniform addresses)
    if (tid < rows)
      a.p.reward[rec.w] = (float)(-(1000.0 / 152.0) * pi8_exact_row(xf, a.p.tb.W, col, n_samples, ga) / (double)n_samples);
    return;
  }
  // ---- rows with gate * actual = 1 first (stable partition inside the workgroup)
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
  sEnv[pos] = rec.w;
  sGa[pos] = (float)ga;
  // ---- fixed-point digits of the row -> LDS image
  pi8_store_row(xf, a.xs, sX[pos]);
  const int eff_tiles = (int)((n_eff + 15u) >> 4);
  float rs[PI8_MT_PER_WAVE][4];  // per lane: sums over its draw columns of the 4 accumulator rows
#pragma unroll
  for (int i = 0; i < PI8_MT_PER_WAVE; ++i) rs[i][0] = rs[i][1] = rs[i][2] = rs[i][3] = 0.0f;
  pi8_v4i P[PI8_MT_PER_WAVE], Q[PI8_MT_PER_WAVE];
  __syncthreads();  // the X image is complete
  pi8_load_a(sX, wave, lane, P, Q);
  for (int n0 = 0; n0 < n_samples; n0 += PI8_NPAD) {
    const int draws = min(PI8_NPAD, n_samples - n0);
    const int ntiles = (draws + 15) >> 4;
    __syncthreads();  // every wave has read its A fragments (first pass) / previous users of sW are done
    if (n0 > 0 || !W2A_PI8_PREFETCH_W) stage_load(n0, draws);
#pragma unroll
    for (int i = 0; i < WREG; ++i) {  // into rows of PI8_WSTRIDE words
      const int idx = tid + i * PI8_THREADS;
      *reinterpret_cast<uint4 *>(&sW[idx >> 4][4 * (idx & 15)]) = wreg[i];
    }
    if (tid < PI8_NPAD * 2) sScale[tid] = screg;
    __syncthreads();
    pi8_accumulate(sW, sScale, sGa, P, Q, rs, ntiles, draws, rows, eff_tiles, wave, lane);
  }
  pi8_reduce(rs, sSum, wave, lane);
  __syncthreads();
  if (tid < rows) a.p.reward[sEnv[tid]] = -(1000.0f / 152.0f) * sSum[tid] / (float)n_samples;
}

#endif  // W2A_POSTERIOR_I8_HIP_H
