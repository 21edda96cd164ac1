// w2a_logit_table.hip.h -- k_logit_table: grouped fp64-MFMA reward precompute; k_pack_wendo
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_LOGIT_TABLE_HIP_H
#define W2A_LOGIT_TABLE_HIP_H

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

#endif  // W2A_LOGIT_TABLE_HIP_H
