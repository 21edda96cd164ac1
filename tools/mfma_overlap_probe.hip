// mfma_overlap_probe.hip -- does fp64 MFMA work overlap other vector work on MI355X? (posterior-mean
// reward kernel.) Three loops with the same trip count, 4 waves per SIMD, no memory traffic:
//   mfma : 8 independent v_mfma_f64_16x16x4_f64 per iteration
//   valu : the f32 sigmoid epilogue of 8 accumulator values per iteration (v_cvt_f32_f64, v_exp_f32, v_rcp_f32, adds)
//   both : the two interleaved in one wave, independent of each other
//   fma64: 32 v_fma_f64 per iteration (the vector unit's fp64 rate, for reference)
//   fmadpp: 32 v_fmac_f64_dpp row_newbcast per iteration (8 chains);  fmaasm: 32 v_fmac_f64 without DPP
//   dpp1ch / dpp2ch: the same 32 DPP FMAs as one / two dependent chains
// If time(both) ~ max(mfma, valu) the units overlap; if ~ mfma + valu they share the issue slot / the datapath.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void k_probe(double *out, int iters, double seed) {
  const int lane = threadIdx.x & 63;
  d4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = d4{seed * (i + 1), seed, seed, seed};
  const double a = seed + lane * 1e-3, b = seed - lane * 1e-3;
  float s[8];
  double f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.0f; f[i] = seed * i; }
  double x = seed * lane;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float z = (float)x + s[i];
        s[i] += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
      }
      x += 1e-9;
    }
    if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = fma(f[i], a, b);
    }
    if (MODE == 4) {  // 32 v_fmac_f64_dpp row_newbcast (the posterior-mean kernel's FMA form), 8 chains
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(f[i]) : "v"(a), "v"(b));
    }
    if (MODE == 6) {  // ONE chain of 32 dependent v_fmac_f64_dpp: is the latency covered by the SIMD's 4 waves?
#pragma unroll
      for (int r = 0; r < 32; ++r)
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(f[0]) : "v"(a), "v"(b));
    }
    if (MODE == 7) {  // two chains
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(f[0]) : "v"(a), "v"(b));
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(f[1]) : "v"(a), "v"(b));
      }
    }
    if (MODE == 5) {  // 32 v_fmac_f64 with plain VGPR operands through the same asm path
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(f[i]) : "v"(a), "v"(b));
    }
  }
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (double)s[i] + f[i];
  if (r == 1.2345e-300) out[0] = r;
}

template <int MODE>
static float run(const char *name, double *out, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int grid = 256 * 4;  // one workgroup of 4 waves per SIMD slot: 4 waves per SIMD
  hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double cyc = ms * 1e-3 * 2.4e9 / iters / 4.0;  // per iteration and wave at 2.4 GHz, 4 waves sharing a SIMD
  printf("%-6s %8.1f us   ~%6.1f cycles per wave-iteration (at 2.4 GHz)\n", name, ms * 1e3, cyc);
  return ms;
}

int main() {
  double *out;
  CHECK(hipMalloc(&out, 64));
  const int iters = 20000;
  const float m = run<0>("mfma", out, iters);
  const float v = run<1>("valu", out, iters);
  const float b = run<2>("both", out, iters);
  run<3>("fma64", out, iters);
  run<4>("fmadpp", out, iters);
  run<5>("fmaasm", out, iters);
  run<6>("dpp1ch", out, iters);
  run<7>("dpp2ch", out, iters);
  printf("both / (mfma + valu) = %.2f   both / max = %.2f\n", b / (m + v), b / (m > v ? m : v));
  return 0;
}
