// exp_sort_configs.hip -- which rocprim onesweep configuration sorts 1 048 576 (key, index) pairs of 17 key bits fastest?
// (the fused sorted reset's sort: w2a_reset_device_rng_sorted; profiles/r06/exp_sort_configs.log)
//   hipcc -O3 --offload-arch=gfx950 tools/exp_sort_configs.hip -o gpurun_out/exp_sort_configs && gpurun_out/exp_sort_configs
#include <hip/hip_runtime.h>
#include <string.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class Cfg>
static void run(const char *name, const uint32_t *k_in, uint32_t *k_out, const uint32_t *v_in, uint32_t *v_out, size_t n, unsigned bits,
                const std::vector<uint32_t> &want) {
  size_t bytes = 0;
  CK(rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, k_in, k_out, v_in, v_out, n, 0u, bits, (hipStream_t)0));
  void *tmp;
  CK(hipMalloc(&tmp, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) CK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, k_in, k_out, v_in, v_out, n, 0u, bits, (hipStream_t)0));
  CK(hipEventRecord(e0, 0));
  const int reps = 50;
  for (int i = 0; i < reps; ++i) CK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, k_in, k_out, v_in, v_out, n, 0u, bits, (hipStream_t)0));
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint32_t> got(n);
  CK(hipMemcpy(got.data(), v_out, 4 * n, hipMemcpyDeviceToHost));
  printf("%-44s bits %2u: %7.1f us per sort, tmp %8zu B, %s\n", name, bits, 1000.0f * ms / reps, bytes, got == want ? "stable order OK" : "WRONG ORDER");
  fflush(stdout);
  CK(hipFree(tmp));
}

int main() {
  const size_t n = 1048576;
  using namespace rocprim;
  for (unsigned rows : {74600u, 72000u}) {
    unsigned bits = 1;
    while ((rows - 1) >> bits) ++bits;
    std::vector<uint32_t> k(n), v(n), want(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; k[i] = (uint32_t)(s % rows); v[i] = (uint32_t)i; }
    std::iota(want.begin(), want.end(), 0u);
    std::stable_sort(want.begin(), want.end(), [&](uint32_t a, uint32_t b) { return k[a] < k[b]; });
    uint32_t *k_in, *k_out, *v_in, *v_out;
    CK(hipMalloc(&k_in, 4 * n)); CK(hipMalloc(&k_out, 4 * n)); CK(hipMalloc(&v_in, 4 * n)); CK(hipMalloc(&v_out, 4 * n));
    CK(hipMemcpy(k_in, k.data(), 4 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(v_in, v.data(), 4 * n, hipMemcpyHostToDevice));
    printf("rows %u\n", rows);
    using D = default_config;
    run<radix_sort_config<D, D, D, 0>>("library default onesweep", k_in, k_out, v_in, v_out, n, bits, want);
    run<radix_sort_config<D, D, D, 0>>("library default onesweep", k_in, k_out, v_in, v_out, n, 28, want);
#define OS(HB, HI, SB, SI, RB) \
  run<radix_sort_config<D, D, radix_sort_onesweep_config<kernel_config<HB, HI>, kernel_config<SB, SI>, RB, block_radix_rank_algorithm::match>, 0>>( \
      "hist<" #HB "," #HI "> sort<" #SB "," #SI "> radix " #RB, k_in, k_out, v_in, v_out, n, bits, want)
    OS(1024, 8, 1024, 8, 8);
    OS(1024, 8, 1024, 8, 9);
    OS(1024, 4, 1024, 4, 9);
    OS(512, 8, 512, 8, 9);
    OS(256, 8, 256, 8, 9);
    OS(1024, 6, 1024, 6, 9);
    OS(1024, 12, 1024, 12, 9);
    OS(1024, 16, 1024, 16, 9);
    OS(1024, 8, 1024, 12, 9);
    OS(256, 12, 1024, 8, 9);
    OS(512, 8, 512, 8, 6);
    bits += 1;  // 18 bits: still two passes of 9
    OS(1024, 8, 1024, 8, 9);
    bits -= 1;
    CK(hipFree(k_in)); CK(hipFree(k_out)); CK(hipFree(v_in)); CK(hipFree(v_out));
  }
  return 0;
}
