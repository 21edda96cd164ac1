// w2a_sort.hip.h -- relabelling sort for episode_order="sorted"
// Part of libw2a.so; included only by w2a_kernels.hip (one translation unit, see the file comment there).
#ifndef W2A_SORT_HIP_H
#define W2A_SORT_HIP_H

// ----------------------------------------------------------------------------------------
// episode_order="sorted": relabel envs so that neighbours share coefficient rows
// ----------------------------------------------------------------------------------------
__global__ void k_sort_keys(const uint4 *cold, uint64_t *keys, uint32_t *idx, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4 c = cold[i];
  keys[i] = ((uint64_t)c.y << 32) | c.x;  // coefficient row (column, draw) in the high word: the bits that are sorted
  idx[i] = (uint32_t)i;
}
__global__ void k_permute_state(StateArrays src, const uint32_t *idx, StateArrays dst, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t j = idx[i];
  dst.cold[i] = src.cold[j];
  dst.hot3[i] = src.hot3[j];
  dst.stepc[i] = src.stepc[j];
}

#endif  // W2A_SORT_HIP_H
