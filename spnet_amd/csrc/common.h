// Shared device/host helpers for the SPNet gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPNET_WAVE 64

static inline int spnet_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Every extern "C" entry point returns 0 on success or the hipError_t of the failed launch.
#define SPNET_RETURN_LAUNCH_STATUS()            \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    return (int)e__;                            \
  } while (0)

// Memory-bound elementwise kernels: cap the grid and grid-stride the rest (256 CUs x 8 blocks).
static inline int spnet_ew_grid(long n_items, int block) {
  long g = (n_items + block - 1) / block;
  if (g > 2048 * 4) g = 2048 * 4;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}

// XCD-aware block remap (bijective for any grid size): blocks that the dispatcher deals to the same
// XCD (b % 8) get a contiguous range of logical ids, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int nx = 8;
  int q = nblk / nx, r = nblk % nx;
  int x = bid % nx, i = bid / nx;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + i;
}
