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

// Keras' moving-statistics update, moving*momentum + value*(1 - momentum), with every product and the sum rounded on
// its own (no fused multiply-add): the same bits from every kernel that performs it.
__device__ __forceinline__ float bn_moving_update(float moving, float momentum, float value) {
  return __fadd_rn(__fmul_rn(momentum, moving), __fmul_rn(1.f - momentum, value));
}

// shift = beta - mean*scale of the BatchNorm affine, likewise without contraction.
__device__ __forceinline__ float bn_shift(float beta, float mean, float scale) {
  return __fsub_rn(beta, __fmul_rn(mean, scale));
}

// Batch statistics of one channel from its column sums (sum, sum of squares over M values) and everything the BatchNorm
// forward derives from them.  Three kernels perform this step (bn_fwd_finalize_kernel, and the consumers it is folded
// into: bn_fwd_fused_vec_kernel, dw3x3_tile_fwd_kernel); every rounding is pinned (no contraction left to the
// compiler) so that they produce the same bits.
struct BnChannelStats {
  float mean, invstd, scale, shift, unbiased_var;
};
__device__ __forceinline__ BnChannelStats bn_channel_stats(double s, double q, long M, float gamma, float beta, float eps) {
  BnChannelStats r;
  const double mean = s / (double)M;
  double var = __dsub_rn(q / (double)M, __dmul_rn(mean, mean));
  if (var < 0.0) var = 0.0;
  r.mean = (float)mean;
  r.invstd = (float)(1.0 / sqrt(__dadd_rn(var, (double)eps)));
  r.scale = __fmul_rn(gamma, r.invstd);
  r.shift = bn_shift(beta, r.mean, r.scale);
  r.unbiased_var = (float)((M > 1) ? __dmul_rn(var, (double)M / (double)(M - 1)) : var);
  return r;
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
