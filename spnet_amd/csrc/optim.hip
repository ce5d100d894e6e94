// Fused multi-tensor optimizer step over the flat parameter buffer.
//
//   Adam                          spnet/models.py:494,537 (Keras form: eps OUTSIDE the bias correction,
//                                 lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed on the host per batch from the
//                                 1-cycle table, spnet/callbacks.py:396-399)
//   add_regularization l2(1e-4)   spnet/models.py:47-71: the first `l2_n` elements of the flat buffer
//                                 are the 10 regularised kernels; their penalty gradient 2*l2*w is folded
//                                 into g, and sum(w^2) over them is reduced on the fly for the logged loss.
//   grad_scale                    1/world_size after the RCCL all-reduce(sum) of the gradients.
//   mask (optional)               flat 0/1 array: elements with 0 are frozen (layer.trainable=False,
//                                 spnet/models.py:361-373), their p/m/v are left untouched.
// 7 x 4 bytes of HBM traffic per parameter (read p,g,m,v; write p,m,v): pure bandwidth.
#include "common.h"

typedef float adam_v4 __attribute__((ext_vector_type(4)));
// g, m, v are streams nobody reads again before the next step has pushed them out of every cache (m + v + g = 600 MB at the
// benchmark's 50 M parameters): loaded and stored past the caches, so that p -- which the next forward reads -- is what
// stays.
__device__ __forceinline__ float4 adam_ld_stream(const float* q) {
  const adam_v4 t = __builtin_nontemporal_load(reinterpret_cast<const adam_v4*>(q));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void adam_st_stream(float* q, float a, float b, float c, float d) {
  __builtin_nontemporal_store(adam_v4{a, b, c, d}, reinterpret_cast<adam_v4*>(q));
}

__global__ __launch_bounds__(256) void adam_l2_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      long n, long l2_n, float lr_t, float beta1,
                                                      float beta2, float eps, float l2,
                                                      float grad_scale, const float* __restrict__ mask,
                                                      float* __restrict__ sq_partial,
                                                      const float* __restrict__ lr_dev) {
  __shared__ float red[4];
  if (lr_dev) lr_t = lr_dev[0];   // step size kept in device memory so a captured hipGraph can be replayed
  const long n4 = n >> 2;
  float sq = 0.f;
  const float twol2 = 2.f * l2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    float4 pv = *reinterpret_cast<float4*>(p + e);
    float4 gv = adam_ld_stream(g + e);
    float4 mv = adam_ld_stream(m + e);
    float4 vv = adam_ld_stream(v + e);
    float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
    float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gj = ga[j] * grad_scale;
      if (e + j < l2_n) {
        sq = fmaf(pa[j], pa[j], sq);
        gj = fmaf(twol2, pa[j], gj);
      }
      ma[j] = beta1 * ma[j] + (1.f - beta1) * gj;
      va[j] = beta2 * va[j] + (1.f - beta2) * gj * gj;
      pa[j] = pa[j] - lr_t * ma[j] / (sqrtf(va[j]) + eps);
    }
    if (mask) {   // frozen layers (trainable=False): leave p, m, v untouched where mask == 0
      const float4 k = *reinterpret_cast<const float4*>(mask + e);
      if (k.x == 0.f) { pa[0] = pv.x; ma[0] = mv.x; va[0] = vv.x; }
      if (k.y == 0.f) { pa[1] = pv.y; ma[1] = mv.y; va[1] = vv.y; }
      if (k.z == 0.f) { pa[2] = pv.z; ma[2] = mv.z; va[2] = vv.z; }
      if (k.w == 0.f) { pa[3] = pv.w; ma[3] = mv.w; va[3] = vv.w; }
    }
    *reinterpret_cast<float4*>(p + e) = make_float4(pa[0], pa[1], pa[2], pa[3]);
    adam_st_stream(m + e, ma[0], ma[1], ma[2], ma[3]);
    adam_st_stream(v + e, va[0], va[1], va[2], va[3]);
  }
  sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0 && sq_partial) sq_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ in, int n,
                                                           float scale, float* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)in[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) * (double)scale);
}

#define ADAM_BLOCKS 2048

static int adam_grid(long n) {
  long g4 = (n / 4 + 255) / 256;
  return (int)(g4 > ADAM_BLOCKS ? ADAM_BLOCKS : (g4 < 1 ? 1 : g4));
}

// The optimizer step over a RANGE of the flat buffer (round 5): the caller passes pointers to the range's first element,
// its length n (multiple of 4) and l2_n = how many of its leading elements are l2-regularised.  No reduction of the
// sum-of-squares partials: spnet_adam_parts(n) floats are left in sq_partial, spnet_adam_l2_sum folds the partials of all
// ranges of a step.  Lets the Dense head's 73 % of the parameters be updated as soon as their gradient is final -- on a
// side stream underneath the backbone's backward -- while the rest waits for the end of backward.
extern "C" long spnet_adam_parts(long n) { return n < 4 ? 0 : adam_grid(n); }
extern "C" int spnet_adam_part(float* p, const float* g, float* m, float* v, long n, long l2_n, float lr_t, float beta1,
                               float beta2, float eps, float l2, float grad_scale, const float* mask, float* sq_partial,
                               const float* lr_t_dev, void* stream) {
  if (n < 4 || (n & 3) || !p || !g || !m || !v) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(adam_l2_kernel, dim3(adam_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, l2_n, lr_t, beta1,
                     beta2, eps, l2, grad_scale, mask, sq_partial, lr_t_dev);
  SPNET_RETURN_LAUNCH_STATUS();
}
// l2_loss_out[0] = l2 * sum of `count` partials (double accumulation, fixed order)
extern "C" int spnet_adam_l2_sum(const float* sq_partial, int count, float l2, float* l2_loss_out, void* stream) {
  if (!sq_partial || !l2_loss_out || count < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sq_partial, count, l2, l2_loss_out);
  SPNET_RETURN_LAUNCH_STATUS();
}

// n must be a multiple of 4 (the flat buffer is padded).  sq_scratch: ADAM_BLOCKS floats.
// l2_loss_out[0] = l2 * sum_{i<l2_n} p_i^2 evaluated BEFORE the update (the penalty of this step's loss).
extern "C" int spnet_adam_step(float* p, const float* g, float* m, float* v, long n, long l2_n,
                               float lr_t, float beta1, float beta2, float eps, float l2,
                               float grad_scale, const float* mask, float* sq_scratch,
                               float* l2_loss_out, const float* lr_t_dev, void* stream) {
  if (n & 3) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  long g4 = (n / 4 + 255) / 256;
  int grid = (int)(g4 > ADAM_BLOCKS ? ADAM_BLOCKS : (g4 < 1 ? 1 : g4));
  hipLaunchKernelGGL(adam_l2_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, n, l2_n, lr_t, beta1,
                     beta2, eps, l2, grad_scale, mask, sq_scratch, lr_t_dev);
  if (sq_scratch && l2_loss_out)
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, sq_scratch, grid, l2, l2_loss_out);
  SPNET_RETURN_LAUNCH_STATUS();
}
