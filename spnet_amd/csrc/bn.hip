// BatchNormalization(axis=-1, momentum 0.99, eps 1e-3) forward / backward on [M][C] fp32 tensors
// (M = batch*H*W pixels).  Keras semantics (call sites spnet/models.py:326-336 and the 40 BN layers
// inside keras.applications.Xception): training normalises with the batch mean and the biased batch
// variance; moving statistics are updated with the Bessel-corrected variance.
//
// All reductions are two-stage and deterministic: per-workgroup partial sums, then a finalize kernel
// that combines them in a fixed order in double precision.
//
// Activation fused behind the affine: 0 none, 1 ReLU, 2 LeakyReLU(0.1), 3 ReLU6 (keras.applications.mobilenet.relu6).
#include "x3t.h"

#define BN_MAX_PARTS 256

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return v > 0.f ? v : 0.1f * v;
  if (act == 3) return fminf(fmaxf(v, 0.f), 6.f);
  return v;
}
__device__ __forceinline__ float act_grad(float out_pre, int act) {
  if (act == 1) return out_pre > 0.f ? 1.f : 0.f;
  if (act == 2) return out_pre > 0.f ? 1.f : 0.1f;
  if (act == 3) return (out_pre > 0.f && out_pre < 6.f) ? 1.f : 0.f;
  return 1.f;
}

// ---------------------------------------------------------------- vector path (C % 4 == 0)
// blockDim = (CL, 256/CL); grid = (ceil(C4/CL), GY).  MODE 0: sum x, sum x^2.
// MODE 1: sum g, sum g*xhat with g = dy * act'(xhat*gamma+beta).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_vec_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, long M, int C,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act,
    float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float4 red4[];   // [blockDim.y][2][blockDim.x]
  const int c4n = C >> 2;
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = c4 < c4n;
  const int c = c4 * 4;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (active) {
    float4 mu = s0, is = s0, ga = s0, be = s0;
    if (MODE == 1) {
      mu = *reinterpret_cast<const float4*>(mean + c);
      is = *reinterpret_cast<const float4*>(invstd + c);
      ga = *reinterpret_cast<const float4*>(gamma + c);
      be = *reinterpret_cast<const float4*>(beta + c);
    }
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < M;
         r += (long)gridDim.y * blockDim.y) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * C + c);
      if (MODE == 0) {
        s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
        s1.x = fmaf(v.x, v.x, s1.x); s1.y = fmaf(v.y, v.y, s1.y);
        s1.z = fmaf(v.z, v.z, s1.z); s1.w = fmaf(v.w, v.w, s1.w);
      } else {
        float4 g = *reinterpret_cast<const float4*>(dy + r * C + c);
        float4 xh;
        xh.x = (v.x - mu.x) * is.x; xh.y = (v.y - mu.y) * is.y;
        xh.z = (v.z - mu.z) * is.z; xh.w = (v.w - mu.w) * is.w;
        if (act) {
          g.x *= act_grad(fmaf(xh.x, ga.x, be.x), act);
          g.y *= act_grad(fmaf(xh.y, ga.y, be.y), act);
          g.z *= act_grad(fmaf(xh.z, ga.z, be.z), act);
          g.w *= act_grad(fmaf(xh.w, ga.w, be.w), act);
        }
        s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
        s1.x = fmaf(g.x, xh.x, s1.x); s1.y = fmaf(g.y, xh.y, s1.y);
        s1.z = fmaf(g.z, xh.z, s1.z); s1.w = fmaf(g.w, xh.w, s1.w);
      }
    }
  }
  const int bx = blockDim.x, by = blockDim.y;
  red4[(threadIdx.y * 2 + 0) * bx + threadIdx.x] = s0;
  red4[(threadIdx.y * 2 + 1) * bx + threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.y == 0 && active) {
    for (int q = 0; q < 2; ++q) {
      float4 s = red4[q * bx + threadIdx.x];
      for (int y = 1; y < by; ++y) {
        const float4 v = red4[(y * 2 + q) * bx + threadIdx.x];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.y * 2 + q) * C + c) = s;
    }
  }
}

// ---------------------------------------------------------------- scalar path (tiny C, e.g. the stem's 3)
// blockDim = 64*C, grid-stride over the flat [M*C] array; a thread's channel is tid % C throughout.
template <int MODE>
__global__ void bn_partial_small_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                        long M, int C, const float* __restrict__ mean,
                                        const float* __restrict__ invstd,
                                        const float* __restrict__ gamma,
                                        const float* __restrict__ beta, int act,
                                        float* __restrict__ partial) {
  extern __shared__ float red[];   // [2][blockDim.x]
  const int ch = threadIdx.x % C;
  float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
  if (MODE == 1) { mu = mean[ch]; is = invstd[ch]; ga = gamma[ch]; be = beta[ch]; }
  float s0 = 0.f, s1 = 0.f;
  const long n = M * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (MODE == 0) {
      s0 += v;
      s1 = fmaf(v, v, s1);
    } else {
      const float xh = (v - mu) * is;
      const float g = dy[i] * act_grad(fmaf(xh, ga, be), act);
      s0 += g;
      s1 = fmaf(g, xh, s1);
    }
  }
  red[threadIdx.x] = s0;
  red[blockDim.x + threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.x < C) {
    float a = 0.f, b = 0.f;
    for (int t = threadIdx.x; t < blockDim.x; t += C) { a += red[t]; b += red[blockDim.x + t]; }
    partial[((long)blockIdx.x * 2 + 0) * C + threadIdx.x] = a;
    partial[((long)blockIdx.x * 2 + 1) * C + threadIdx.x] = b;
  }
}

// C == 3 (the stem): 4 pixels = 12 floats = three float4 whose channel pattern is fixed
// ([c0 c1 c2 c0][c1 c2 c0 c1][c2 c0 c1 c2]), so a thread streams 48-byte groups with vector loads and keeps
// its three channel sums in registers.  M % 4 == 0.  One partial row per workgroup, fixed order:
// thread -> wave (shuffles) -> workgroup (LDS, wave order).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_c3_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ dy, long M,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int act,
                                                            float* __restrict__ partial) {
  __shared__ float red[4][6];
  float mu[3] = {0.f, 0.f, 0.f}, is[3] = {0.f, 0.f, 0.f}, ga[3] = {0.f, 0.f, 0.f}, be[3] = {0.f, 0.f, 0.f};
  if (MODE == 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { mu[c] = mean[c]; is[c] = invstd[c]; ga[c] = gamma[c]; be[c] = beta[c]; }
  }
  float s0[3] = {0.f, 0.f, 0.f}, s1[3] = {0.f, 0.f, 0.f};
  const long groups = M / 4;
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long)gridDim.x * blockDim.x) {
    float v[12], g[12];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(x + gi * 12 + q * 4);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
      if (MODE == 1) {
        const float4 u = *reinterpret_cast<const float4*>(dy + gi * 12 + q * 4);
        g[4 * q] = u.x; g[4 * q + 1] = u.y; g[4 * q + 2] = u.z; g[4 * q + 3] = u.w;
      }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      const int c = e % 3;
      if (MODE == 0) {
        s0[c] += v[e];
        s1[c] = fmaf(v[e], v[e], s1[c]);
      } else {
        const float xh = (v[e] - mu[c]) * is[c];
        const float gg = g[e] * act_grad(fmaf(xh, ga[c], be[c]), act);
        s0[c] += gg;
        s1[c] = fmaf(gg, xh, s1[c]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a = wave_sum(s0[c]), b = wave_sum(s1[c]);
    if (lane == 0) { red[wv][c] = a; red[wv][3 + c] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int q = threadIdx.x / 3, c = threadIdx.x % 3;
    partial[((long)blockIdx.x * 2 + q) * 3 + c] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

// Combine partial [P][2][C] in a fixed order: 256 threads = 16 channels x 16 interleaved groups of
// partial rows, summed in double (4 independent loads in flight per thread), then the 16 group sums
// are added in group order.  Returns the channel owned by this thread (group 0 only) or -1.
#define BN_FIN_CH 16
__device__ __forceinline__ int combine_partials(const float* __restrict__ partial, int P, int C,
                                                double* s_out, double* q_out) {
  __shared__ double cred[2][16][BN_FIN_CH];
  const int lane = threadIdx.x & (BN_FIN_CH - 1), g = threadIdx.x >> 4;
  const int c = blockIdx.x * BN_FIN_CH + lane;
  double s = 0.0, q = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int p = g; p < P; p += 16) {
      s += (double)partial[((long)p * 2 + 0) * C + c];
      q += (double)partial[((long)p * 2 + 1) * C + c];
    }
  }
  cred[0][g][lane] = s;
  cred[1][g][lane] = q;
  __syncthreads();
  if (g != 0 || c >= C) return -1;
  double ss = 0.0, qq = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { ss += cred[0][k][lane]; qq += cred[1][k][lane]; }
  *s_out = ss;
  *q_out = qq;
  return c;
}

// ---------------------------------------------------------------- many partial rows: two stages
// The entry-flow GEMMs leave thousands of partial rows (M = 372,000 pixels in 32-row tiles: P = 11,625 -- 12 MB), and a
// finalize grid of C/16 workgroups (8 for 128 channels) walks them in 48 us.  Stage 1 spreads the walk over
// C/16 x S workgroups: slice s = rows [s*L, (s+1)*L) (the last slice takes the remainder) is combined exactly like
// combine_partials() and its two double sums are left IN PLACE as (hi, lo) float pairs -- hi in row s*L, lo in row
// s*L + 1, in the workgroup's own channel columns, which no other workgroup reads -- so no extra workspace is needed
// and hi + lo carries 48 significant bits.  Stage 2 (combine_slices) adds the S slice sums in the same 16-group order.
// Fixed order throughout: deterministic, but not the order of the one-stage walk (used from BN_SLICE_MIN_P rows on; the
// folded finalize kernels handle at most 128 rows and stay bit-identical to the one-stage kernel).
#define BN_SLICE_MIN_P 1024
#define BN_SLICES 64
__global__ __launch_bounds__(256) void bn_slice_partials_kernel(float* __restrict__ partial, int P, int C, int L, int S) {
  __shared__ double cred[2][16][BN_FIN_CH];
  const int lane = threadIdx.x & (BN_FIN_CH - 1), g = threadIdx.x >> 4;
  const int c = blockIdx.x * BN_FIN_CH + lane;
  const int sl = blockIdx.y;
  const int p0 = sl * L, p1 = (sl == S - 1) ? P : p0 + L;
  double s = 0.0, q = 0.0;
  if (c < C) {
#pragma unroll 4
    for (int p = p0 + g; p < p1; p += 16) {
      s += (double)partial[((long)p * 2 + 0) * C + c];
      q += (double)partial[((long)p * 2 + 1) * C + c];
    }
  }
  cred[0][g][lane] = s;
  cred[1][g][lane] = q;
  __syncthreads();                                   // every read of this workgroup's columns is done
  if (g != 0 || c >= C) return;
  double ss = 0.0, qq = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { ss += cred[0][k][lane]; qq += cred[1][k][lane]; }
  const float sh = (float)ss, qh = (float)qq;
  partial[((long)p0 * 2 + 0) * C + c] = sh;
  partial[((long)p0 * 2 + 1) * C + c] = qh;
  partial[((long)(p0 + 1) * 2 + 0) * C + c] = (float)(ss - (double)sh);
  partial[((long)(p0 + 1) * 2 + 1) * C + c] = (float)(qq - (double)qh);
}

// Stage 2: the S slice sums (hi in row s*L, lo in row s*L + 1) -> the channel's two sums, 16 interleaved groups.
__device__ __forceinline__ int combine_slices(const float* __restrict__ partial, int S, int L, int C,
                                              double* s_out, double* q_out) {
  __shared__ double cred2[2][16][BN_FIN_CH];
  const int lane = threadIdx.x & (BN_FIN_CH - 1), g = threadIdx.x >> 4;
  const int c = blockIdx.x * BN_FIN_CH + lane;
  double s = 0.0, q = 0.0;
  if (c < C) {
    for (int k = g; k < S; k += 16) {
      const long r = (long)k * L;
      s += (double)partial[(r * 2 + 0) * C + c] + (double)partial[((r + 1) * 2 + 0) * C + c];
      q += (double)partial[(r * 2 + 1) * C + c] + (double)partial[((r + 1) * 2 + 1) * C + c];
    }
  }
  cred2[0][g][lane] = s;
  cred2[1][g][lane] = q;
  __syncthreads();
  if (g != 0 || c >= C) return -1;
  double ss = 0.0, qq = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { ss += cred2[0][k][lane]; qq += cred2[1][k][lane]; }
  *s_out = ss;
  *q_out = qq;
  return c;
}

// ---------------------------------------------------------------- finalize (16 channels per workgroup)
// Forward: partial [P][2][C] -> batch mean / invstd, affine coefficients, moving-stat update.
// L > 0: `partial` holds P = S slice sums left by bn_slice_partials_kernel (rows of L apart).
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(
    const float* __restrict__ partial, int P, int C, long M, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ moving_mean, float* __restrict__ moving_var,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, float* __restrict__ scale,
    float* __restrict__ shift, float eps, float momentum, int L) {
  double s, q;
  const int c = L > 0 ? combine_slices(partial, P, L, C, &s, &q) : combine_partials(partial, P, C, &s, &q);
  if (c < 0) return;
  const BnChannelStats st = bn_channel_stats(s, q, M, gamma[c], beta[c], eps);
  save_mean[c] = st.mean;
  save_invstd[c] = st.invstd;
  scale[c] = st.scale;
  shift[c] = st.shift;
  moving_mean[c] = bn_moving_update(moving_mean[c], momentum, st.mean);
  moving_var[c] = bn_moving_update(moving_var[c], momentum, st.unbiased_var);
}

// Inference: affine from the moving statistics.
__global__ __launch_bounds__(256) void bn_infer_coeffs_kernel(
    int C, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ moving_mean, const float* __restrict__ moving_var,
    float* __restrict__ scale, float* __restrict__ shift, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * rsqrtf(moving_var[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - moving_mean[c] * sc;
}

// Backward: partial [P][2][C] -> dgamma, dbeta and the three per-channel coefficients of
//   dx = k1 * g + k2 * xhat + k3,  g = dy * act'(.)
// BLEND: the coefficients are written in the form the GEMM operand blend consumes (TileStage XF),
//   dx = k1 * g + k2 * x + k3  on the RAW pre-normalisation x  (k2' = k2*invstd, k3' = k3 - k2*invstd*mean),
// as three arrays `cld` floats apart; otherwise dx = k1 * g + k2 * xhat + k3 (bn_bwd_apply kernels).
template <int BLEND>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    const float* __restrict__ partial, int P, int C, long M, const float* __restrict__ gamma,
    const float* __restrict__ invstd, const float* __restrict__ mean, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ k1, float* __restrict__ k2, float* __restrict__ k3) {
  double sg, sgx;
  const int c = combine_partials(partial, P, C, &sg, &sgx);
  if (c < 0) return;
  dbeta[c] = (float)sg;
  dgamma[c] = (float)sgx;
  const double is = (double)invstd[c];
  const double a = (double)gamma[c] * is;
  const double b = -a * sgx / (double)M, d = -a * sg / (double)M;
  k1[c] = (float)a;
  if (BLEND) {
    k2[c] = (float)(b * is);
    k3[c] = (float)(d - b * is * (double)mean[c]);
  } else {
    k2[c] = (float)b;
    k3[c] = (float)d;
  }
}

// ---------------------------------------------------------------- apply kernels
// y = act(x*scale + shift) (+ residual)
__global__ __launch_bounds__(256) void bn_apply_vec_kernel(const float* __restrict__ x, long M, int C,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int act,
                                                           const float* __restrict__ residual,
                                                           float* __restrict__ y, long ldy) {
  const int c4n = C >> 2;
  const long total = M * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    float4 o;
    o.x = act_fwd(fmaf(v.x, sc.x, sh.x), act);
    o.y = act_fwd(fmaf(v.y, sc.y, sh.y), act);
    o.z = act_fwd(fmaf(v.z, sc.z, sh.z), act);
    o.w = act_fwd(fmaf(v.w, sc.w, sh.w), act);
    if (residual) {
      const float4 r = *reinterpret_cast<const float4*>(residual + i * 4);
      o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
    }
    // (ldy != C: the output is a column block of a wider tensor -- a branch written straight into its Concatenate)
    *reinterpret_cast<float4*>(y + (ldy == C ? i * 4 : (i / c4n) * ldy + c)) = o;
  }
}

__global__ __launch_bounds__(256) void bn_apply_scalar_kernel(const float* __restrict__ x, long n,
                                                              int C, const float* __restrict__ scale,
                                                              const float* __restrict__ shift, int act,
                                                              const float* __restrict__ residual,
                                                              int res_bcast, float* __restrict__ y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    float o = act_fwd(fmaf(x[i], scale[c], shift[c]), act);
    if (residual) o += residual[res_bcast ? i / C : i];   // res_bcast: one residual value per pixel
    y[i] = o;
  }
}

// dx = k1*g + k2*xhat + k3
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, long M, int C,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act,
    const float* __restrict__ k1, const float* __restrict__ k2, const float* __restrict__ k3,
    float* __restrict__ dx, unsigned short* __restrict__ planes, long plane_stride) {
  // planes != NULL: dx goes out as the bf16x3 planes of the [M][C] matrix (x3t.h) -- the operand of the pointwise layer's
  // data-gradient and weight-gradient GEMMs -- instead of fp32
  const int c4n = C >> 2;
  const long total = M * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    float4 g = *reinterpret_cast<const float4*>(dy + i * 4);
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 is = *reinterpret_cast<const float4*>(invstd + c);
    float4 xh;
    xh.x = (v.x - mu.x) * is.x; xh.y = (v.y - mu.y) * is.y;
    xh.z = (v.z - mu.z) * is.z; xh.w = (v.w - mu.w) * is.w;
    if (act) {
      const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
      const float4 be = *reinterpret_cast<const float4*>(beta + c);
      g.x *= act_grad(fmaf(xh.x, ga.x, be.x), act);
      g.y *= act_grad(fmaf(xh.y, ga.y, be.y), act);
      g.z *= act_grad(fmaf(xh.z, ga.z, be.z), act);
      g.w *= act_grad(fmaf(xh.w, ga.w, be.w), act);
    }
    const float4 a = *reinterpret_cast<const float4*>(k1 + c);
    const float4 b = *reinterpret_cast<const float4*>(k2 + c);
    const float4 d = *reinterpret_cast<const float4*>(k3 + c);
    float4 o;
    o.x = fmaf(a.x, g.x, fmaf(b.x, xh.x, d.x));
    o.y = fmaf(a.y, g.y, fmaf(b.y, xh.y, d.y));
    o.z = fmaf(a.z, g.z, fmaf(b.z, xh.z, d.z));
    o.w = fmaf(a.w, g.w, fmaf(b.w, xh.w, d.w));
    if (planes) x3t_store4(planes, plane_stride, x3t_off(i / c4n, c, (C + 31) >> 5), o.x, o.y, o.z, o.w);
    else *reinterpret_cast<float4*>(dx + i * 4) = o;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_scalar_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, long n, int C,
    const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int act,
    const float* __restrict__ k1, const float* __restrict__ k2, const float* __restrict__ k3,
    float* __restrict__ dx) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const float xh = (x[i] - mean[c]) * invstd[c];
    const float g = dy[i] * act_grad(fmaf(xh, gamma[c], beta[c]), act);
    dx[i] = fmaf(k1[c], g, fmaf(k2[c], xh, k3[c]));
  }
}

// ---------------------------------------------------------------- finalize folded into the apply pass
// Where a BatchNorm's statistics arrive as FEW partial rows (the 728-channel middle flow, the exit flow: P <= 128),
// the finalize kernel is a dependent 5-7 us launch in front of an elementwise pass that takes 10-13 us.  These kernels
// do both: a workgroup owns 32 channels x a slab of rows, first combines the partial rows of ITS channels -- the
// arithmetic of combine_partials() in the same order, so every workgroup of a channel chunk holds the coefficients the
// finalize kernel would have written, bit for bit -- then streams its slab.  The workgroups of slab 0 publish the
// per-channel results (dgamma / dbeta, or mean / invstd / scale / shift + the moving-statistics update).
// blockDim = 256 = 8 channel quads x 32 rows; grid = (ceil(C/32), slabs).
#define BN_FUSE_CH 32
#define BN_FUSE_AHEAD 3     // rows per thread in flight across the prologue (a 96-row slab entirely)
__device__ __forceinline__ bool chunk_sums(const float* __restrict__ partial, int P, int C, int c0, double* dred,
                                           double* s_out, double* q_out) {
  const int tid = threadIdx.x;
  const int ch = tid % BN_FUSE_CH, grp = tid / BN_FUSE_CH;      // 8 thread groups, two of the 16 row groups each
  const int c = c0 + ch;
  for (int g = grp; g < 16; g += 256 / BN_FUSE_CH) {
    double s = 0.0, q = 0.0;
    if (c < C) {
#pragma unroll 4
      for (int p = g; p < P; p += 16) {
        s += (double)partial[((long)p * 2 + 0) * C + c];
        q += (double)partial[((long)p * 2 + 1) * C + c];
      }
    }
    dred[(0 * 16 + g) * BN_FUSE_CH + ch] = s;
    dred[(1 * 16 + g) * BN_FUSE_CH + ch] = q;
  }
  __syncthreads();
  if (tid >= BN_FUSE_CH || c >= C) return false;
  double ss = 0.0, qq = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { ss += dred[(0 * 16 + k) * BN_FUSE_CH + ch]; qq += dred[(1 * 16 + k) * BN_FUSE_CH + ch]; }
  *s_out = ss;
  *q_out = qq;
  return true;
}

// Training forward: partial -> (mean, invstd, scale, shift, moving statistics) and y = act(x*scale + shift) (+ residual).
__global__ __launch_bounds__(256) void bn_fwd_fused_vec_kernel(
    const float* __restrict__ x, long M, int C, const float* __restrict__ partial, int P,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ moving_mean,
    float* __restrict__ moving_var, float* __restrict__ save_mean, float* __restrict__ save_invstd,
    float* __restrict__ scale, float* __restrict__ shift, float eps, float momentum, int act,
    const float* __restrict__ residual, float* __restrict__ y, int rows_per_slab, int chunks, long ldy) {
  __shared__ double dred[2 * 16 * BN_FUSE_CH];
  __shared__ __attribute__((aligned(16))) float cf[2][BN_FUSE_CH];
  // XCD-aware order: a row of C floats is not a whole number of 128-byte lines (C = 728), so neighbouring channel
  // chunks of a slab share cache lines; the workgroups one XCD receives cover contiguous (slab, chunk) ids and the
  // two halves of such a line meet in ONE L2 instead of being fetched by two.
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = wid % chunks, by = wid / chunks;
  const int c0 = bx * BN_FUSE_CH;
  const int lane = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int c = c0 + lane * 4;
  const long rbeg = (long)by * rows_per_slab;
  const long rend = min(M, rbeg + rows_per_slab);
  // The slab's first rows are fetched BEFORE the coefficients exist: their latency and the reduction of the partial
  // rows (an L2 round trip, two barriers) overlap instead of adding up.
  float4 v0[BN_FUSE_AHEAD], q0[BN_FUSE_AHEAD];
#pragma unroll
  for (int u = 0; u < BN_FUSE_AHEAD; ++u) {
    const long r = rbeg + r0 + 32 * u;
    const bool ok = r < rend && c < C;
    const long i = ok ? r * C + c : 0;
    v0[u] = *reinterpret_cast<const float4*>(x + i);
    q0[u] = residual ? *reinterpret_cast<const float4*>(residual + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  double s, q;
  if (chunk_sums(partial, P, C, c0, dred, &s, &q)) {
    const int cc = c0 + threadIdx.x;
    const BnChannelStats st = bn_channel_stats(s, q, M, gamma[cc], beta[cc], eps);
    cf[0][threadIdx.x] = st.scale;
    cf[1][threadIdx.x] = st.shift;
    if (by == 0) {
      save_mean[cc] = st.mean;
      save_invstd[cc] = st.invstd;
      scale[cc] = st.scale;
      shift[cc] = st.shift;
      moving_mean[cc] = bn_moving_update(moving_mean[cc], momentum, st.mean);
      moving_var[cc] = bn_moving_update(moving_var[cc], momentum, st.unbiased_var);
    }
  }
  __syncthreads();
  if (c >= C) return;
  const float4 sc = *reinterpret_cast<const float4*>(&cf[0][lane * 4]);
  const float4 sh = *reinterpret_cast<const float4*>(&cf[1][lane * 4]);
  auto emit = [&](long r, const float4 v, const float4 rr) {
    float4 o;
    o.x = act_fwd(fmaf(v.x, sc.x, sh.x), act);
    o.y = act_fwd(fmaf(v.y, sc.y, sh.y), act);
    o.z = act_fwd(fmaf(v.z, sc.z, sh.z), act);
    o.w = act_fwd(fmaf(v.w, sc.w, sh.w), act);
    if (residual) { o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
    *reinterpret_cast<float4*>(y + r * ldy + c) = o;
  };
#pragma unroll
  for (int u = 0; u < BN_FUSE_AHEAD; ++u) {
    const long r = rbeg + r0 + 32 * u;
    if (r < rend) emit(r, v0[u], q0[u]);
  }
  for (long r = rbeg + r0 + 32 * BN_FUSE_AHEAD; r < rend; r += 32) {
    const long i = r * C + c;
    const float4 v = *reinterpret_cast<const float4*>(x + i);
    const float4 rr = residual ? *reinterpret_cast<const float4*>(residual + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    emit(r, v, rr);
  }
}

// Backward: partial (sum g, sum g*xhat) -> dgamma, dbeta and dx = k1*g + k2*xhat + k3 (no activation behind the BN:
// g already carries any mask).
__global__ __launch_bounds__(256) void bn_bwd_fused_vec_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, long M, int C, const float* __restrict__ partial, int P,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
    const float* __restrict__ invstd, int act, float* __restrict__ dgamma, float* __restrict__ dbeta,
    float* __restrict__ dx, int rows_per_slab, int chunks, unsigned short* __restrict__ planes, long plane_stride) {
  __shared__ double dred[2 * 16 * BN_FUSE_CH];
  __shared__ __attribute__((aligned(16))) float cf[7][BN_FUSE_CH];   // k1, k2, k3, mean, invstd, gamma, beta
  const int wid = xcd_remap(blockIdx.x, gridDim.x);                  // XCD-aware order (see the forward kernel)
  const int bx = wid % chunks, by = wid / chunks;
  const int c0 = bx * BN_FUSE_CH;
  const int lane = threadIdx.x & 7, r0 = threadIdx.x >> 3;
  const int c = c0 + lane * 4;
  const long rbeg = (long)by * rows_per_slab;
  const long rend = min(M, rbeg + rows_per_slab);
  float4 v0[BN_FUSE_AHEAD], g0[BN_FUSE_AHEAD];      // fetched before the coefficients exist (see the forward kernel)
#pragma unroll
  for (int u = 0; u < BN_FUSE_AHEAD; ++u) {
    const long r = rbeg + r0 + 32 * u;
    const bool ok = r < rend && c < C;
    const long i = ok ? r * C + c : 0;
    v0[u] = *reinterpret_cast<const float4*>(x + i);
    g0[u] = *reinterpret_cast<const float4*>(dy + i);
  }
  double sg, sgx;
  if (chunk_sums(partial, P, C, c0, dred, &sg, &sgx)) {
    const int cc = c0 + threadIdx.x;
    const double is = (double)invstd[cc];
    const double a = (double)gamma[cc] * is;
    const double b = -a * sgx / (double)M, d = -a * sg / (double)M;
    cf[0][threadIdx.x] = (float)a;
    cf[1][threadIdx.x] = (float)b;
    cf[2][threadIdx.x] = (float)d;
    cf[3][threadIdx.x] = mean[cc];
    cf[4][threadIdx.x] = invstd[cc];
    cf[5][threadIdx.x] = gamma[cc];
    cf[6][threadIdx.x] = act ? beta[cc] : 0.f;
    if (by == 0) {
      dbeta[cc] = (float)sg;
      dgamma[cc] = (float)sgx;
    }
  }
  __syncthreads();
  if (c >= C) return;
  const float4 k1 = *reinterpret_cast<const float4*>(&cf[0][lane * 4]);
  const float4 k2 = *reinterpret_cast<const float4*>(&cf[1][lane * 4]);
  const float4 k3 = *reinterpret_cast<const float4*>(&cf[2][lane * 4]);
  const float4 mu = *reinterpret_cast<const float4*>(&cf[3][lane * 4]);
  const float4 is = *reinterpret_cast<const float4*>(&cf[4][lane * 4]);
  const float4 ga = *reinterpret_cast<const float4*>(&cf[5][lane * 4]);
  const float4 be = *reinterpret_cast<const float4*>(&cf[6][lane * 4]);
  auto emit = [&](long r, const float4 v, float4 g) {
    float4 xh, o;
    xh.x = (v.x - mu.x) * is.x; xh.y = (v.y - mu.y) * is.y;
    xh.z = (v.z - mu.z) * is.z; xh.w = (v.w - mu.w) * is.w;
    if (act) {          // an activation behind the BatchNorm: its mask, exactly as bn_bwd_apply_vec_kernel forms it
      g.x *= act_grad(fmaf(xh.x, ga.x, be.x), act);
      g.y *= act_grad(fmaf(xh.y, ga.y, be.y), act);
      g.z *= act_grad(fmaf(xh.z, ga.z, be.z), act);
      g.w *= act_grad(fmaf(xh.w, ga.w, be.w), act);
    }
    o.x = fmaf(k1.x, g.x, fmaf(k2.x, xh.x, k3.x));
    o.y = fmaf(k1.y, g.y, fmaf(k2.y, xh.y, k3.y));
    o.z = fmaf(k1.z, g.z, fmaf(k2.z, xh.z, k3.z));
    o.w = fmaf(k1.w, g.w, fmaf(k2.w, xh.w, k3.w));
    if (planes) x3t_store4(planes, plane_stride, x3t_off(r, c, (C + 31) >> 5), o.x, o.y, o.z, o.w);   // (see bn_bwd_apply_vec_kernel)
    else *reinterpret_cast<float4*>(dx + r * C + c) = o;
  };
#pragma unroll
  for (int u = 0; u < BN_FUSE_AHEAD; ++u) {
    const long r = rbeg + r0 + 32 * u;
    if (r < rend) emit(r, v0[u], g0[u]);
  }
  for (long r = rbeg + r0 + 32 * BN_FUSE_AHEAD; r < rend; r += 32) {
    const long i = r * C + c;
    emit(r, *reinterpret_cast<const float4*>(x + i), *reinterpret_cast<const float4*>(dy + i));
  }
}

#define BN_FUSE_MAX_P 128
// ... and, on small tensors (Inception-ResNet-v2 at batch 16: 9,744 x 32..96), up to 512 rows: there the stand-alone
// finalize is a second dependent launch in front of a pass that is itself at the launch floor, and the 2-16 workgroups
// of the fused form re-reading 512 x 2 x 32 partial sums each is noise.
static inline bool bn_fuse_ok(int P, long M, int C) {
  return P <= BN_FUSE_MAX_P || (P <= 512 && M * C <= (4L << 20));
}
// slabs of at least 64 rows, about eight workgroups per CU (their prologues overlap each other's streaming)
static int bn_fuse_rows_per_slab(long M, int C) {
  const int gx = (C + BN_FUSE_CH - 1) / BN_FUSE_CH;
  long gy = 2048 / gx;
  if (gy < 1) gy = 1;
  long rows = (M + gy - 1) / gy;
  if (rows < 64) rows = 64;
  return (int)((rows + 31) / 32 * 32);
}

// ---------------------------------------------------------------- host side
static int bn_chan_lanes(int c4n) {
  int cl = 8;
  while (cl < c4n && cl < 64) cl <<= 1;
  return cl;
}

static int bn_parts(long M, int C) {
  if (C == 3 && (M & 3) == 0) {
    long g = (M / 4 + 256L * 4 - 1) / (256L * 4);
    if (g > 1024) g = 1024;
    return g < 1 ? 1 : (int)g;
  }
  if (C & 3) {
    long g = (M * C + (64L * C) * 8 - 1) / ((64L * C) * 8);
    if (g > BN_MAX_PARTS) g = BN_MAX_PARTS;
    return g < 1 ? 1 : (int)g;
  }
  const int cl = bn_chan_lanes(C / 4);
  const int by = 256 / cl;
  const int gx = (C / 4 + cl - 1) / cl;
  long gy = (M + (long)by * 8 - 1) / ((long)by * 8);   // >= 8 rows per thread
  long cap = 2048 / gx;
  if (cap > BN_MAX_PARTS) cap = BN_MAX_PARTS;
  if (cap < 1) cap = 1;
  if (gy > cap) gy = cap;
  return gy < 1 ? 1 : (int)gy;
}

// floats of scratch needed by spnet_bn_fwd_train / spnet_bn_bwd for an [M][C] tensor
extern "C" long spnet_bn_ws(long M, int C) { return (long)bn_parts(M, C) * 2 * C; }

template <int MODE>
static void launch_partial(const float* x, const float* dy, long M, int C, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, int act,
                           float* partial, int parts, hipStream_t st) {
  if (C == 3 && (M & 3) == 0) {
    hipLaunchKernelGGL(bn_partial_c3_kernel<MODE>, dim3(parts), dim3(256), 0, st, x, dy, M, mean, invstd, gamma,
                       beta, act, partial);
  } else if (C & 3) {
    const int bd = 64 * C;
    hipLaunchKernelGGL(bn_partial_small_kernel<MODE>, dim3(parts), dim3(bd), 2 * bd * sizeof(float),
                       st, x, dy, M, C, mean, invstd, gamma, beta, act, partial);
  } else {
    const int cl = bn_chan_lanes(C / 4);
    const int by = 256 / cl;
    dim3 grid((C / 4 + cl - 1) / cl, parts), block(cl, by);
    hipLaunchKernelGGL(bn_partial_vec_kernel<MODE>, grid, block, (size_t)by * 2 * cl * sizeof(float4),
                       st, x, dy, M, C, mean, invstd, gamma, beta, act, partial);
  }
}

static void launch_apply(const float* x, long M, int C, const float* scale, const float* shift,
                         int act, const float* residual, int res_bcast, float* y, hipStream_t st, long ldy = 0) {
  if (ldy <= 0) ldy = C;
  if (C & 3) {
    const long n = M * C;
    hipLaunchKernelGGL(bn_apply_scalar_kernel, dim3(spnet_ew_grid(n, 256)), dim3(256), 0, st, x, n, C,
                       scale, shift, act, residual, res_bcast, y);
  } else {
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_apply_vec_kernel, dim3(spnet_ew_grid(n4, 256)), dim3(256), 0, st, x, M, C,
                       scale, shift, act, residual, y, ldy);
  }
}

// Training forward.  Outputs: y, save_mean[C], save_invstd[C] (for backward), moving stats updated
// in place.  scale_shift: 2*C floats of scratch.  workspace: spnet_bn_ws(M,C) floats.
// The `_ld` forms write y with a row stride of ldy floats (>= C, a multiple of 4; C % 4 == 0): the output is a column
// block of a wider tensor -- an inception branch written straight into its Concatenate.
extern "C" int spnet_bn_fwd_train_ld(const float* x, long M, int C, const float* gamma,
                                     const float* beta, float* moving_mean, float* moving_var,
                                     float* save_mean, float* save_invstd, float* scale_shift, int act,
                                     const float* residual, int res_bcast, float* y, long ldy, float eps,
                                     float momentum, float* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if ((C & 3) && C > 4) return (int)hipErrorInvalidValue;
  if (ldy != C && ((C & 3) || (ldy & 3) || ldy < C)) return (int)hipErrorInvalidValue;
  if (res_bcast && !(C & 3)) return (int)hipErrorInvalidValue;
  const int parts = bn_parts(M, C);
  launch_partial<0>(x, nullptr, M, C, nullptr, nullptr, nullptr, nullptr, 0, workspace, parts, st);
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, st, workspace, parts,
                     C, M, gamma, beta, moving_mean, moving_var, save_mean, save_invstd, scale_shift,
                     scale_shift + C, eps, momentum, 0);
  launch_apply(x, M, C, scale_shift, scale_shift + C, act, residual, res_bcast, y, st, ldy);
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_bn_fwd_train(const float* x, long M, int C, const float* gamma,
                                  const float* beta, float* moving_mean, float* moving_var,
                                  float* save_mean, float* save_invstd, float* scale_shift, int act,
                                  const float* residual, int res_bcast, float* y, float eps,
                                  float momentum, float* workspace, void* stream) {
  return spnet_bn_fwd_train_ld(x, M, C, gamma, beta, moving_mean, moving_var, save_mean, save_invstd, scale_shift, act,
                               residual, res_bcast, y, C, eps, momentum, workspace, stream);
}

// Inference forward from the moving statistics.
extern "C" int spnet_bn_fwd_infer_ld(const float* x, long M, int C, const float* gamma, const float* beta,
                                     const float* moving_mean, const float* moving_var,
                                     float* scale_shift, int act, const float* residual, int res_bcast,
                                     float* y, long ldy, float eps, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if ((C & 3) && C > 4) return (int)hipErrorInvalidValue;
  if (res_bcast && !(C & 3)) return (int)hipErrorInvalidValue;
  if (ldy != C && ((C & 3) || (ldy & 3) || ldy < C)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_infer_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, gamma, beta,
                     moving_mean, moving_var, scale_shift, scale_shift + C, eps);
  launch_apply(x, M, C, scale_shift, scale_shift + C, act, residual, res_bcast, y, st, ldy);
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_bn_fwd_infer(const float* x, long M, int C, const float* gamma, const float* beta,
                                  const float* moving_mean, const float* moving_var,
                                  float* scale_shift, int act, const float* residual, int res_bcast,
                                  float* y, float eps, void* stream) {
  return spnet_bn_fwd_infer_ld(x, M, C, gamma, beta, moving_mean, moving_var, scale_shift, act, residual, res_bcast, y, C,
                               eps, stream);
}

// Backward of y = act(BN(x)): dx, dgamma[C], dbeta[C].  x is the BN *input* saved by the forward.
// coeffs: 3*C floats of scratch.  workspace: spnet_bn_ws(M,C) floats.  dx may alias dy.
static bool bn_planes_ok(const void* p, long M, int C) {
  return p && !(((uintptr_t)p) & 15) && !(C & 3) && 3 * x3t_plane_elems(M, C) * 2 < (1L << 31);
}

static int bn_bwd_impl(const float* x, const float* dy, long M, int C, const float* gamma,
                       const float* beta, const float* save_mean, const float* save_invstd,
                       int act, float* dx, unsigned short* planes, float* dgamma, float* dbeta, float* coeffs,
                       float* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if ((C & 3) && C > 4) return (int)hipErrorInvalidValue;
  if (planes && !bn_planes_ok(planes, M, C)) return (int)hipErrorInvalidValue;
  const long ps = x3t_plane_elems(M, C);
  const int parts = bn_parts(M, C);
  launch_partial<1>(x, dy, M, C, save_mean, save_invstd, gamma, beta, act, workspace, parts, st);
  if (!(C & 3) && bn_fuse_ok(parts, M, C)) {      // few partial rows: finalize folded into the apply pass (same results)
    const int rps = bn_fuse_rows_per_slab(M, C);
    const int chunks = (C + BN_FUSE_CH - 1) / BN_FUSE_CH;
    dim3 grid((unsigned)(chunks * ((M + rps - 1) / rps)));
    hipLaunchKernelGGL(bn_bwd_fused_vec_kernel, grid, dim3(256), 0, st, x, dy, M, C, workspace, parts, gamma, beta,
                       save_mean, save_invstd, act, dgamma, dbeta, dx, rps, chunks, planes, ps);
    SPNET_RETURN_LAUNCH_STATUS();
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel<0>, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, st, workspace, parts,
                     C, M, gamma, save_invstd, save_mean, dgamma, dbeta, coeffs, coeffs + C, coeffs + 2 * C);
  if (C & 3) {
    const long n = M * C;
    hipLaunchKernelGGL(bn_bwd_apply_scalar_kernel, dim3(spnet_ew_grid(n, 256)), dim3(256), 0, st, x,
                       dy, n, C, save_mean, save_invstd, gamma, beta, act, coeffs, coeffs + C,
                       coeffs + 2 * C, dx);
  } else {
    const long n4 = M * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(spnet_ew_grid(n4, 256)), dim3(256), 0, st, x, dy,
                       M, C, save_mean, save_invstd, gamma, beta, act, coeffs, coeffs + C,
                       coeffs + 2 * C, dx, planes, ps);
  }
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_bn_bwd(const float* x, const float* dy, long M, int C, const float* gamma,
                            const float* beta, const float* save_mean, const float* save_invstd,
                            int act, float* dx, float* dgamma, float* dbeta, float* coeffs,
                            float* workspace, void* stream) {
  return bn_bwd_impl(x, dy, M, C, gamma, beta, save_mean, save_invstd, act, dx, nullptr, dgamma, dbeta, coeffs, workspace, stream);
}
// ... with dx written as the bf16x3 planes of the [M][C] matrix (csrc/x3t.h; zeroed allocation of
// 3 * spnet_bf16x3_plane_elems(M, C) bf16; C % 4 == 0): the operand of spnet_gemm_bf16x3_pp / _wgrad_batched
extern "C" int spnet_bn_bwd_x3(const float* x, const float* dy, long M, int C, const float* gamma,
                               const float* beta, const float* save_mean, const float* save_invstd,
                               int act, void* dx_planes, float* dgamma, float* dbeta, float* coeffs,
                               float* workspace, void* stream) {
  if (!dx_planes) return (int)hipErrorInvalidValue;
  return bn_bwd_impl(x, dy, M, C, gamma, beta, save_mean, save_invstd, act, nullptr, reinterpret_cast<unsigned short*>(dx_planes),
                     dgamma, dbeta, coeffs, workspace, stream);
}

// ---------------------------------------------------------------- split entry points (fused pipelines)
// The statistics may come from another kernel's epilogue (spnet_gemm_f32_colstats forward,
// spnet_dwconv3x3_tiled_bwd backward) as partial[P][2][C]; these entries do only the remaining steps.

// partial -> batch mean/invstd, scale_shift[2C], moving-stat update.  No pass over the activations.
extern "C" int spnet_bn_finalize_fwd(float* partial, int P, long M, int C, const float* gamma,
                                     const float* beta, float* moving_mean, float* moving_var,
                                     float* save_mean, float* save_invstd, float* scale_shift, float eps,
                                     float momentum, void* stream) {
  const unsigned gx = (C + BN_FIN_CH - 1) / BN_FIN_CH;
  if (P >= BN_SLICE_MIN_P) {     // `partial` is scratch of the caller: the slice sums are left in its own rows
    const int L = P / BN_SLICES, S = BN_SLICES;
    hipLaunchKernelGGL(bn_slice_partials_kernel, dim3(gx, S), dim3(256), 0, (hipStream_t)stream,
                       partial, P, C, L, S);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, partial, S, C, M, gamma,
                       beta, moving_mean, moving_var, save_mean, save_invstd, scale_shift, scale_shift + C, eps,
                       momentum, L);
    SPNET_RETURN_LAUNCH_STATUS();
  }
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(gx), dim3(256), 0,
                     (hipStream_t)stream, partial, P, C, M, gamma, beta, moving_mean, moving_var, save_mean,
                     save_invstd, scale_shift, scale_shift + C, eps, momentum, 0);
  SPNET_RETURN_LAUNCH_STATUS();
}

// scale_shift[2C] from the moving statistics (inference).
extern "C" int spnet_bn_infer_coeffs(int C, const float* gamma, const float* beta, const float* moving_mean,
                                     const float* moving_var, float* scale_shift, float eps, void* stream) {
  hipLaunchKernelGGL(bn_infer_coeffs_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, gamma,
                     beta, moving_mean, moving_var, scale_shift, scale_shift + C, eps);
  SPNET_RETURN_LAUNCH_STATUS();
}

// y = act(x*scale + shift) (+ residual)
extern "C" int spnet_bn_apply(const float* x, long M, int C, const float* scale_shift, int act,
                              const float* residual, int res_bcast, float* y, void* stream) {
  if ((C & 3) && C > 4) return (int)hipErrorInvalidValue;
  if (res_bcast && !(C & 3)) return (int)hipErrorInvalidValue;
  launch_apply(x, M, C, scale_shift, scale_shift + C, act, residual, res_bcast, y, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}

// spnet_bn_finalize_fwd + spnet_bn_apply in ONE launch where the statistics arrive as at most 128 partial rows (the
// closing BatchNorm of an Xception middle block, whose output x + BN(.) is materialised); otherwise the two launches.
extern "C" int spnet_bn_finalize_apply_ld(float* partial, int P, const float* x, long M, int C, const float* gamma,
                                          const float* beta, float* moving_mean, float* moving_var, float* save_mean,
                                          float* save_invstd, float* scale_shift, int act, const float* residual, float* y,
                                          long ldy, float eps, float momentum, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if ((C & 3) || P < 1 || (ldy & 3) || ldy < C) return (int)hipErrorInvalidValue;
  if (bn_fuse_ok(P, M, C)) {
    const int rps = bn_fuse_rows_per_slab(M, C);
    const int chunks = (C + BN_FUSE_CH - 1) / BN_FUSE_CH;
    dim3 grid((unsigned)(chunks * ((M + rps - 1) / rps)));
    hipLaunchKernelGGL(bn_fwd_fused_vec_kernel, grid, dim3(256), 0, st, x, M, C, partial, P, gamma, beta, moving_mean,
                       moving_var, save_mean, save_invstd, scale_shift, scale_shift + C, eps, momentum, act, residual, y,
                       rps, chunks, ldy);
    SPNET_RETURN_LAUNCH_STATUS();
  }
  {     // many partial rows: the two-stage finalize of spnet_bn_finalize_fwd
    const int rc = spnet_bn_finalize_fwd(partial, P, M, C, gamma, beta, moving_mean, moving_var, save_mean, save_invstd,
                                         scale_shift, eps, momentum, stream);
    if (rc) return rc;
  }
  launch_apply(x, M, C, scale_shift, scale_shift + C, act, residual, 0, y, st, ldy);
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_bn_finalize_apply(float* partial, int P, const float* x, long M, int C, const float* gamma,
                                       const float* beta, float* moving_mean, float* moving_var, float* save_mean,
                                       float* save_invstd, float* scale_shift, int act, const float* residual, float* y,
                                       float eps, float momentum, void* stream) {
  return spnet_bn_finalize_apply_ld(partial, P, x, M, C, gamma, beta, moving_mean, moving_var, save_mean, save_invstd,
                                    scale_shift, act, residual, y, C, eps, momentum, stream);
}

// Backward given the two per-channel sums as partial[P][2][C] (sum g, sum g*xhat; g already includes
// any activation mask): dgamma, dbeta, dx = k1*g + k2*xhat + k3.
static int bn_bwd_from_partials_impl(const float* x, const float* dy, long M, int C, const float* gamma,
                                     const float* beta, const float* save_mean,
                                     const float* save_invstd, int P, const float* partial, float* dx, unsigned short* planes,
                                     float* dgamma, float* dbeta, float* coeffs, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (C & 3) return (int)hipErrorInvalidValue;
  if (planes && !bn_planes_ok(planes, M, C)) return (int)hipErrorInvalidValue;
  const long ps = x3t_plane_elems(M, C);
  if (bn_fuse_ok(P, M, C)) {                                  // few partial rows: one launch (bit-identical results)
    const int rps = bn_fuse_rows_per_slab(M, C);
    const int chunks = (C + BN_FUSE_CH - 1) / BN_FUSE_CH;
    dim3 grid((unsigned)(chunks * ((M + rps - 1) / rps)));
    hipLaunchKernelGGL(bn_bwd_fused_vec_kernel, grid, dim3(256), 0, st, x, dy, M, C, partial, P, gamma, beta, save_mean,
                       save_invstd, 0, dgamma, dbeta, dx, rps, chunks, planes, ps);
    SPNET_RETURN_LAUNCH_STATUS();
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel<0>, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, st, partial, P,
                     C, M, gamma, save_invstd, save_mean, dgamma, dbeta, coeffs, coeffs + C, coeffs + 2 * C);
  const long n4 = M * (C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_vec_kernel, dim3(spnet_ew_grid(n4, 256)), dim3(256), 0, st, x, dy, M, C,
                     save_mean, save_invstd, gamma, beta, 0, coeffs, coeffs + C, coeffs + 2 * C, dx, planes, ps);
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_bn_bwd_from_partials(const float* x, const float* dy, long M, int C, const float* gamma,
                                          const float* beta, const float* save_mean,
                                          const float* save_invstd, int P, const float* partial, float* dx,
                                          float* dgamma, float* dbeta, float* coeffs, void* stream) {
  return bn_bwd_from_partials_impl(x, dy, M, C, gamma, beta, save_mean, save_invstd, P, partial, dx, nullptr, dgamma, dbeta,
                                   coeffs, stream);
}
// ... with dx as bf16x3 planes (see spnet_bn_bwd_x3)
extern "C" int spnet_bn_bwd_from_partials_x3(const float* x, const float* dy, long M, int C, const float* gamma,
                                             const float* beta, const float* save_mean,
                                             const float* save_invstd, int P, const float* partial, void* dx_planes,
                                             float* dgamma, float* dbeta, float* coeffs, void* stream) {
  if (!dx_planes) return (int)hipErrorInvalidValue;
  return bn_bwd_from_partials_impl(x, dy, M, C, gamma, beta, save_mean, save_invstd, P, partial, nullptr,
                                   reinterpret_cast<unsigned short*>(dx_planes), dgamma, dbeta, coeffs, stream);
}

// The reduction half of the backward only: dgamma, dbeta and the blend coefficients [k1 | k2' | k3'] (cld floats
// apart, the caller keeps the tail beyond C zero) with which a GEMM builds dx = k1*g + k2'*x + k3' while it stages
// its operand (spnet_gemm_f32_bnblend) -- no pass over the activations here.  Sums from partial[P][2][C] ...
extern "C" int spnet_bn_bwd_coeffs_from_partials(int P, const float* partial, long M, int C, const float* gamma,
                                                 const float* save_mean, const float* save_invstd, float* dgamma,
                                                 float* dbeta, float* coef, int cld, void* stream) {
  if (cld < C || !coef) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel<1>, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, (hipStream_t)stream,
                     partial, P, C, M, gamma, save_invstd, save_mean, dgamma, dbeta, coef, coef + cld, coef + 2 * cld);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ... or from an own reduction pass over (x, dy) (no activation behind this BatchNorm).  workspace: spnet_bn_ws(M,C).
extern "C" int spnet_bn_bwd_coeffs(const float* x, const float* dy, long M, int C, const float* gamma,
                                   const float* beta, const float* save_mean, const float* save_invstd,
                                   float* dgamma, float* dbeta, float* coef, int cld, float* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if ((C & 3) || cld < C || !coef) return (int)hipErrorInvalidValue;
  const int parts = bn_parts(M, C);
  launch_partial<1>(x, dy, M, C, save_mean, save_invstd, gamma, beta, 0, workspace, parts, st);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel<1>, dim3((C + BN_FIN_CH - 1) / BN_FIN_CH), dim3(256), 0, st, workspace, parts,
                     C, M, gamma, save_invstd, save_mean, dgamma, dbeta, coef, coef + cld, coef + 2 * cld);
  SPNET_RETURN_LAUNCH_STATUS();
}
