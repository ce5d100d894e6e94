// Direct 3x3 convolutions with a handful of channels, NHWC fp32, weights HWIO [3][3][CIN][COUT]:
//   stem   conv2d_1 (1->3), conv2d_2/3 (3->3): stride 1, SAME            spnet/models.py:321,330,335
//   block1_conv1 (3->32): stride 2, VALID                                 keras Xception (models.py:359)
// These layers carry ~0.3 % of the network's MACs but touch the largest tensors (full-resolution
// frames), so they are plain bandwidth-bound loops: one thread per pixel, all output channels in
// registers, weights broadcast from LDS.
#include "common.h"

template <int CIN, int COUT, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void conv3x3_small_fwd_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ w,
                                                                float* __restrict__ y, int Bn, int H,
                                                                int W, int OH, int OW) {
  __shared__ float ws[9 * CIN * COUT];
  for (int i = threadIdx.x; i < 9 * CIN * COUT; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const long total = (long)Bn * OH * OW;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (long)gridDim.x * blockDim.x) {
    const int ow = (int)(p % OW);
    long t = p / OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = oh * STRIDE - PAD + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = ow * STRIDE - PAD + kw;
        if (ww < 0 || ww >= W) continue;
        const float* xp = x + (((long)b * H + h) * W + ww) * CIN;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float xv = xp[ci];
          const float* wp = ws + ((kh * 3 + kw) * CIN + ci) * COUT;
#pragma unroll
          for (int co = 0; co < COUT; ++co) acc[co] = fmaf(xv, wp[co], acc[co]);
        }
      }
    }
    float* yp = y + p * COUT;
#pragma unroll
    for (int co = 0; co < COUT; ++co) yp[co] = acc[co];
  }
}

// dx[b,h,w,ci] = sum_{kh,kw,co} dy[b,oh,ow,co] * w[kh,kw,ci,co],  oh*S - PAD + kh = h
template <int CIN, int COUT, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void conv3x3_small_bwd_data_kernel(const float* __restrict__ dy,
                                                                     const float* __restrict__ w,
                                                                     float* __restrict__ dx, int Bn,
                                                                     int H, int W, int OH, int OW) {
  __shared__ float ws[9 * CIN * COUT];
  for (int i = threadIdx.x; i < 9 * CIN * COUT; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const long total = (long)Bn * H * W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (long)gridDim.x * blockDim.x) {
    const int ww = (int)(p % W);
    long t = p / W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    float acc[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int nh = h + PAD - kh;
      if (nh < 0 || (nh % STRIDE) != 0) continue;
      const int oh = nh / STRIDE;
      if (oh >= OH) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int nw = ww + PAD - kw;
        if (nw < 0 || (nw % STRIDE) != 0) continue;
        const int ow = nw / STRIDE;
        if (ow >= OW) continue;
        const float* gp = dy + (((long)b * OH + oh) * OW + ow) * COUT;
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float g = gp[co];
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci)
            acc[ci] = fmaf(g, ws[((kh * 3 + kw) * CIN + ci) * COUT + co], acc[ci]);
        }
      }
    }
    float* dp = dx + p * CIN;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) dp[ci] = acc[ci];
  }
}

// Weight gradient, small filters (9*CIN*COUT <= 81): every thread accumulates the whole filter over
// its pixels in registers; wave shuffle + LDS combine; one partial filter per workgroup.
template <int CIN, int COUT, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void conv3x3_small_bwd_weight_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ dy,
                                                                       float* __restrict__ partial,
                                                                       int Bn, int H, int W, int OH,
                                                                       int OW) {
  constexpr int NW = 9 * CIN * COUT;
  __shared__ float red[4][NW];
  float acc[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) acc[i] = 0.f;
  const long total = (long)Bn * OH * OW;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < total;
       p += (long)gridDim.x * blockDim.x) {
    const int ow = (int)(p % OW);
    long t = p / OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float g[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) g[co] = dy[p * COUT + co];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = oh * STRIDE - PAD + kh;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = ow * STRIDE - PAD + kw;
        const bool ok = (h >= 0 && h < H && ww >= 0 && ww < W);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          const float xv = ok ? x[(((long)b * H + h) * W + ww) * CIN + ci] : 0.f;
#pragma unroll
          for (int co = 0; co < COUT; ++co)
            acc[((kh * 3 + kw) * CIN + ci) * COUT + co] = fmaf(xv, g[co], acc[((kh * 3 + kw) * CIN + ci) * COUT + co]);
        }
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NW; i += blockDim.x)
    partial[(long)blockIdx.x * NW + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// Weight gradient for block1_conv1 (3 -> 32, 864 weights): lanes run over the 32 output channels,
// 8 lane-groups split the 27 (tap, ci) pairs; each group loops over the workgroup's pixels.
template <int CIN, int COUT, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void conv3x3_wide_bwd_weight_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ dy,
                                                                      float* __restrict__ partial,
                                                                      int Bn, int H, int W, int OH,
                                                                      int OW) {
  static_assert(COUT == 32, "lane layout assumes 32 output channels");
  constexpr int NTC = 9 * CIN;                 // (tap, ci) pairs
  constexpr int GROUPS = 8;
  constexpr int PER = (NTC + GROUPS - 1) / GROUPS;
  const int co = threadIdx.x & 31;
  const int grp = threadIdx.x >> 5;
  float acc[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) acc[j] = 0.f;
  const long total = (long)Bn * OH * OW;
  // PB pixels per trip with all their loads issued before the FMAs: the loop is latency-bound
  // (a handful of dependent-free scalar loads per pixel), so memory-level parallelism is what counts.
  constexpr int PB = 8;
  for (long p0 = (long)blockIdx.x * PB; p0 < total; p0 += (long)gridDim.x * PB) {
    float g[PB], xv[PB][PER];
#pragma unroll
    for (int u = 0; u < PB; ++u) {
      const long p = p0 + u;
      const bool pv = p < total;
      const int ow = pv ? (int)(p % OW) : 0;
      const long t = pv ? p / OW : 0;
      const int oh = (int)(t % OH);
      const int b = (int)(t / OH);
      g[u] = pv ? dy[p * COUT + co] : 0.f;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        const int tc = grp * PER + j;
        float v = 0.f;
        if (pv && tc < NTC) {
          const int tap = tc / CIN, ci = tc % CIN;
          const int h = oh * STRIDE - PAD + tap / 3, ww = ow * STRIDE - PAD + tap % 3;
          if (h >= 0 && h < H && ww >= 0 && ww < W) v = x[(((long)b * H + h) * W + ww) * CIN + ci];
        }
        xv[u][j] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < PB; ++u)
#pragma unroll
      for (int j = 0; j < PER; ++j) acc[j] = fmaf(xv[u][j], g[u], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int tc = grp * PER + j;
    if (tc < NTC) partial[(long)blockIdx.x * (NTC * COUT) + tc * COUT + co] = acc[j];
  }
}

extern "C" int spnet_reduce_rows(const float* in, int P, int L, float* out, void* stream);   // dwconv.hip

template <int CIN, int COUT, int STRIDE, int PAD>
static int conv_small_dispatch(int op, const float* a, const float* b, float* out, int B, int H, int W,
                               float* workspace, long ws_floats, hipStream_t st) {
  const int OH = (PAD == 1) ? H : (H - 3) / STRIDE + 1;
  const int OW = (PAD == 1) ? W : (W - 3) / STRIDE + 1;
  constexpr int NW = 9 * CIN * COUT;
  if (op == 0) {  // forward: a = x, b = w
    const long total = (long)B * OH * OW;
    hipLaunchKernelGGL((conv3x3_small_fwd_kernel<CIN, COUT, STRIDE, PAD>), dim3(spnet_ew_grid(total, 256)),
                       dim3(256), 0, st, a, b, out, B, H, W, OH, OW);
  } else if (op == 1) {  // backward data: a = dy, b = w
    const long total = (long)B * H * W;
    hipLaunchKernelGGL((conv3x3_small_bwd_data_kernel<CIN, COUT, STRIDE, PAD>),
                       dim3(spnet_ew_grid(total, 256)), dim3(256), 0, st, a, b, out, B, H, W, OH, OW);
  } else {  // backward weight: a = x, b = dy
    const long total = (long)B * OH * OW;
    int parts;
    if constexpr (NW <= 81) {
      parts = spnet_cdiv(total, 256 * 16);
      if (parts > 512) parts = 512;
      if (parts < 1) parts = 1;
      if ((long)parts * NW > ws_floats) return (int)hipErrorInvalidValue;
      hipLaunchKernelGGL((conv3x3_small_bwd_weight_kernel<CIN, COUT, STRIDE, PAD>), dim3(parts),
                         dim3(256), 0, st, a, b, workspace, B, H, W, OH, OW);
    } else {
      parts = (int)(total < 1024 ? total : 1024);
      if ((long)parts * NW > ws_floats) return (int)hipErrorInvalidValue;
      hipLaunchKernelGGL((conv3x3_wide_bwd_weight_kernel<CIN, COUT, STRIDE, PAD>), dim3(parts),
                         dim3(256), 0, st, a, b, workspace, B, H, W, OH, OW);
    }
    return spnet_reduce_rows(workspace, parts, NW, out, (void*)st);
  }
  return (int)hipGetLastError();
}

// op: 0 forward (a=x, b=w, out=y) | 1 backward-data (a=dy, b=w, out=dx) | 2 backward-weight
// (a=x, b=dy, out=dw; needs workspace of >= 1024*9*cin*cout floats).
// Supported (cin, cout, stride, same): (1,3,1,1) (3,3,1,1) (3,32,2,0).
extern "C" int spnet_conv3x3_small(int op, int cin, int cout, int stride, int same, const float* a,
                                   const float* b, float* out, int B, int H, int W, float* workspace,
                                   long ws_floats, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (cin == 1 && cout == 3 && stride == 1 && same == 1)
    return conv_small_dispatch<1, 3, 1, 1>(op, a, b, out, B, H, W, workspace, ws_floats, st);
  if (cin == 3 && cout == 3 && stride == 1 && same == 1)
    return conv_small_dispatch<3, 3, 1, 1>(op, a, b, out, B, H, W, workspace, ws_floats, st);
  if (cin == 3 && cout == 32 && stride == 2 && same == 0)
    return conv_small_dispatch<3, 32, 2, 0>(op, a, b, out, B, H, W, workspace, ws_floats, st);
  return (int)hipErrorInvalidValue;
}
