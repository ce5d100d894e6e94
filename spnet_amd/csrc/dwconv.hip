// Depthwise 3x3 / stride 1 / SAME convolutions on NHWC fp32 tensors (C % 4 == 0): the depthwise
// step of keras SeparableConv2D inside Xception (call site spnet/models.py:357-359; 34 layers per
// forward).  Pure HBM-bound work (9 MAC per 8 bytes): lanes run along the channel axis so every
// global access is a 16-byte-per-lane coalesced segment; each thread slides a 3-row register window
// down a strip of output rows so a tile row is fetched once per strip instead of three times.
//
//   fwd        y  = dw3x3(relu?(x), w)
//   bwd_data   dx = dw3x3(dy, flip(w)) * (x > 0 if relu_in) (+ add)
//   bwd_weight dw[tap][c] = sum_{b,h,w} relu?(x)[b,h+kh-1,w+kw-1,c] * dy[b,h,w,c]   (two-stage, deterministic)
#include "common.h"

#define DW_STRIP 4

__device__ __forceinline__ float4 f4_relu(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ void f4_fma(float4& a, const float4 x, const float4 w) {
  a.x = fmaf(x.x, w.x, a.x);
  a.y = fmaf(x.y, w.y, a.y);
  a.z = fmaf(x.z, w.z, a.z);
  a.w = fmaf(x.w, w.w, a.w);
}

// MODE 0: forward (optional relu on load).  MODE 1: backward-data (taps flipped; epilogue masks with
// the saved forward input and adds `add`).
template <int MODE>
__global__ __launch_bounds__(256) void dw3x3_kernel(const float* __restrict__ in,
                                                    const float* __restrict__ wt,
                                                    float* __restrict__ out, int Bn, int H, int W,
                                                    int C, int relu_in, const float* __restrict__ xmask,
                                                    const float* __restrict__ add) {
  const int c4n = C >> 2;
  const int nstrip = (H + DW_STRIP - 1) / DW_STRIP;
  const long total = (long)Bn * nstrip * W * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int w = (int)(t % W);
    t /= W;
    const int s = (int)(t % nstrip);
    const int b = (int)(t / nstrip);
    const int h0 = s * DW_STRIP;
    const int c = c4 * 4;

    float4 k[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int src = (MODE == 1) ? (8 - tp) : tp;
      k[tp] = *reinterpret_cast<const float4*>(wt + (long)src * C + c);
    }
    float4 acc[DW_STRIP];
#pragma unroll
    for (int r = 0; r < DW_STRIP; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);

    const float* base = in + (long)b * H * W * C + c;
#pragma unroll
    for (int rr = 0; rr < DW_STRIP + 2; ++rr) {
      const int h = h0 - 1 + rr;
      if (h < 0 || h >= H) continue;
      const float* rowp = base + (long)h * W * C;
      float4 v[3];
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int ww = w - 1 + dw;
        if (ww >= 0 && ww < W) {
          float4 x = *reinterpret_cast<const float4*>(rowp + (long)ww * C);
          if (MODE == 0 && relu_in) x = f4_relu(x);
          v[dw] = x;
        } else {
          v[dw] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      // input row rr contributes to output row r = rr - kh (kh = 0..2)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int r = rr - kh;
        if (r >= 0 && r < DW_STRIP) {
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) f4_fma(acc[r], v[dw], k[kh * 3 + dw]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < DW_STRIP; ++r) {
      const int h = h0 + r;
      if (h >= H) break;
      const long o = (((long)b * H + h) * W + w) * C + c;
      float4 res = acc[r];
      if (MODE == 1) {
        if (relu_in) {
          const float4 xm = *reinterpret_cast<const float4*>(xmask + o);
          res.x = xm.x > 0.f ? res.x : 0.f;
          res.y = xm.y > 0.f ? res.y : 0.f;
          res.z = xm.z > 0.f ? res.z : 0.f;
          res.w = xm.w > 0.f ? res.w : 0.f;
        }
        if (add) {
          const float4 a = *reinterpret_cast<const float4*>(add + o);
          res.x += a.x; res.y += a.y; res.z += a.z; res.w += a.w;
        }
      }
      *reinterpret_cast<float4*>(out + o) = res;
    }
  }
}

// Stage 1 of the weight gradient.  Work unit = one image row segment (b, h, w in [w0, w0+SEG)).
// blockDim = (CL, 256/CL): x runs over channel quads, y over units; per-thread 9 float4 partial sums
// are combined across y in LDS and one partial [9][C] row is written per workgroup (grid.y rows).
#define DWW_SEG 32
__global__ __launch_bounds__(256) void dw3x3_bwd_weight_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, int Bn,
    int H, int W, int C, int relu_in) {
  extern __shared__ __attribute__((aligned(16))) float4 red[];   // [blockDim.y][9][blockDim.x]
  const int c4n = C >> 2;
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = c4 < c4n;
  const int c = c4 * 4;
  const int nseg = (W + DWW_SEG - 1) / DWW_SEG;
  const long nunits = (long)Bn * H * nseg;

  float4 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) acc[tp] = make_float4(0.f, 0.f, 0.f, 0.f);

  if (active) {
    for (long u = (long)blockIdx.y * blockDim.y + threadIdx.y; u < nunits;
         u += (long)gridDim.y * blockDim.y) {
      const int sg = (int)(u % nseg);
      long t = u / nseg;
      const int h = (int)(t % H);
      const int b = (int)(t / H);
      const int w0 = sg * DWW_SEG;
      const int w1 = min(W, w0 + DWW_SEG);
      const float* xb = x + (long)b * H * W * C + c;
      const float* dyr = dy + (((long)b * H + h) * W) * C + c;
      // sliding 3x3 window over columns: win[kh][0..2] = x[h+kh-1][w-1..w+1]
      float4 win[3][3];
      const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int hh = h + kh - 1;
        const bool hv = (hh >= 0 && hh < H);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ww = w0 - 1 + j;
          float4 v = zero;
          if (hv && ww >= 0 && ww < W) {
            v = *reinterpret_cast<const float4*>(xb + ((long)hh * W + ww) * C);
            if (relu_in) v = f4_relu(v);
          }
          win[kh][j] = v;
        }
      }
      for (int w = w0; w < w1; ++w) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int hh = h + kh - 1;
          float4 v = zero;
          if (hh >= 0 && hh < H && w + 1 < W) {
            v = *reinterpret_cast<const float4*>(xb + ((long)hh * W + w + 1) * C);
            if (relu_in) v = f4_relu(v);
          }
          win[kh][2] = v;
        }
        const float4 g = *reinterpret_cast<const float4*>(dyr + (long)w * C);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) f4_fma(acc[kh * 3 + kw], win[kh][kw], g);
          win[kh][0] = win[kh][1];
          win[kh][1] = win[kh][2];
        }
      }
    }
  }
  const int bx = blockDim.x, by = blockDim.y;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) red[(threadIdx.y * 9 + tp) * bx + threadIdx.x] = acc[tp];
  __syncthreads();
  if (threadIdx.y == 0 && active) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      float4 s = red[tp * bx + threadIdx.x];
      for (int y = 1; y < by; ++y) {
        const float4 v = red[(y * 9 + tp) * bx + threadIdx.x];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.y * 9 + tp) * C + c) = s;
    }
  }
}

// out[l] = sum_p in[p*L + l]   (deterministic order; used by several two-stage reductions)
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, int P, int L,
                                                          float* __restrict__ out) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  float s = 0.f;
  if (col < L)
    for (int p = g; p < P; p += 4) s += in[(long)p * L + col];
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && col < L) out[col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

static int chan_lanes(int c4n) {
  int cl = 8;
  while (cl < c4n && cl < 64) cl <<= 1;
  return cl;
}

extern "C" int spnet_dwconv3x3_fwd(const float* x, const float* w, float* y, int B, int H, int W,
                                   int C, int relu_in, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  const long total = (long)B * ((H + DW_STRIP - 1) / DW_STRIP) * W * (C / 4);
  hipLaunchKernelGGL(dw3x3_kernel<0>, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, w, y, B, H, W, C, relu_in, (const float*)nullptr,
                     (const float*)nullptr);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_dwconv3x3_bwd_data(const float* dy, const float* w, float* dx, int B, int H,
                                        int W, int C, int relu_in, const float* x_fwd,
                                        const float* add, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  if (relu_in && !x_fwd) return (int)hipErrorInvalidValue;
  const long total = (long)B * ((H + DW_STRIP - 1) / DW_STRIP) * W * (C / 4);
  hipLaunchKernelGGL(dw3x3_kernel<1>, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, w, dx, B, H, W, C, relu_in, x_fwd, add);
  SPNET_RETURN_LAUNCH_STATUS();
}

// workspace must hold spnet_dwconv3x3_bwd_weight_ws(B,H,W,C) floats.
extern "C" long spnet_dwconv3x3_bwd_weight_ws(int B, int H, int W, int C) {
  const int cl = chan_lanes(C / 4);
  const int by = 256 / cl;
  const long nunits = (long)B * H * ((W + DWW_SEG - 1) / DWW_SEG);
  long gy = (nunits + by - 1) / by;
  if (gy > 512) gy = 512;
  if (gy < 1) gy = 1;
  return gy * 9 * C;
}

extern "C" int spnet_dwconv3x3_bwd_weight(const float* x, const float* dy, float* dw, int B, int H,
                                          int W, int C, int relu_in, float* workspace, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  const int c4n = C / 4;
  const int cl = chan_lanes(c4n);
  const int by = 256 / cl;
  const long nunits = (long)B * H * ((W + DWW_SEG - 1) / DWW_SEG);
  long gy = (nunits + by - 1) / by;
  if (gy > 512) gy = 512;
  if (gy < 1) gy = 1;
  dim3 grid((c4n + cl - 1) / cl, (unsigned)gy), block(cl, by);
  const size_t shm = (size_t)by * 9 * cl * sizeof(float4);
  hipLaunchKernelGGL(dw3x3_bwd_weight_partial_kernel, grid, block, shm, (hipStream_t)stream, x, dy,
                     workspace, B, H, W, C, relu_in);
  const int L = 9 * C;
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((L + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     workspace, (int)gy, L, dw);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_reduce_rows(const float* in, int P, int L, float* out, void* stream) {
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((L + 63) / 64), dim3(256), 0, (hipStream_t)stream, in,
                     P, L, out);
  SPNET_RETURN_LAUNCH_STATUS();
}
