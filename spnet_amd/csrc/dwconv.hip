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

// ================================================================================================
// LDS-tiled depthwise kernels (the ones the engine uses).
//
// A workgroup owns TH x TW output pixels x CC4 channel quads.  Phase 1 stages the (TH+2) x (TW+2) halo
// tile into LDS with every global element fetched once per tile (16 B per lane, CC4 lanes contiguous
// along C, zero fill outside the image = SAME padding).  Phase 2: thread (channel quad, column, row
// strip) slides a 3-row window down its strip reading 3 float4 per row from LDS (a wave reads 1 KiB
// of consecutive LDS per instruction: conflict-free), so HBM sees ~1.2-1.3x the tensor on the read
// side and exactly the tensor on the write side.
//
// The backward kernel fuses the data gradient (dx = dw3x3(dz, flip w) * relu-mask + add) and the
// weight gradient (9 taps x C partial sums per workgroup, combined across the 32 threads that share a
// channel quad through LDS, then by reduce_rows over workgroups): x and dz are each read once.
// ================================================================================================
template <int CC4, int TW, int RS, int NS>
__global__ __launch_bounds__(256) void dw3x3_tile_fwd_kernel(const float* __restrict__ in,
                                                             const float* __restrict__ wt,
                                                             float* __restrict__ out, int H, int W, int C,
                                                             int relu_in, int tiles_h, int tiles_w,
                                                             int cchunks, const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift) {
  constexpr int TH = RS * NS;
  constexpr int PW = TW + 2;
  static_assert(CC4 * TW * NS == 256, "thread layout");
  extern __shared__ __attribute__((aligned(16))) float4 tile[];   // [(TH+2)][PW][CC4]
  const int c4n = C >> 2;
  int bid = blockIdx.x;
  const int cc = bid % cchunks; bid /= cchunks;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int b = bid / tiles_h;
  const int h0 = th * TH, w0 = tw * TW, c40 = cc * CC4;
  const float* base = in + (long)b * H * W * C;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < (TH + 2) * PW * CC4; idx += 256) {
    const int l = idx % CC4, p = idx / CC4;
    const int pw = p % PW, ph = p / PW;
    const int h = h0 - 1 + ph, w = w0 - 1 + pw, c4 = c40 + l;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h >= 0 && h < H && w >= 0 && w < W && c4 < c4n) {
      v = *reinterpret_cast<const float4*>(base + ((long)h * W + w) * C + c4 * 4);
      if (in_scale) {   // the producer's BatchNorm affine, applied on load (its output is never stored)
        const float4 sc = *reinterpret_cast<const float4*>(in_scale + c4 * 4);
        const float4 sh = *reinterpret_cast<const float4*>(in_shift + c4 * 4);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      }
      if (relu_in) v = f4_relu(v);
    }
    tile[idx] = v;
  }
  __syncthreads();
  const int l = tid % CC4, tcol = (tid / CC4) % TW, strip = tid / (CC4 * TW);
  const int c4 = c40 + l;
  if (c4 >= c4n) return;
  float4 k[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) k[tp] = *reinterpret_cast<const float4*>(wt + (long)tp * C + c4 * 4);
  float4 acc[RS];
#pragma unroll
  for (int r = 0; r < RS; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int rr = 0; rr < RS + 2; ++rr) {
    const float4* row = tile + ((strip * RS + rr) * PW + tcol) * CC4 + l;
    const float4 v0 = row[0], v1 = row[CC4], v2 = row[2 * CC4];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int r = rr - kh;
      if (r >= 0 && r < RS) {
        f4_fma(acc[r], v0, k[kh * 3 + 0]);
        f4_fma(acc[r], v1, k[kh * 3 + 1]);
        f4_fma(acc[r], v2, k[kh * 3 + 2]);
      }
    }
  }
  const int w = w0 + tcol;
  if (w < W) {
#pragma unroll
    for (int r = 0; r < RS; ++r) {
      const int h = h0 + strip * RS + r;
      if (h < H)
        *reinterpret_cast<float4*>(out + (((long)b * H + h) * W + w) * C + c4 * 4) = acc[r];
    }
  }
}

template <int CC4, int TW, int RS, int NS>
__global__ __launch_bounds__(256) void dw3x3_tile_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ wt,
    float* __restrict__ dx, float* __restrict__ partial, const float* __restrict__ add, int H, int W,
    int C, int relu_in, int tiles_h, int tiles_w, int cchunks, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* __restrict__ bn_mean,
    const float* __restrict__ bn_invstd, float* __restrict__ bn_partial) {
  constexpr int TH = RS * NS;
  constexpr int PW = TW + 2;
  constexpr int TILE = (TH + 2) * PW * CC4;
  constexpr int GROUP = TW * NS;            // threads that share one channel quad
  static_assert(CC4 * TW * NS == 256, "thread layout");
  static_assert(9 * GROUP * CC4 <= 2 * TILE, "reduction scratch fits in the tiles");
  extern __shared__ __attribute__((aligned(16))) float4 tile[];   // dz tile, then x tile
  float4* tdz = tile;
  float4* tx = tile + TILE;
  const int c4n = C >> 2;
  int bid = blockIdx.x;
  const int cc = bid % cchunks; bid /= cchunks;
  const int sp = bid;                        // spatial workgroup index = partial row
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int b = bid / tiles_h;
  const int h0 = th * TH, w0 = tw * TW, c40 = cc * CC4;
  const long ibase = (long)b * H * W * C;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < TILE; idx += 256) {
    const int l = idx % CC4, p = idx / CC4;
    const int pw = p % PW, ph = p / PW;
    const int h = h0 - 1 + ph, w = w0 - 1 + pw, c4 = c40 + l;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), v = g;
    if (h >= 0 && h < H && w >= 0 && w < W && c4 < c4n) {
      const long o = ibase + ((long)h * W + w) * C + c4 * 4;
      g = *reinterpret_cast<const float4*>(dz + o);
      v = *reinterpret_cast<const float4*>(x + o);
      if (in_scale) {
        const float4 sc = *reinterpret_cast<const float4*>(in_scale + c4 * 4);
        const float4 sh = *reinterpret_cast<const float4*>(in_shift + c4 * 4);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      }
    }
    tdz[idx] = g;
    tx[idx] = v;        // the forward input BEFORE its ReLU (post-affine): sign = mask, relu() = operand
  }
  __syncthreads();
  const int l = tid % CC4, tcol = (tid / CC4) % TW, strip = tid / (CC4 * TW);
  const int c4 = c40 + l;
  const bool active = c4 < c4n;
  float4 accw[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) accw[tp] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 bsg = make_float4(0.f, 0.f, 0.f, 0.f), bsgx = bsg;   // BatchNorm-backward sums of the producer
  if (active) {
    float4 kf[9];   // flipped taps for the data gradient
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) kf[tp] = *reinterpret_cast<const float4*>(wt + (long)(8 - tp) * C + c4 * 4);
    float4 accd[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) accd[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int rr = 0; rr < RS + 2; ++rr) {
      const int o = ((strip * RS + rr) * PW + tcol) * CC4 + l;
      const float4 g0 = tdz[o], g1 = tdz[o + CC4], g2 = tdz[o + 2 * CC4];
      float4 x0 = tx[o], x1 = tx[o + CC4], x2 = tx[o + 2 * CC4];
      if (relu_in) { x0 = f4_relu(x0); x1 = f4_relu(x1); x2 = f4_relu(x2); }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int r = rr - kh;               // output row of this strip fed by tile row rr through tap row kh
        if (r >= 0 && r < RS) {
          f4_fma(accd[r], g0, kf[kh * 3 + 0]);
          f4_fma(accd[r], g1, kf[kh * 3 + 1]);
          f4_fma(accd[r], g2, kf[kh * 3 + 2]);
          // weight gradient: x row rr, tap row kh, times dz at the centre of output row r
          const float4 gc = tdz[((strip * RS + r + 1) * PW + tcol + 1) * CC4 + l];
          f4_fma(accw[kh * 3 + 0], x0, gc);
          f4_fma(accw[kh * 3 + 1], x1, gc);
          f4_fma(accw[kh * 3 + 2], x2, gc);
        }
      }
    }
    const int w = w0 + tcol;
    if (w < W) {
#pragma unroll
      for (int r = 0; r < RS; ++r) {
        const int h = h0 + strip * RS + r;
        if (h < H) {
          const long o = ibase + ((long)h * W + w) * C + c4 * 4;
          float4 res = accd[r];
          if (relu_in) {
            const float4 xm = tx[((strip * RS + r + 1) * PW + tcol + 1) * CC4 + l];
            res.x = xm.x > 0.f ? res.x : 0.f;
            res.y = xm.y > 0.f ? res.y : 0.f;
            res.z = xm.z > 0.f ? res.z : 0.f;
            res.w = xm.w > 0.f ? res.w : 0.f;
          }
          if (add) {
            const float4 a = *reinterpret_cast<const float4*>(add + o);
            res.x += a.x; res.y += a.y; res.z += a.z; res.w += a.w;
          }
          *reinterpret_cast<float4*>(dx + o) = res;
          if (bn_partial) {
            // res = dL/d(BN output of the producer); xhat = (raw - mean) * invstd from the raw pre-BN value
            const float4 raw = *reinterpret_cast<const float4*>(x + o);
            const float4 mu = *reinterpret_cast<const float4*>(bn_mean + c4 * 4);
            const float4 is = *reinterpret_cast<const float4*>(bn_invstd + c4 * 4);
            bsg.x += res.x; bsg.y += res.y; bsg.z += res.z; bsg.w += res.w;
            bsgx.x = fmaf(res.x, (raw.x - mu.x) * is.x, bsgx.x);
            bsgx.y = fmaf(res.y, (raw.y - mu.y) * is.y, bsgx.y);
            bsgx.z = fmaf(res.z, (raw.z - mu.z) * is.z, bsgx.z);
            bsgx.w = fmaf(res.w, (raw.w - mu.w) * is.w, bsgx.w);
          }
        }
      }
    }
  }
  // combine the 9 tap sums over the GROUP threads of each channel quad (fixed order), one partial row
  // per spatial workgroup
  __syncthreads();
  const int grp = tid / CC4;                 // 0..GROUP-1
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) tile[(tp * GROUP + grp) * CC4 + l] = accw[tp];
  __syncthreads();
  if (tid < 9 * CC4) {
    const int tp = tid / CC4, ll = tid % CC4;
    const int c4o = c40 + ll;
    if (c4o < c4n) {
      float4 s = tile[(tp * GROUP) * CC4 + ll];
      for (int g = 1; g < GROUP; ++g) {
        const float4 v = tile[(tp * GROUP + g) * CC4 + ll];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(partial + ((long)sp * 9 + tp) * C + c4o * 4) = s;
    }
  }
  if (bn_partial) {   // second round through the same scratch: the two BatchNorm-backward sums
    __syncthreads();
    tile[(0 * GROUP + grp) * CC4 + l] = bsg;
    tile[(1 * GROUP + grp) * CC4 + l] = bsgx;
    __syncthreads();
    if (tid < 2 * CC4) {
      const int q = tid / CC4, ll = tid % CC4;
      const int c4o = c40 + ll;
      if (c4o < c4n) {
        float4 s = tile[(q * GROUP) * CC4 + ll];
        for (int g = 1; g < GROUP; ++g) {
          const float4 v = tile[(q * GROUP + g) * CC4 + ll];
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(bn_partial + ((long)sp * 2 + q) * C + c4o * 4) = s;
      }
    }
  }
}

// out[l] = sum_p in[p*L + l]   (deterministic order; used by several two-stage reductions).
// Workgroup = 16 columns x 16 interleaved row groups; grid.y > 1 splits the rows into grid.y strided
// slices written to out[y*L + l] (a second call then folds the slices).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ in, int P, int L,
                                                          float* __restrict__ out) {
  __shared__ float red[16][16];
  const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + lane;
  float s = 0.f;
  if (col < L) {
#pragma unroll 4
    for (int p = blockIdx.y + g * gridDim.y; p < P; p += 16 * gridDim.y) s += in[(long)p * L + col];
  }
  red[g][lane] = s;
  __syncthreads();
  if (g == 0 && col < L) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    out[(long)blockIdx.y * L + col] = t;
  }
}

// Fold P rows of L floats into out[L].  For many rows a first pass leaves 32 slices at the tail of the
// input buffer's own rows 0..31 region is NOT touched: the slices go to `scratch` (32*L floats).
static void launch_reduce_rows(const float* in, int P, int L, float* out, float* scratch, hipStream_t st) {
  const int gx = (L + 15) / 16;
  if (P > 128 && scratch) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(gx, 32), dim3(256), 0, st, in, P, L, scratch);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(gx, 1), dim3(256), 0, st, scratch, 32, L, out);
  } else {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(gx, 1), dim3(256), 0, st, in, P, L, out);
  }
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
  launch_reduce_rows(workspace, (int)gy, 9 * C, dw, nullptr, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_reduce_rows(const float* in, int P, int L, float* out, void* stream) {
  launch_reduce_rows(in, P, L, out, nullptr, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------- tiled entry points
struct DwGeom { int cfg, th, tw, cc4, tiles_h, tiles_w, cchunks; long nblk; size_t lds_fwd, lds_bwd; };
static DwGeom dw_geom(int B, int H, int W, int C) {
  DwGeom g;
  g.cfg = (H <= 6 && W <= 8) ? 1 : 0;       // 1: 6x8 tile x 16 quads (exit flow); 0: 12x16 tile x 8 quads
  g.th = g.cfg ? 6 : 12;
  g.tw = g.cfg ? 8 : 16;
  g.cc4 = g.cfg ? 16 : 8;
  g.tiles_h = (H + g.th - 1) / g.th;
  g.tiles_w = (W + g.tw - 1) / g.tw;
  g.cchunks = (C / 4 + g.cc4 - 1) / g.cc4;
  g.nblk = (long)B * g.tiles_h * g.tiles_w * g.cchunks;
  g.lds_fwd = (size_t)(g.th + 2) * (g.tw + 2) * g.cc4 * sizeof(float4);
  g.lds_bwd = 2 * g.lds_fwd;
  return g;
}

extern "C" int spnet_dwconv3x3_tiled_fwd(const float* x, const float* w, float* y, int B, int H, int W,
                                         int C, int relu_in, const float* in_scale,
                                         const float* in_shift, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  const DwGeom g = dw_geom(B, H, W, C);
  if (g.cfg)
    hipLaunchKernelGGL((dw3x3_tile_fwd_kernel<16, 8, 3, 2>), dim3((unsigned)g.nblk), dim3(256), g.lds_fwd,
                       (hipStream_t)stream, x, w, y, H, W, C, relu_in, g.tiles_h, g.tiles_w, g.cchunks, in_scale,
                       in_shift);
  else
    hipLaunchKernelGGL((dw3x3_tile_fwd_kernel<8, 16, 6, 2>), dim3((unsigned)g.nblk), dim3(256), g.lds_fwd,
                       (hipStream_t)stream, x, w, y, H, W, C, relu_in, g.tiles_h, g.tiles_w, g.cchunks, in_scale,
                       in_shift);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" long spnet_dwconv3x3_tiled_rows(int B, int H, int W, int C) {
  const DwGeom g = dw_geom(B, H, W, C);
  return (long)B * g.tiles_h * g.tiles_w;   // rows of the [rows][2][C] BatchNorm partial buffer
}

// floats of workspace needed by spnet_dwconv3x3_tiled_bwd
extern "C" long spnet_dwconv3x3_tiled_bwd_ws(int B, int H, int W, int C) {
  const DwGeom g = dw_geom(B, H, W, C);
  return ((long)B * g.tiles_h * g.tiles_w + 32) * 9 * C;      // partial rows + 32 second-level slices
}

// Fused backward: dx = dw3x3(dy, flip w) * (xin > 0 if relu_in) (+ add);  dw[3][3][C] = weight gradient,
// xin = x_fwd*in_scale + in_shift when the producer's BatchNorm affine is fused (else x_fwd).  With
// bn_partial != NULL the kernel also emits the producer BatchNorm's backward sums (sum dx, sum dx*xhat)
// as [rows][2][C] partials, rows = spnet_dwconv3x3_tiled_rows().
extern "C" int spnet_dwconv3x3_tiled_bwd(const float* dy, const float* x_fwd, const float* w, float* dx,
                                         float* dw, int B, int H, int W, int C, int relu_in,
                                         const float* add, float* workspace, const float* in_scale,
                                         const float* in_shift, const float* bn_mean,
                                         const float* bn_invstd, float* bn_partial, void* stream) {
  if (bn_partial && (!bn_mean || !bn_invstd)) return (int)hipErrorInvalidValue;
  if (C & 3) return (int)hipErrorInvalidValue;
  const DwGeom g = dw_geom(B, H, W, C);
  if (g.cfg)
    hipLaunchKernelGGL((dw3x3_tile_bwd_kernel<16, 8, 3, 2>), dim3((unsigned)g.nblk), dim3(256), g.lds_bwd,
                       (hipStream_t)stream, dy, x_fwd, w, dx, workspace, add, H, W, C, relu_in, g.tiles_h,
                       g.tiles_w, g.cchunks, in_scale, in_shift, bn_mean, bn_invstd, bn_partial);
  else
    hipLaunchKernelGGL((dw3x3_tile_bwd_kernel<8, 16, 6, 2>), dim3((unsigned)g.nblk), dim3(256), g.lds_bwd,
                       (hipStream_t)stream, dy, x_fwd, w, dx, workspace, add, H, W, C, relu_in, g.tiles_h,
                       g.tiles_w, g.cchunks, in_scale, in_shift, bn_mean, bn_invstd, bn_partial);
  const int P = B * g.tiles_h * g.tiles_w, L = 9 * C;
  launch_reduce_rows(workspace, P, L, dw, workspace + (long)P * L, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}
