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
  // XCD-aware order: an XCD's workgroups cover a contiguous run of pixels, so the image rows that vertically
  // neighbouring windows share stay in one L2
  for (long p = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; p < total;
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

// The two 3 -> 3 stem convolutions ('same', stride 1), forward and data gradient, round 5: a thread owns a RUN of four
// pixels of one image row.  Per tap row it loads the run as three aligned 16-byte words plus the pixel on either side
// (two 12-byte loads) -- 15 loads per four pixels where one thread per pixel issued 108 -- and the 81 weights sit in scalar
// registers (uniform loads from constant indices) instead of LDS.  Every output's fmaf chain keeps the order of
// conv3x3_small_{fwd,bwd_data}_kernel (tap rows, tap columns, then channels), so the results are theirs bit for bit;
// the taps that fall outside the image multiply a zero input instead of being skipped.  W % 4 == 0.
//   BWD = 0:  y[p][co]  = sum_{kh,kw,ci} x[h + kh - 1][w + kw - 1][ci] * k[kh][kw][ci][co]
//   BWD = 1:  dx[p][ci] = sum_{kh,kw,co} dy[h + 1 - kh][w + 1 - kw][co] * k[kh][kw][ci][co]
struct c3_px { float c[3]; };
template <int BWD>
__global__ __launch_bounds__(256) void conv3x3_c3_run4_kernel(const float* __restrict__ in, const float* __restrict__ k,
                                                              float* __restrict__ out, int Bn, int H, int W) {
  const int runs_w = W >> 2;
  const long total = (long)Bn * H * runs_w;
  for (long q = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; q < total;
       q += (long)gridDim.x * blockDim.x) {
    const int rw = (int)(q % runs_w);
    const long t = q / runs_w;
    const int h = (int)(t % H);
    const long img = t / H;
    const int w0 = rw * 4;
    float acc[4][3];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[p][c] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int rr = BWD ? h + 1 - kh : h + kh - 1;
      if (rr < 0 || rr >= H) continue;
      const float* rp = in + ((img * H + rr) * W + w0) * 3;
      float v[18];                                 // pixels w0 - 1 .. w0 + 4, three channels each
      const float4 a0 = *reinterpret_cast<const float4*>(rp), a1 = *reinterpret_cast<const float4*>(rp + 4),
                   a2 = *reinterpret_cast<const float4*>(rp + 8);
      v[3] = a0.x; v[4] = a0.y; v[5] = a0.z; v[6] = a0.w; v[7] = a1.x; v[8] = a1.y; v[9] = a1.z; v[10] = a1.w;
      v[11] = a2.x; v[12] = a2.y; v[13] = a2.z; v[14] = a2.w;
      c3_px l = {{0.f, 0.f, 0.f}}, r = {{0.f, 0.f, 0.f}};
      if (w0 > 0) l = *reinterpret_cast<const c3_px*>(rp - 3);
      if (w0 + 4 < W) r = *reinterpret_cast<const c3_px*>(rp + 12);
      v[0] = l.c[0]; v[1] = l.c[1]; v[2] = l.c[2]; v[15] = r.c[0]; v[16] = r.c[1]; v[17] = r.c[2];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int px = BWD ? p + 2 - kw : p + kw;          // index of the input pixel in v (pixel w0 - 1 is 0)
          if (!BWD) {
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
              for (int co = 0; co < 3; ++co)
                acc[p][co] = fmaf(v[px * 3 + ci], k[((kh * 3 + kw) * 3 + ci) * 3 + co], acc[p][co]);
          } else {
#pragma unroll
            for (int co = 0; co < 3; ++co)
#pragma unroll
              for (int ci = 0; ci < 3; ++ci)
                acc[p][ci] = fmaf(v[px * 3 + co], k[((kh * 3 + kw) * 3 + ci) * 3 + co], acc[p][ci]);
          }
        }
      }
    }
    float* op = out + ((img * H + h) * W + w0) * 3;
    *reinterpret_cast<float4*>(op) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[1][0]);
    *reinterpret_cast<float4*>(op + 4) = make_float4(acc[1][1], acc[1][2], acc[2][0], acc[2][1]);
    *reinterpret_cast<float4*>(op + 8) = make_float4(acc[2][2], acc[3][0], acc[3][1], acc[3][2]);
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
  // XCD-aware order: an XCD's workgroups cover a contiguous run of pixels, so the image rows that vertically
  // neighbouring windows share stay in one L2
  for (long p = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; p < total;
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
  // XCD-aware order: an XCD's workgroups cover a contiguous run of pixels, so the image rows that vertically
  // neighbouring windows share stay in one L2
  for (long p = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; p < total;
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

// Weight gradient of the 3 -> 3 stem convs (stride 1, SAME) on rows whose width is a multiple of 8: a thread owns RUNS of
// eight consecutive output pixels.  Per run it fetches dy (24 floats) and, tap row by tap row, the ten input pixels the
// eight windows of that row touch (six 16-byte loads + the two edge pixels), instead of 27 + 3 scalar loads per pixel;
// the 81 sums stay in registers across the thread's runs.  Same two-stage deterministic reduction as the generic kernel.
__global__ __launch_bounds__(256) void conv3x3_c3_bwd_weight_rows_kernel(const float* __restrict__ x,
                                                                         const float* __restrict__ dy,
                                                                         float* __restrict__ partial, int Bn, int H,
                                                                         int W) {
  constexpr int NW = 81;
  __shared__ float red[4][NW];
  float acc[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) acc[i] = 0.f;
  const int runs_w = W / 8;
  const long total = (long)Bn * H * runs_w;
  for (long run = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; run < total;
       run += (long)gridDim.x * blockDim.x) {
    const int rw = (int)(run % runs_w);
    long t = run / runs_w;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    const int w0 = rw * 8;
    float g[24];                                   // dy of the eight pixels, [pixel][co]
    {
      const float4* gp = reinterpret_cast<const float4*>(dy + (((long)b * H + h) * W + w0) * 3);
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const float4 v = gp[q];
        g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
      }
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h - 1 + kh;
      float xr[30];                                // input pixels w0-1 .. w0+8 of row hh, [pixel][ci]; zero outside
      if (hh >= 0 && hh < H) {
        const float* row = x + (((long)b * H + hh) * W) * 3;
        const float4* xp = reinterpret_cast<const float4*>(row + (long)w0 * 3);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const float4 v = xp[q];
          xr[3 + 4 * q] = v.x; xr[3 + 4 * q + 1] = v.y; xr[3 + 4 * q + 2] = v.z; xr[3 + 4 * q + 3] = v.w;
        }
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
          xr[ci] = (w0 > 0) ? row[(long)(w0 - 1) * 3 + ci] : 0.f;
          xr[27 + ci] = (w0 + 8 < W) ? row[(long)(w0 + 8) * 3 + ci] : 0.f;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 30; ++i) xr[i] = 0.f;
      }
#pragma unroll
      for (int px = 0; px < 8; ++px)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
          for (int ci = 0; ci < 3; ++ci) {
            const float xv = xr[(px + kw) * 3 + ci];
#pragma unroll
            for (int co = 0; co < 3; ++co)
              acc[((kh * 3 + kw) * 3 + ci) * 3 + co] = fmaf(xv, g[px * 3 + co], acc[((kh * 3 + kw) * 3 + ci) * 3 + co]);
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

// ------------------------------------------------------------------------------------------------
// block1_conv1 (3 -> 32, stride 2, VALID): the one small conv whose OUTPUT is wide (32 channels at
// half resolution = 2.6x the bytes of its input), so each of its three passes is organised around
// coalesced 128-byte pixel records of y / dy:
//   forward   LDS-staged input patch per 4x32-pixel output tile; 8 threads per pixel, each owning one
//             quad of output channels with its 27 weight quads in registers -> float4 stores, 1 KB per wave
//   dX        one thread per 2x2 block of input pixels: the 4 dy pixels around it feed its 12 outputs
//             through all 9 taps with no stride/parity divergence; weights are wave-uniform (scalar loads)
//   dW        same tile + dy tile in LDS, thread = (tap*ci, channel quad), one partial filter per workgroup
// ------------------------------------------------------------------------------------------------
#define C1_TH 4
#define C1_TW 32
#define C1_PR (2 * C1_TH + 1)          // patch rows
#define C1_PC (2 * C1_TW + 1)          // patch pixels per row
#define C1_PLD (3 * C1_PC + 1)         // floats per patch row (padded)

__device__ __forceinline__ void c1_stage_patch(float* __restrict__ patch, const float* __restrict__ x, int b,
                                               int H, int W, int oh0, int ow0, int tid) {
  for (int e = tid; e < C1_PR * 3 * C1_PC; e += 256) {
    const int r = e / (3 * C1_PC), c = e % (3 * C1_PC);
    const int h = 2 * oh0 + r, wc = 2 * ow0 * 3 + c;          // wc: float index inside the image row
    patch[r * C1_PLD + c] = (h < H && wc < 3 * W) ? x[((long)b * H + h) * W * 3 + wc] : 0.f;
  }
}

__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w, float* __restrict__ y,
                                                        int H, int W, int OH, int OW) {
  __shared__ float patch[C1_PR * C1_PLD];
  const int tid = threadIdx.x;
  const int b = blockIdx.z, oh0 = blockIdx.y * C1_TH, ow0 = blockIdx.x * C1_TW;
  const int cq = tid & 7, pl = tid >> 3;
  float4 wr[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wr[k] = *reinterpret_cast<const float4*>(w + k * 32 + cq * 4);
  c1_stage_patch(patch, x, b, H, W, oh0, ow0, tid);
  __syncthreads();
  const int ow = ow0 + pl;
#pragma unroll
  for (int r = 0; r < C1_TH; ++r) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const float* pp = patch + (2 * r + kh) * C1_PLD + 6 * pl;    // 9 consecutive floats: [kw][ci]
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const float xv = pp[q];
        const float4 wv = wr[kh * 9 + q];
        acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y);
        acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w);
      }
    }
    const int oh = oh0 + r;
    if (oh < OH && ow < OW)
      *reinterpret_cast<float4*>(y + (((long)b * OH + oh) * OW + ow) * 32 + cq * 4) = acc;
  }
}

// One thread per 2x2 block (i, j) of input pixels; dy pixel (i - di, j - dj), di, dj in {0, 1}, reaches
// input pixel (2i + ph, 2j + pw) through tap (ph + 2 di, pw + 2 dj) when that tap index is <= 2.
__global__ __launch_bounds__(256) void conv1_bwd_data_kernel(const float* __restrict__ dy,
                                                             const float* __restrict__ w,
                                                             float* __restrict__ dx, int Bn, int H, int W,
                                                             int OH, int OW) {
  const int H2 = (H + 1) >> 1, W2 = (W + 1) >> 1;
  const long total = (long)Bn * H2 * W2;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int j = (int)(t % W2);
    long u = t / W2;
    const int i = (int)(u % H2);
    const int b = (int)(u / H2);
    float acc[2][2][3];
#pragma unroll
    for (int a = 0; a < 12; ++a) (&acc[0][0][0])[a] = 0.f;
#pragma unroll
    for (int di = 0; di < 2; ++di) {
#pragma unroll
      for (int dj = 0; dj < 2; ++dj) {
        const int oh = i - di, ow = j - dj;
        if (oh < 0 || oh >= OH || ow < 0 || ow >= OW) continue;
        const float4* gp = reinterpret_cast<const float4*>(dy + (((long)b * OH + oh) * OW + ow) * 32);
        float g[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 v = gp[q];
          g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          if (ph + 2 * di > 2) continue;
#pragma unroll
          for (int pw = 0; pw < 2; ++pw) {
            if (pw + 2 * dj > 2) continue;
            const float* wt = w + (((ph + 2 * di) * 3 + (pw + 2 * dj)) * 3) * 32;   // wave-uniform
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
              float s = acc[ph][pw][ci];
#pragma unroll
              for (int co = 0; co < 32; ++co) s = fmaf(g[co], wt[ci * 32 + co], s);
              acc[ph][pw][ci] = s;
            }
          }
        }
      }
    }
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int h = 2 * i + ph;
      if (h >= H) continue;
      float* dp = dx + (((long)b * H + h) * W + 2 * j) * 3;
      if (2 * j + 1 < W) {       // 6 consecutive floats, 8-byte aligned
        reinterpret_cast<float2*>(dp)[0] = make_float2(acc[ph][0][0], acc[ph][0][1]);
        reinterpret_cast<float2*>(dp)[1] = make_float2(acc[ph][0][2], acc[ph][1][0]);
        reinterpret_cast<float2*>(dp)[2] = make_float2(acc[ph][1][1], acc[ph][1][2]);
      } else {
        dp[0] = acc[ph][0][0]; dp[1] = acc[ph][0][1]; dp[2] = acc[ph][0][2];
      }
    }
  }
}

// Workgroups walk over output tiles (grid-stride); thread (tg = tid >> 3, cq = tid & 7) owns
// dW[tap*3+ci = tg][4 cq .. 4 cq + 3] and adds x_patch(pixel, tg) * dy(pixel, quad) over the tile's pixels.
__global__ __launch_bounds__(256) void conv1_bwd_weight_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ dy,
                                                               float* __restrict__ partial, int Bn, int H,
                                                               int W, int OH, int OW, int tiles_w,
                                                               int tiles_h) {
  __shared__ float patch[C1_PR * C1_PLD];
  __shared__ __attribute__((aligned(16))) float gt[C1_TH * C1_TW * 32];
  const int tid = threadIdx.x;
  const int cq = tid & 7, tg = tid >> 3;
  const int tc = tg < 27 ? tg : 26;
  const int toff = (tc / 9) * C1_PLD + (tc % 9);                 // (kh, kw*3+ci) inside the patch
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int ntiles = Bn * tiles_h * tiles_w;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int twi = tile % tiles_w;
    const int thi = (tile / tiles_w) % tiles_h;
    const int b = tile / (tiles_w * tiles_h);
    const int oh0 = thi * C1_TH, ow0 = twi * C1_TW;
    __syncthreads();                                             // previous tile fully consumed
    c1_stage_patch(patch, x, b, H, W, oh0, ow0, tid);
    for (int e = tid; e < C1_TH * C1_TW * 8; e += 256) {         // dy tile, float4 granules, zero outside
      const int px = e >> 3, q = e & 7;
      const int oh = oh0 + px / C1_TW, ow = ow0 + px % C1_TW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oh < OH && ow < OW) v = *reinterpret_cast<const float4*>(dy + (((long)b * OH + oh) * OW + ow) * 32 + q * 4);
      *reinterpret_cast<float4*>(gt + px * 32 + q * 4) = v;
    }
    __syncthreads();
#pragma unroll 8
    for (int px = 0; px < C1_TH * C1_TW; ++px) {
      const float xv = patch[2 * (px / C1_TW) * C1_PLD + 6 * (px % C1_TW) + toff];
      const float4 g = *reinterpret_cast<const float4*>(gt + px * 32 + cq * 4);
      acc.x = fmaf(xv, g.x, acc.x); acc.y = fmaf(xv, g.y, acc.y);
      acc.z = fmaf(xv, g.z, acc.z); acc.w = fmaf(xv, g.w, acc.w);
    }
  }
  if (tg < 27) *reinterpret_cast<float4*>(partial + (long)blockIdx.x * 864 + tg * 32 + cq * 4) = acc;
}

extern "C" int spnet_reduce_rows(const float* in, int P, int L, float* out, void* stream);   // dwconv.hip

template <int CIN, int COUT, int STRIDE, int PAD>
static int conv_small_dispatch(int op, const float* a, const float* b, float* out, int B, int H, int W,
                               float* workspace, long ws_floats, hipStream_t st) {
  const int OH = (PAD == 1) ? H : (H - 3) / STRIDE + 1;
  const int OW = (PAD == 1) ? W : (W - 3) / STRIDE + 1;
  constexpr int NW = 9 * CIN * COUT;
  if constexpr (CIN == 3 && COUT == 32 && STRIDE == 2 && PAD == 0) {
    const int tw = spnet_cdiv(OW, C1_TW), th = spnet_cdiv(OH, C1_TH);
    if (op == 0) {
      hipLaunchKernelGGL(conv1_fwd_kernel, dim3(tw, th, B), dim3(256), 0, st, a, b, out, H, W, OH, OW);
    } else if (op == 1) {
      const long total = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
      hipLaunchKernelGGL(conv1_bwd_data_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, st, a, b, out, B,
                         H, W, OH, OW);
    } else {
      long parts = (long)B * tw * th;
      if (parts > 768) parts = 768;
      if (parts * NW > ws_floats) return (int)hipErrorInvalidValue;
      hipLaunchKernelGGL(conv1_bwd_weight_kernel, dim3((unsigned)parts), dim3(256), 0, st, a, b, workspace, B, H,
                         W, OH, OW, tw, th);
      return spnet_reduce_rows(workspace, (int)parts, NW, out, (void*)st);
    }
    return (int)hipGetLastError();
  }
  if constexpr (CIN == 3 && COUT == 3 && STRIDE == 1 && PAD == 1) {
    if (op < 2 && (W & 3) == 0 && !((((uintptr_t)a) | ((uintptr_t)out)) & 15)) {     // runs of four pixels (bit-identical)
      const long total = (long)B * H * (W / 4);
      if (op == 0)
        hipLaunchKernelGGL(conv3x3_c3_run4_kernel<0>, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, st, a, b, out, B, H, W);
      else
        hipLaunchKernelGGL(conv3x3_c3_run4_kernel<1>, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, st, a, b, out, B, H, W);
      return (int)hipGetLastError();
    }
  }
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
    if (CIN == 3 && COUT == 3 && STRIDE == 1 && PAD == 1 && (W & 7) == 0) {
      // runs of eight pixels per thread, about three runs per thread (the 81 wave reductions of the epilogue are a
      // fixed cost per workgroup)
      parts = spnet_cdiv(total / 8, 256 * 3);
      if (parts > 512) parts = 512;
      if (parts < 1) parts = 1;
      if ((long)parts * NW > ws_floats) return (int)hipErrorInvalidValue;
      hipLaunchKernelGGL(conv3x3_c3_bwd_weight_rows_kernel, dim3(parts), dim3(256), 0, st, a, b, workspace, B, H, W);
    } else if constexpr (NW <= 81) {
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

// ================================================================================================
// Stem head, fused: conv2d_1 (1 -> 3, 3x3 SAME) + AveragePooling2D(2) and the skip connection's
// AveragePooling2D(2) of the input (spnet/models.py:321-323, 337).  The full-resolution 3-channel tensor
// (75 MB per 32 frames of 384x512, the largest activation of the network) is never written or read:
//   forward   thread = one pooled pixel: its 4x4 input window (zero padded) gives the four conv outputs of the
//             2x2 pooling cell; same accumulation order as conv3x3_small_fwd_kernel + avgpool2_fwd_kernel, so the
//             result is bit-identical to the unfused pair
//   backward  dW[kh][kw][c] = sum over pooled pixels of 0.25*dp[c] * (the four window entries the tap sees);
//             the pooling backward (a 4x upsampled, 75 MB gradient) is folded into the weight-gradient sum.
//             The layer has no data gradient (it reads the input frame).
// ================================================================================================
__global__ __launch_bounds__(256) void stem_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ p1, float* __restrict__ px, int Bn, int H,
                                                            int W, int OH, int OW) {
  __shared__ float ws[27];
  if (threadIdx.x < 27) ws[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const long total = (long)Bn * OH * OW;
  for (long p = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; p < total;
       p += (long)gridDim.x * blockDim.x) {
    const int ow = (int)(p % OW);
    long t = p / OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float win[4][4];                             // rows 2oh-1 .. 2oh+2, columns 2ow-1 .. 2ow+2
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = 2 * oh - 1 + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ww = 2 * ow - 1 + c;
        win[r][c] = (h >= 0 && h < H && ww >= 0 && ww < W) ? x[((long)b * H + h) * W + ww] : 0.f;
      }
    }
    float cv[2][2][3];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            // taps outside the image contribute nothing (the unfused kernel skips them; adding 0*w is the same value)
            const float xv = win[dy + kh][dx + kw];
#pragma unroll
            for (int co = 0; co < 3; ++co) acc[co] = fmaf(xv, ws[(kh * 3 + kw) * 3 + co], acc[co]);
          }
#pragma unroll
        for (int co = 0; co < 3; ++co) cv[dy][dx][co] = acc[co];
      }
#pragma unroll
    for (int co = 0; co < 3; ++co)
      p1[p * 3 + co] = ((cv[0][0][co] + cv[0][1][co]) + (cv[1][0][co] + cv[1][1][co])) * 0.25f;
    px[p] = ((win[1][1] + win[1][2]) + (win[2][1] + win[2][2])) * 0.25f;
  }
}

__global__ __launch_bounds__(256) void stem_head_bwd_weight_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ dp,
                                                                   float* __restrict__ partial, int Bn, int H, int W,
                                                                   int OH, int OW) {
  __shared__ float red[4][27];
  float acc[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) acc[i] = 0.f;
  const long total = (long)Bn * OH * OW;
  for (long p = (long)xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; p < total;
       p += (long)gridDim.x * blockDim.x) {
    const int ow = (int)(p % OW);
    long t = p / OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float win[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int h = 2 * oh - 1 + r;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ww = 2 * ow - 1 + c;
        win[r][c] = (h >= 0 && h < H && ww >= 0 && ww < W) ? x[((long)b * H + h) * W + ww] : 0.f;
      }
    }
    float g[3];
#pragma unroll
    for (int co = 0; co < 3; ++co) g[co] = 0.25f * dp[p * 3 + co];      // AveragePooling2D backward
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float xs = (win[kh][kw] + win[kh][kw + 1]) + (win[kh + 1][kw] + win[kh + 1][kw + 1]);
#pragma unroll
        for (int co = 0; co < 3; ++co) acc[(kh * 3 + kw) * 3 + co] = fmaf(xs, g[co], acc[(kh * 3 + kw) * 3 + co]);
      }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 27; ++i) {
    const float s = wave_sum(acc[i]);
    if (lane == 0) red[wv][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 27)
    partial[(long)blockIdx.x * 27 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// op 0: p1 [B][H/2][W/2][3] = avgpool2(conv3x3_same(x, w)), px [B][H/2][W/2] = avgpool2(x)   (a = w, out = p1, out2 = px)
// op 2: dw [3][3][1][3] from the gradient dp1 of p1                                          (a = dp1, out = dw)
extern "C" int spnet_stem_head(int op, const float* x, const float* a, float* out, float* out2, int B, int H, int W,
                               float* workspace, long ws_floats, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int OH = H / 2, OW = W / 2;
  if (OH < 1 || OW < 1 || !x || !a || !out) return (int)hipErrorInvalidValue;
  const long total = (long)B * OH * OW;
  if (op == 0) {
    if (!out2) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(stem_head_fwd_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, st, x, a, out, out2, B, H, W,
                       OH, OW);
  } else if (op == 2) {
    int parts = spnet_cdiv(total, 256 * 8);
    if (parts > 512) parts = 512;
    if (parts < 1) parts = 1;
    if (!workspace || (long)parts * 27 > ws_floats) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(stem_head_bwd_weight_kernel, dim3(parts), dim3(256), 0, st, x, a, workspace, B, H, W, OH, OW);
    return spnet_reduce_rows(workspace, parts, 27, out, (void*)st);
  } else {
    return (int)hipErrorInvalidValue;
  }
  SPNET_RETURN_LAUNCH_STATUS();
}
