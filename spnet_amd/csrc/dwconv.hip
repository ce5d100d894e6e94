// Depthwise 3x3 / SAME convolutions on NHWC fp32 tensors (C % 4 == 0): the depthwise step of keras
// SeparableConv2D inside Xception (call site spnet/models.py:357-359; 34 stride-1 layers per forward) and MobileNet's
// DepthwiseConv2D (stride 1 | 2, spnet/models.py:346-355).  Pure HBM-bound work (9 MAC per 8 bytes): lanes run along
// the channel axis so every global access is a 16-byte-per-lane coalesced segment.
//
//   LDS-tiled stride-1 kernels (forward; fused backward = data + weight gradient + producer BatchNorm sums)
//   strided gather kernels (stride 1 | 2: forward, data gradient, weight gradient)
//   row reductions shared by the two-stage (deterministic) weight-gradient sums
#include "x3t.h"

__device__ __forceinline__ float4 f4_relu(float4 v) {
  return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ void f4_fma(float4& a, const float4 x, const float4 w) {
  a.x = fmaf(x.x, w.x, a.x);
  a.y = fmaf(x.y, w.y, a.y);
  a.z = fmaf(x.z, w.z, a.z);
  a.w = fmaf(x.w, w.w, a.w);
}

// ================================================================================================
// LDS-tiled depthwise kernels (the ones the engine uses).
//
// A workgroup owns TH x TW output pixels x CC4 channel quads.  Phase 1 stages the (TH+2) x (TW+2) halo
// tile into LDS with every global element fetched once per tile (16 B per lane, CC4 lanes contiguous
// along C, zero fill outside the image = SAME padding).  Phase 2: thread = (channel PAIR, column); it
// walks down its column with a rolled loop, keeping a three-row window of partial sums in registers
// (a tile row is read from LDS once: 3 x 8 B per row, a wave reads 512 contiguous bytes per
// instruction, conflict-free).  Register use is independent of the tile height, so the kernels stay at
// high occupancy; HBM sees ~1.2-1.3x the tensor on the read side and exactly the tensor on the write.
//
// The backward kernel fuses, in one pass over x and dz:
//   data gradient     dx = dw3x3(dz, flip w) * relu-mask + add
//   weight gradient   9 taps x C partial sums per workgroup (LDS combine, then reduce_rows)
//   BatchNorm backward sums of the PRODUCER layer (sum dx, sum dx*xhat) when that layer's affine is
//   applied here on load (its normalised output is never stored).
// ================================================================================================
__device__ __forceinline__ float2 f2_relu(float2 v) { return make_float2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f)); }
__device__ __forceinline__ void f2_fma(float2& a, const float2 x, const float2 w) {
  a.x = fmaf(x.x, w.x, a.x);
  a.y = fmaf(x.y, w.y, a.y);
}
__device__ __forceinline__ float2 f2_mul(const float2 x, const float2 w) { return make_float2(x.x * w.x, x.y * w.y); }

// stage a (TH+2) x (TW+2) x CC4 halo tile of `src` into LDS (float4 per lane, zero outside the image)
template <int CC4, int TW, int TH>
__device__ __forceinline__ void dw_stage_tile(float4* __restrict__ tile, const float* __restrict__ src, long ibase,
                                              int h0, int w0, int c40, int H, int W, int C, int c4n, int tid) {
  constexpr int PW = TW + 2;
  for (int idx = tid; idx < (TH + 2) * PW * CC4; idx += 256) {
    const int l = idx % CC4, p = idx / CC4;
    const int pw = p % PW, ph = p / PW;
    const int h = h0 - 1 + ph, w = w0 - 1 + pw, c4 = c40 + l;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h >= 0 && h < H && w >= 0 && w < W && c4 < c4n)
      v = *reinterpret_cast<const float4*>(src + ibase + ((long)h * W + w) * C + c4 * 4);
    tile[idx] = v;
  }
}

// The producer BatchNorm's forward finalize, folded into this kernel's prologue (fin.partial != NULL): every workgroup
// turns the [rows][2][C] column sums the producing GEMM's epilogue left behind into scale / shift for ITS OWN CC4*4
// channels -- the arithmetic of bn_fwd_finalize_kernel (bn.hip) in the same order: 16 interleaved groups of partial
// rows summed in double, the group sums added in group order, so the coefficients are bit-identical to the stand-alone
// kernel's and identical in every workgroup, and no workgroup ever waits for another -- and the workgroups of tile 0
// also publish mean / invstd / scale / shift (the backward pass reads them) and update the moving statistics.
// One dependent launch (~7 us + its boundary) less per BatchNorm for ~1 us of prologue; rows <= 128.
struct DwBnFinalize {
  const float* partial;     // [rows][2][C] (sum, sum of squares) or NULL
  const float* gamma;
  const float* beta;
  float* moving_mean;
  float* moving_var;
  float* save_mean;
  float* save_invstd;
  float* scale;             // out: [C]
  float* shift;             // out: [C]
  long M;                   // pixels the statistics were taken over
  int rows;
  float eps, momentum;
};

template <int CC4, int TW, int TH>
__global__ __launch_bounds__(256) void dw3x3_tile_fwd_kernel(const float* __restrict__ in,
                                                             const float* __restrict__ wt,
                                                             float* __restrict__ out, int H, int W, int C,
                                                             int relu_in, int tiles_h, int tiles_w,
                                                             int cchunks, const float* __restrict__ in_scale,
                                                             const float* __restrict__ in_shift, DwBnFinalize fin,
                                                             unsigned short* __restrict__ planes, long plane_stride) {
  // planes != NULL: the output goes out as the bf16x3 planes of the [B*H*W][C] matrix (x3t.h) -- the A operand of the
  // pointwise GEMMs that follow (forward, weight gradient) -- instead of fp32: 6 bytes per element written, no fp32 copy.
  constexpr int PW = TW + 2;
  constexpr int CC2 = 2 * CC4;
  static_assert(CC2 * TW == 256, "thread layout");
  extern __shared__ __attribute__((aligned(16))) float4 tile[];   // [(TH+2)][PW][CC4], raw values
  const int c4n = C >> 2;
  // XCD-aware order: the workgroups one XCD receives cover a contiguous run of (tile, channel chunk) ids, so
  // the channel chunks of one pixel tile (whose 128-byte lines interleave) and the halos of neighbouring tiles
  // meet in ONE L2 instead of eight.  Measured: 728-channel planes -17..-25 %, 256 channels neutral, 128
  // channels (4 chunks per tile) +3 % -> only from 8 chunks per tile on.
  int bid = (cchunks >= 8) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int cc = bid % cchunks; bid /= cchunks;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int b = bid / tiles_h;
  const int h0 = th * TH, w0 = tw * TW, c40 = cc * CC4;
  const long ibase = (long)b * H * W * C;
  const int tid = threadIdx.x;
  const int l2 = tid % CC2, tcol = tid / CC2;
  const int c2 = c40 * 2 + l2;                       // channel-pair index
  float2 sc = make_float2(1.f, 1.f), sh = make_float2(0.f, 0.f);
  const bool affine = in_scale != nullptr || fin.partial != nullptr;
  // With the finalize prologue the halo tile is fetched FIRST, into registers (the raw tensor does not depend on the
  // statistics): its global loads are in flight while the partial rows are fetched and combined, instead of
  // starting behind two barriers and a double-precision chain (the 12x16x728 planes: 16.1 us with the fetch behind
  // the prologue).  The tile buffer is the prologue's scratch, so the values wait in registers until it is dead.
  constexpr int TILE_F4 = (TH + 2) * PW * CC4, NPRE = (TILE_F4 + 255) / 256;
  float4 pre[NPRE];
  if (fin.partial) {
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int idx = tid + i * 256;
      const int l = idx % CC4, p = idx / CC4;
      const int pw = p % PW, ph = p / PW;
      const int h = h0 - 1 + ph, w = w0 - 1 + pw, c4 = c40 + l;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < TILE_F4 && h >= 0 && h < H && w >= 0 && w < W && c4 < c4n)
        v = *reinterpret_cast<const float4*>(in + ibase + ((long)h * W + w) * C + c4 * 4);
      pre[i] = v;
    }
    // (the tile buffer doubles as reduction scratch: 2*16*NCH doubles + 2*NCH floats fit every tile shape)
    constexpr int NCH = CC4 * 4, GR = 256 / NCH;     // channels of this workgroup, thread groups per channel
    static_assert(2 * 16 * NCH * 8 + 2 * NCH * 4 <= (TH + 2) * PW * CC4 * 16, "reduction scratch inside the tile");
    double* dred = reinterpret_cast<double*>(tile);  // [2][16][NCH]
    float* ssh = reinterpret_cast<float*>(dred + 2 * 16 * NCH);   // [2][NCH]
    const int ch = tid % NCH, grp = tid / NCH;
    const int cg = c40 * 4 + ch;
    for (int g = grp; g < 16; g += GR) {
      double s0 = 0.0, q0 = 0.0;
      if (cg < C)
        for (int p = g; p < fin.rows; p += 16) {
          s0 += (double)fin.partial[((long)p * 2 + 0) * C + cg];
          q0 += (double)fin.partial[((long)p * 2 + 1) * C + cg];
        }
      dred[(0 * 16 + g) * NCH + ch] = s0;
      dred[(1 * 16 + g) * NCH + ch] = q0;
    }
    __syncthreads();
    if (tid < NCH && cg < C) {
      double ss = 0.0, qq = 0.0;
#pragma unroll
      for (int g = 0; g < 16; ++g) { ss += dred[(0 * 16 + g) * NCH + ch]; qq += dred[(1 * 16 + g) * NCH + ch]; }
      const BnChannelStats st = bn_channel_stats(ss, qq, fin.M, fin.gamma[cg], fin.beta[cg], fin.eps);
      ssh[ch] = st.scale;
      ssh[NCH + ch] = st.shift;
      if (b == 0 && th == 0 && tw == 0) {            // one publisher per channel chunk
        fin.save_mean[cg] = st.mean;
        fin.save_invstd[cg] = st.invstd;
        fin.scale[cg] = st.scale;
        fin.shift[cg] = st.shift;
        fin.moving_mean[cg] = bn_moving_update(fin.moving_mean[cg], fin.momentum, st.mean);
        fin.moving_var[cg] = bn_moving_update(fin.moving_var[cg], fin.momentum, st.unbiased_var);
      }
    }
    __syncthreads();
    if (c2 * 2 < C) {
      sc = make_float2(ssh[l2 * 2], ssh[l2 * 2 + 1]);
      sh = make_float2(ssh[NCH + l2 * 2], ssh[NCH + l2 * 2 + 1]);
    }
    __syncthreads();                                 // scratch is dead: the tile may be staged over it
#pragma unroll
    for (int i = 0; i < NPRE; ++i)
      if (tid + i * 256 < TILE_F4) tile[tid + i * 256] = pre[i];
  } else {
    dw_stage_tile<CC4, TW, TH>(tile, in, ibase, h0, w0, c40, H, W, C, c4n, tid);
  }
  __syncthreads();
  if (c2 * 2 >= C) return;
  const float2* t2 = reinterpret_cast<const float2*>(tile);
  float2 k[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) k[tp] = *reinterpret_cast<const float2*>(wt + (long)tp * C + c2 * 2);
  if (in_scale != nullptr && fin.partial == nullptr) {
    sc = *reinterpret_cast<const float2*>(in_scale + c2 * 2);
    sh = *reinterpret_cast<const float2*>(in_shift + c2 * 2);
  }
  const int w = w0 + tcol;
  // output tile-row t (1..TH) = sum_kh in[t+kh-1] . k[kh]; input row rr feeds t = rr+1 (kh 0), rr (kh 1), rr-1 (kh 2)
  float2 a_prev = make_float2(0.f, 0.f), a_cur = a_prev;
#ifndef DW_FWD_UNROLL
#define DW_FWD_UNROLL 2
#endif
#pragma unroll DW_FWD_UNROLL
  for (int rr = 0; rr < TH + 2; ++rr) {
    const int o = ((rr * PW + tcol) * CC4) * 2 + l2;
    float2 v0 = t2[o], v1 = t2[o + CC2], v2 = t2[o + 2 * CC2];
    // zero padding must stay zero: the affine / relu only apply to pixels inside the image
    const int h = h0 - 1 + rr;
    const bool rv = (h >= 0 && h < H);
    if (affine) {
      const bool c0 = rv && (w - 1 >= 0), c1 = rv && (w < W), c2v = rv && (w + 1 < W);
      v0 = c0 ? make_float2(fmaf(v0.x, sc.x, sh.x), fmaf(v0.y, sc.y, sh.y)) : v0;
      v1 = c1 ? make_float2(fmaf(v1.x, sc.x, sh.x), fmaf(v1.y, sc.y, sh.y)) : v1;
      v2 = c2v ? make_float2(fmaf(v2.x, sc.x, sh.x), fmaf(v2.y, sc.y, sh.y)) : v2;
    }
    if (relu_in) { v0 = f2_relu(v0); v1 = f2_relu(v1); v2 = f2_relu(v2); }
    f2_fma(a_prev, v0, k[6]); f2_fma(a_prev, v1, k[7]); f2_fma(a_prev, v2, k[8]);
    f2_fma(a_cur, v0, k[3]); f2_fma(a_cur, v1, k[4]); f2_fma(a_cur, v2, k[5]);
    float2 a_next = f2_mul(v0, k[0]);
    f2_fma(a_next, v1, k[1]); f2_fma(a_next, v2, k[2]);
    const int t = rr - 1;                            // completed output tile-row
    const int ho = h0 + t - 1;
    if (t >= 1 && t <= TH && ho < H && w < W) {
      if (planes) x3t_store2(planes, plane_stride, x3t_off(((long)b * H + ho) * W + w, c2 * 2, (C + 31) >> 5), a_prev.x, a_prev.y);
      else *reinterpret_cast<float2*>(out + ibase + ((long)ho * W + w) * C + c2 * 2) = a_prev;
    }
    a_prev = a_cur;
    a_cur = a_next;
  }
}

// Backward.  With offsets d in {-1,0,1}^2 and forward y[p] = sum_d xin[p+d] k[d]:
//   dx[q]  = sum_d dz[q-d] k[d]          dk[d] = sum_q xin[q] dz[q-d]
// so BOTH gradients consume the same dz neighbourhood of the centre pixel q: only the dz halo tile goes
// through LDS, x (and the residual-branch gradient `add`) are read once, at the centres, straight from
// global memory (128 contiguous bytes per pixel and wave row), one row ahead of their use.
template <int CC4, int TW, int TH>
__global__ __launch_bounds__(256) void dw3x3_tile_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ wt,
    float* __restrict__ dx, float* __restrict__ partial, const float* __restrict__ add, int H, int W,
    int C, int relu_in, int tiles_h, int tiles_w, int cchunks, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* __restrict__ bn_mean,
    const float* __restrict__ bn_invstd, float* __restrict__ bn_partial, const float* __restrict__ bn_x) {
  constexpr int PW = TW + 2;
  constexpr int CC2 = 2 * CC4;
  constexpr int TILE = (TH + 2) * PW * CC4;          // float4 per tile
  static_assert(CC2 * TW == 256, "thread layout");
  extern __shared__ __attribute__((aligned(16))) float4 tile[];   // dz halo tile; >= 11*256 float2 of reduction scratch
  const int c4n = C >> 2;
  // XCD-aware order: the workgroups one XCD receives cover a contiguous run of (tile, channel chunk) ids, so
  // the channel chunks of one pixel tile (whose 128-byte lines interleave) and the halos of neighbouring tiles
  // meet in ONE L2 instead of eight.  Measured: 728-channel planes -17..-25 %, 256 channels neutral, 128
  // channels (4 chunks per tile) +3 % -> only from 8 chunks per tile on.
  int bid = (cchunks >= 8) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int cc = bid % cchunks; bid /= cchunks;
  const int sp = bid;                                // spatial workgroup index = partial row
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int b = bid / tiles_h;
  const int h0 = th * TH, w0 = tw * TW, c40 = cc * CC4;
  const long ibase = (long)b * H * W * C;
  const int tid = threadIdx.x;
  dw_stage_tile<CC4, TW, TH>(tile, dz, ibase, h0, w0, c40, H, W, C, c4n, tid);
  __syncthreads();
  const int l2 = tid % CC2, tcol = tid / CC2;
  const int c2 = c40 * 2 + l2;
  const bool active = c2 * 2 < C;
  const float2* tdz = reinterpret_cast<const float2*>(tile);
  const float2 zero2 = make_float2(0.f, 0.f);
  float2 accw[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) accw[tp] = zero2;
  float2 bsg = zero2, bsgx = zero2;                  // BatchNorm-backward sums of the producer
  const int w = w0 + tcol;
  if (active && w < W) {
    float2 k[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) k[tp] = *reinterpret_cast<const float2*>(wt + (long)tp * C + c2 * 2);
    float2 sc = make_float2(1.f, 1.f), sh = zero2, mu = zero2, is = zero2;
    const bool affine = in_scale != nullptr;
    if (affine) {
      sc = *reinterpret_cast<const float2*>(in_scale + c2 * 2);
      sh = *reinterpret_cast<const float2*>(in_shift + c2 * 2);
    }
    if (bn_partial) {
      mu = *reinterpret_cast<const float2*>(bn_mean + c2 * 2);
      is = *reinterpret_cast<const float2*>(bn_invstd + c2 * 2);
    }
    const long cbase = ibase + (long)w * C + c2 * 2;     // + h*W*C per row
    // dz window rows (r-1, r, r+1) x columns (w-1, w, w+1); row index r is tile-local (1..TH = outputs)
    float2 m0, m1, m2, c0, c1, c2r, n0, n1, n2;
    {
      const int o0 = ((0 * PW + tcol) * CC4) * 2 + l2, o1 = ((1 * PW + tcol) * CC4) * 2 + l2;
      m0 = tdz[o0]; m1 = tdz[o0 + CC2]; m2 = tdz[o0 + 2 * CC2];
      c0 = tdz[o1]; c1 = tdz[o1 + CC2]; c2r = tdz[o1 + 2 * CC2];
    }
    // x / add / bn_x at the centre pixels are read straight from global memory, PD rows ahead of their use
    // (a ring of PD register slots): one row of FMAs is far too short to cover a global load.
    constexpr int PD = 3;
    static_assert(TH % PD == 0, "prefetch ring");
    float2 xq[PD], aq[PD], bq[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      xq[d] = zero2; aq[d] = zero2; bq[d] = zero2;
      const int hh = h0 + d;
      if (hh < H) {
        xq[d] = *reinterpret_cast<const float2*>(x + cbase + (long)hh * W * C);
        if (add) aq[d] = *reinterpret_cast<const float2*>(add + cbase + (long)hh * W * C);
        if (bn_x) bq[d] = *reinterpret_cast<const float2*>(bn_x + cbase + (long)hh * W * C);
      }
    }
#pragma unroll
    for (int t = 1; t <= TH; ++t) {
      const int ho = h0 + t - 1;
      if (ho >= H) break;
      const int o2 = (((t + 1) * PW + tcol) * CC4) * 2 + l2;
      n0 = tdz[o2]; n1 = tdz[o2 + CC2]; n2 = tdz[o2 + 2 * CC2];
      const int slot = (t - 1) % PD;
      const float2 raw = xq[slot], ad = aq[slot];
      const float2 pre = bn_x ? bq[slot] : raw;       // pre-BN value the statistics' BatchNorm normalised
      if (t - 1 + PD < TH && ho + PD < H) {           // refill the slot with the row PD ahead
        const long ro = cbase + (long)(ho + PD) * W * C;
        xq[slot] = *reinterpret_cast<const float2*>(x + ro);
        if (add) aq[slot] = *reinterpret_cast<const float2*>(add + ro);
        if (bn_x) bq[slot] = *reinterpret_cast<const float2*>(bn_x + ro);
      }
      float2 a = raw;
      if (affine) a = make_float2(fmaf(raw.x, sc.x, sh.x), fmaf(raw.y, sc.y, sh.y));
      const float2 xin = relu_in ? f2_relu(a) : a;
      // dz[q-d]: d = (kh-1, kw-1)  ->  window row (1-kh)+1 -> kh=0: next row (n*), kh=1: centre, kh=2: previous (m*);
      //                                 column kw=0: right (index 2), kw=1: centre, kw=2: left (index 0)
      float2 res = f2_mul(n2, k[0]);
      f2_fma(res, n1, k[1]); f2_fma(res, n0, k[2]);
      f2_fma(res, c2r, k[3]); f2_fma(res, c1, k[4]); f2_fma(res, c0, k[5]);
      f2_fma(res, m2, k[6]); f2_fma(res, m1, k[7]); f2_fma(res, m0, k[8]);
      f2_fma(accw[0], xin, n2); f2_fma(accw[1], xin, n1); f2_fma(accw[2], xin, n0);
      f2_fma(accw[3], xin, c2r); f2_fma(accw[4], xin, c1); f2_fma(accw[5], xin, c0);
      f2_fma(accw[6], xin, m2); f2_fma(accw[7], xin, m1); f2_fma(accw[8], xin, m0);
      if (relu_in) {
        res.x = a.x > 0.f ? res.x : 0.f;
        res.y = a.y > 0.f ? res.y : 0.f;
      }
      if (add) { res.x += ad.x; res.y += ad.y; }
      *reinterpret_cast<float2*>(dx + cbase + (long)ho * W * C) = res;
      if (bn_partial) {   // res = dL/d(BN output of the producer); xhat from the pre-BN value
        bsg.x += res.x; bsg.y += res.y;
        bsgx.x = fmaf(res.x, (pre.x - mu.x) * is.x, bsgx.x);
        bsgx.y = fmaf(res.y, (pre.y - mu.y) * is.y, bsgx.y);
      }
      m0 = c0; m1 = c1; m2 = c2r;
      c0 = n0; c1 = n1; c2r = n2;
    }
  }
  // combine the 9 tap sums (+ the 2 BatchNorm sums) over the TW threads of each channel pair, fixed order
  __syncthreads();
  float2* red = reinterpret_cast<float2*>(tile);
  const int grp = tid / CC2;                         // 0..TW-1
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) red[(tp * TW + grp) * CC2 + l2] = accw[tp];
  red[(9 * TW + grp) * CC2 + l2] = bsg;
  red[(10 * TW + grp) * CC2 + l2] = bsgx;
  __syncthreads();
  for (int q = tid; q < 11 * CC2; q += 256) {
    const int tp = q / CC2, ll = q % CC2;
    const int c2o = c40 * 2 + ll;
    if (c2o * 2 < C && (tp < 9 || bn_partial)) {
      float2 s = red[(tp * TW) * CC2 + ll];
      for (int g = 1; g < TW; ++g) {
        const float2 v = red[(tp * TW + g) * CC2 + ll];
        s.x += v.x; s.y += v.y;
      }
      if (tp < 9) *reinterpret_cast<float2*>(partial + ((long)sp * 9 + tp) * C + c2o * 2) = s;
      else *reinterpret_cast<float2*>(bn_partial + ((long)sp * 2 + (tp - 9)) * C + c2o * 2) = s;
    }
  }
}

// ================================================================================================
// Streaming depthwise kernels (round 4): for planes that are large enough to be pure bandwidth (the entry flow at batch
// 32, every plane of the batch-128 inference plan) the tile kernels above leave one load -> barrier -> compute -> store
// chain per workgroup with nothing in flight behind it and re-read a 2-row halo per 12-row tile.  Here a WAVE is the
// unit and nothing is shared: it owns SW = 64/CL columns x CL channel quads (CL x 16 contiguous bytes per pixel) and
// MARCHES DOWN its rows.  Each lane fetches only its own column -- PD rows ahead, in a register ring, so PD (+PD edge)
// loads are in flight per lane at any time -- the left / right neighbours come from the adjacent lanes (ds_bpermute: no
// LDS memory, no barrier), and the two columns just outside the wave's strip from one extra, 2 x CL-lane load per row.
// Three running rows of partial sums turn every input row into one output row: no tile, no H halo (except at the row
// segments the host cuts long planes into to fill the chip), the same fmaf order as the tile kernels (bit-identical).
// ================================================================================================
// (native 4-float vectors: selects and loads on them stay single dwordx4 / v_cndmask operations; every load below is
// UNCONDITIONAL on a clamped, always-valid address and masked afterwards, so the loop bodies are branch-free and the
// compiler's waitcnt insertion keeps PD rows of loads in flight instead of draining the queue at every block boundary)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f v4_ld(const float* p) { return *reinterpret_cast<const v4f*>(p); }
__device__ __forceinline__ void v4_st(float* p, const v4f v) { *reinterpret_cast<v4f*>(p) = v; }
__device__ __forceinline__ v4f v4_shfl_up(const v4f v, int d) {
  return v4f{__shfl_up(v.x, d, 64), __shfl_up(v.y, d, 64), __shfl_up(v.z, d, 64), __shfl_up(v.w, d, 64)};
}
__device__ __forceinline__ v4f v4_shfl_down(const v4f v, int d) {
  return v4f{__shfl_down(v.x, d, 64), __shfl_down(v.y, d, 64), __shfl_down(v.z, d, 64), __shfl_down(v.w, d, 64)};
}
__device__ __forceinline__ v4f v4_fma(const v4f a, const v4f b, const v4f c) {
  return v4f{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w)};
}
__device__ __forceinline__ v4f v4_relu(const v4f v) {
  return v4f{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
}
__device__ __forceinline__ v4f v4_sel(bool c, const v4f a, const v4f b) { return c ? a : b; }
// The streamed tensors go through raw BUFFER loads / stores (one descriptor per image): a lane whose byte offset lies
// beyond the descriptor's range reads zeros / stores nothing WITHOUT touching memory and without a branch -- lanes outside
// the image or the channel range, and the 48 lanes that take no part in the edge-column load, are switched off by
// giving them DW_OOB as their offset.  The row advance travels in the scalar offset (not range-checked: rows are
// clamped to the image and masked afterwards).  32-bit offsets: the entry points require H*W*C*4 < 2^31 per image.
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int DW_OOB = 0x7ffffff0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dw_rsrc(const float* p, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
// (cache policy left at the default: nontemporal stores bought +5 % on the 93x125 / 47x63 planes in isolation and cost
// 7-25 % on the 24x32 / 12x16 ones, nontemporal loads cost everywhere -- the edge columns and row-segment halos live
// on cache hits; profiles/r04_b_diag_dw_stream_sweep.txt)
__device__ __forceinline__ v4f dw_bl(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void dw_bs(__amdgpu_buffer_rsrc_t r, const v4f v, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), r, voff, soff, 0);
}

// wave id -> (batch, row segment, column strip, channel chunk), channel chunk fastest: the four waves of a workgroup
// read 4 x CL x 16 contiguous bytes per pixel where the plane has that many channels
struct DwStreamGeom { int strips, cchunks, segs, rows_per_seg; long waves; };

template <int CL, int PD, bool X3>
__global__ __launch_bounds__(256) void dw3x3_stream_fwd_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                               float* __restrict__ out, int H, int W, int C, int relu_in,
                                                               const float* __restrict__ in_scale,
                                                               const float* __restrict__ in_shift, DwStreamGeom g,
                                                               unsigned short* __restrict__ planes, long plane_stride) {
  // X3: the output as bf16x3 planes (see dw3x3_tile_fwd_kernel; a template parameter, so that a row of the march stays ONE
  // basic block with counted waits); a wave's store then covers 8 consecutive
  // pixels x 32 channels = 512 contiguous bytes of a piece, three times.
  constexpr int SW = 64 / CL;
  const int lane = threadIdx.x & 63;
  // (the wave index is uniform: readfirstlane tells the compiler, so that descriptors and row offsets live in SGPRs)
  long gw = (long)xcd_remap(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (gw >= g.waves) return;                          // whole wave
  const int cc = (int)(gw % g.cchunks); gw /= g.cchunks;
  const int strip = (int)(gw % g.strips); gw /= g.strips;
  const int seg = (int)(gw % g.segs);
  const int b = (int)(gw / g.segs);
  const int l = lane % CL, col = lane / CL;
  const int c4n = C >> 2;
  const int c4 = cc * CL + l, w = strip * SW + col;
  const bool own_ok = c4 < c4n && w < W;
  // the one column outside the strip this lane fetches: the left one (lanes of column 0), the right one (column SW-1)
  const int we = col == 0 ? w - 1 : w + 1;
  const bool edge_lane = col == 0 || col == SW - 1;
  const bool edge_ok = c4 < c4n && edge_lane && we >= 0 && we < W;
  const int h_lo = seg * g.rows_per_seg;
  const int h_hi = min(H, h_lo + g.rows_per_seg);     // output rows [h_lo, h_hi)
  const long rstride = (long)W * C;
  const int rbytes = (int)rstride * 4;                // bytes per image row
  const __amdgpu_buffer_rsrc_t rs_in = dw_rsrc(in + (long)b * H * rstride, H * rbytes);
  const __amdgpu_buffer_rsrc_t rs_out = X3 ? __builtin_amdgcn_make_buffer_rsrc(planes, 0, (int)(3 * plane_stride * 2), 0x00020000)
                                               : dw_rsrc(out + (long)b * H * rstride, H * rbytes);
  const int nk = (C + 31) >> 5;
  const int pbytes = (int)(plane_stride * 2);
  const int vo = own_ok ? (w * C + c4 * 4) * 4 : DW_OOB;          // byte offsets inside an image row
  const int ve = edge_ok ? (we * C + c4 * 4) * 4 : DW_OOB;
  const int c4c = min(c4, c4n - 1) * 4;               // (clamped: per-channel vectors are fetched by every lane)
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
  v4f k[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) k[tp] = v4_ld(wt + (long)tp * C + c4c);
  const bool affine = in_scale != nullptr;
  v4f sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4;
  if (affine) {
    sc = v4_ld(in_scale + c4c);
    sh = v4_ld(in_shift + c4c);
  }
  // input rows r_first .. r_last feed output rows h_lo .. h_hi-1; rows outside the image contribute zeros (SAME
  // padding); loads beyond r_last (the ring runs PD rows ahead) re-read the last row and are never used
  const int r_first = h_lo - 1, r_last = h_hi;
  const int r_max = min(r_last, H - 1);
  v4f ring[PD], ering[PD];
#pragma unroll
  for (int d = 0; d < PD; ++d) {
    const int ro = min(max(r_first + d, 0), r_max) * rbytes;
    ring[d] = dw_bl(rs_in, vo, ro);
    ering[d] = dw_bl(rs_in, ve, ro);
  }
  v4f a_prev = zero4, a_cur = zero4;
  const int niter = (r_last - r_first + PD) / PD;     // ceil(rows / PD)
  for (int it = 0; it < niter; ++it) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int r = r_first + it * PD + d;
      v4f v1 = ring[d], e = ering[d];
      {                                               // refill the slot with the row PD ahead
        const int ro = min(r + PD, r_max) * rbytes;
        ring[d] = dw_bl(rs_in, vo, ro);
        ering[d] = dw_bl(rs_in, ve, ro);
      }
      const bool rin = r >= 0 && r < H;               // wave-uniform
      // zero padding must stay zero: the affine / relu only apply to pixels inside the image
      if (affine) { v1 = v4_fma(v1, sc, sh); e = v4_fma(e, sc, sh); }
      if (relu_in) { v1 = v4_relu(v1); e = v4_relu(e); }
      v1 = v4_sel(rin && own_ok, v1, zero4);
      e = v4_sel(rin && edge_ok, e, zero4);
      v4f v0 = v4_shfl_up(v1, CL), v2 = v4_shfl_down(v1, CL);
      v0 = v4_sel(col == 0, e, v0);
      v2 = v4_sel(col == SW - 1, e, v2);
      // output row t = sum_kh in[t+kh-1] . k[kh]; input row r feeds t = r+1 (kh 0), r (kh 1), r-1 (kh 2)
      a_prev = v4_fma(v0, k[6], a_prev); a_prev = v4_fma(v1, k[7], a_prev); a_prev = v4_fma(v2, k[8], a_prev);
      a_cur = v4_fma(v0, k[3], a_cur); a_cur = v4_fma(v1, k[4], a_cur); a_cur = v4_fma(v2, k[5], a_cur);
      v4f a_next = v0 * k[0];
      a_next = v4_fma(v1, k[1], a_next); a_next = v4_fma(v2, k[2], a_next);
      const int t = r - 1;                            // completed output row
      if (X3) {
        unsigned h0, m0, l0, h1, m1, l1;
        x3_split_pk(a_prev.x, a_prev.y, h0, m0, l0);
        x3_split_pk(a_prev.z, a_prev.w, h1, m1, l1);
        const int m = (b * H + min(max(t, 0), H - 1)) * W + w;
        const int po = (t >= h_lo && t < h_hi && own_ok) ? x3t_off32_bytes(m, c4 * 4, nk) : DW_OOB;
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(v2u{h0, h1}, rs_out, po, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(v2u{m0, m1}, rs_out, po, pbytes, 0);
        __builtin_amdgcn_raw_buffer_store_b64(v2u{l0, l1}, rs_out, po, 2 * pbytes, 0);
      } else {
        dw_bs(rs_out, a_prev, (t >= h_lo && t < h_hi) ? vo : DW_OOB, min(max(t, 0), H - 1) * rbytes);
      }
      a_prev = a_cur;
      a_cur = a_next;
    }
  }
}

// Streaming fused backward (same outputs as dw3x3_tile_bwd_kernel: dx, per-"row" weight-gradient partial sums, the
// producer BatchNorm's two backward sums).  The wave marches down its strip with a 3-row window of dz (own column from
// the register ring, neighbours from the adjacent lanes, the two outside columns from the edge load); x, the
// residual-branch gradient `add` and bn_x are only needed at the centre pixel: own-column rings, PD rows ahead.  The 9 tap
// sums and the 2 BatchNorm sums stay in registers for the whole march and are folded over the wave's SW columns by a
// butterfly at the end (fixed order): ONE partial row per (batch, row segment, column strip), written by the waves of
// its channel chunks.  dx has the tile kernel's fmaf order (bit-identical); the partial sums group differently.
template <int CL, int PD, bool ADD, bool BNX>
__global__ __launch_bounds__(256) void dw3x3_stream_bwd_kernel(
    const float* __restrict__ dz, const float* __restrict__ x, const float* __restrict__ wt, float* __restrict__ dx,
    float* __restrict__ partial, const float* __restrict__ add, int H, int W, int C, int relu_in,
    const float* __restrict__ in_scale, const float* __restrict__ in_shift, const float* __restrict__ bn_mean,
    const float* __restrict__ bn_invstd, float* __restrict__ bn_partial, const float* __restrict__ bn_x, DwStreamGeom g) {
  constexpr int SW = 64 / CL;
  const int lane = threadIdx.x & 63;
  // (the wave index is uniform: readfirstlane tells the compiler, so that descriptors and row offsets live in SGPRs)
  long gw = (long)xcd_remap(blockIdx.x, gridDim.x) * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (gw >= g.waves) return;                          // whole wave
  const int cc = (int)(gw % g.cchunks); gw /= g.cchunks;
  const long sp = gw;                                 // partial row: (batch, segment, strip)
  const int strip = (int)(gw % g.strips); gw /= g.strips;
  const int seg = (int)(gw % g.segs);
  const int b = (int)(gw / g.segs);
  const int l = lane % CL, col = lane / CL;
  const int c4n = C >> 2;
  const int c4 = cc * CL + l, w = strip * SW + col;
  const bool chan_ok = c4 < c4n;
  const bool own_ok = chan_ok && w < W;
  const int we = col == 0 ? w - 1 : w + 1;
  const bool edge_lane = col == 0 || col == SW - 1;
  const bool edge_ok = chan_ok && edge_lane && we >= 0 && we < W;
  const int h_lo = seg * g.rows_per_seg;
  const int h_hi = min(H, h_lo + g.rows_per_seg);     // output rows [h_lo, h_hi)
  const long rstride = (long)W * C;
  const int rbytes = (int)rstride * 4;                // bytes per image row
  const long ibase = (long)b * H * rstride;
  const __amdgpu_buffer_rsrc_t rs_dz = dw_rsrc(dz + ibase, H * rbytes), rs_x = dw_rsrc(x + ibase, H * rbytes);
  const __amdgpu_buffer_rsrc_t rs_dx = dw_rsrc(dx + ibase, H * rbytes);
  const __amdgpu_buffer_rsrc_t rs_add = dw_rsrc(ADD ? add + ibase : x, ADD ? H * rbytes : 0);
  const __amdgpu_buffer_rsrc_t rs_bx = dw_rsrc(BNX ? bn_x + ibase : x, BNX ? H * rbytes : 0);
  const int vo = own_ok ? (w * C + c4 * 4) * 4 : DW_OOB;          // byte offsets inside an image row
  const int ve = edge_ok ? (we * C + c4 * 4) * 4 : DW_OOB;
  const int c4c = min(c4, c4n - 1) * 4;               // (clamped: per-channel vectors are fetched by every lane)
  const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
  v4f k[9], accw[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    k[tp] = v4_ld(wt + (long)tp * C + c4c);
    accw[tp] = zero4;
  }
  v4f bsg = zero4, bsgx = zero4;
  const bool affine = in_scale != nullptr;
  v4f sc = {1.f, 1.f, 1.f, 1.f}, sh = zero4, mu = zero4, is = zero4;
  if (affine) {
    sc = v4_ld(in_scale + c4c);
    sh = v4_ld(in_shift + c4c);
  }
  if (bn_partial) {
    mu = v4_ld(bn_mean + c4c);
    is = v4_ld(bn_invstd + c4c);
  }
  // dz rows r_first .. r_last feed output rows h_lo .. h_hi-1 (row r completes output row r-1); the centre tensors of
  // output row t = r - 1 travel in the slot of dz row r (the two rows in front of h_lo are fetched and ignored: the
  // loop body stays branch-free)
  const int r_first = h_lo - 1, r_last = h_hi;
  const int r_max = min(r_last, H - 1);
  v4f zr[PD], ze[PD], xr[PD], ar[PD], br[PD];
#pragma unroll
  for (int d = 0; d < PD; ++d) {
    const int ro = min(max(r_first + d, 0), r_max) * rbytes;
    zr[d] = dw_bl(rs_dz, vo, ro);
    ze[d] = dw_bl(rs_dz, ve, ro);
    const int to = min(max(r_first + d - 1, 0), h_hi - 1) * rbytes;       // centre tensors of output row r - 1
    xr[d] = dw_bl(rs_x, vo, to);
    if (ADD) ar[d] = dw_bl(rs_add, vo, to);
    if (BNX) br[d] = dw_bl(rs_bx, vo, to);
  }
  // the 3-row dz window as a ring of its own: the row that arrives in step d is written to win[d % 3]; the centre row
  // is then win[(d+2) % 3] and the previous one win[(d+1) % 3] -- static roles (PD % 3 == 0), no register moves
  static_assert(PD % 3 == 0, "window roles rotate with the unrolled steps");
  v4f win[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) win[i][0] = win[i][1] = win[i][2] = zero4;
  const int niter = (r_last - r_first + PD) / PD;     // ceil(rows / PD)
  for (int it = 0; it < niter; ++it) {
#pragma unroll
    for (int d = 0; d < PD; ++d) {
      const int r = r_first + it * PD + d;
      v4f n1 = zr[d], e = ze[d];
      {
        const int ro = min(r + PD, r_max) * rbytes;
        zr[d] = dw_bl(rs_dz, vo, ro);
        ze[d] = dw_bl(rs_dz, ve, ro);
      }
      const bool rin = r >= 0 && r < H;               // wave-uniform (lanes outside the image already read zeros)
      n1 = v4_sel(rin, n1, zero4);
      e = v4_sel(rin, e, zero4);
      v4f n0 = v4_shfl_up(n1, CL), n2 = v4_shfl_down(n1, CL);
      n0 = v4_sel(col == 0, e, n0);
      n2 = v4_sel(col == SW - 1, e, n2);
      win[d % 3][0] = n0; win[d % 3][1] = n1; win[d % 3][2] = n2;
      const v4f c0 = win[(d + 2) % 3][0], c1 = win[(d + 2) % 3][1], c2 = win[(d + 2) % 3][2];
      const v4f m0 = win[(d + 1) % 3][0], m1 = win[(d + 1) % 3][1], m2 = win[(d + 1) % 3][2];
      const int t = r - 1;
      const v4f raw = xr[d];
      v4f ad = zero4, pre = raw;
      if (ADD) ad = ar[d];
      if (BNX) pre = br[d];
      {                                               // refill the centre slot with the row PD ahead
        const int to = min(max(t + PD, 0), h_hi - 1) * rbytes;
        xr[d] = dw_bl(rs_x, vo, to);
        if (ADD) ar[d] = dw_bl(rs_add, vo, to);
        if (BNX) br[d] = dw_bl(rs_bx, vo, to);
      }
      const bool tin = t >= h_lo && t < h_hi;         // wave-uniform: the first two rows only fill the window
      v4f a = raw;
      if (affine) a = v4_fma(raw, sc, sh);
      const v4f xin = v4_sel(tin && own_ok, relu_in ? v4_relu(a) : a, zero4);
      // dz[q-d]: kh = 0 -> next row (n*), 1 -> centre (c*), 2 -> previous (m*); kw = 0 -> right (index 2), 2 -> left (0)
      v4f res = n2 * k[0];
      res = v4_fma(n1, k[1], res); res = v4_fma(n0, k[2], res);
      res = v4_fma(c2, k[3], res); res = v4_fma(c1, k[4], res); res = v4_fma(c0, k[5], res);
      res = v4_fma(m2, k[6], res); res = v4_fma(m1, k[7], res); res = v4_fma(m0, k[8], res);
      // (xin is zero for lanes / rows that are not output pixels: their tap sums stay untouched)
      accw[0] = v4_fma(xin, n2, accw[0]); accw[1] = v4_fma(xin, n1, accw[1]); accw[2] = v4_fma(xin, n0, accw[2]);
      accw[3] = v4_fma(xin, c2, accw[3]); accw[4] = v4_fma(xin, c1, accw[4]); accw[5] = v4_fma(xin, c0, accw[5]);
      accw[6] = v4_fma(xin, m2, accw[6]); accw[7] = v4_fma(xin, m1, accw[7]); accw[8] = v4_fma(xin, m0, accw[8]);
      if (relu_in) {
        res.x = a.x > 0.f ? res.x : 0.f; res.y = a.y > 0.f ? res.y : 0.f;
        res.z = a.z > 0.f ? res.z : 0.f; res.w = a.w > 0.f ? res.w : 0.f;
      }
      if (ADD) res += ad;
      dw_bs(rs_dx, res, tin ? vo : DW_OOB, min(max(t, 0), H - 1) * rbytes);
      if (bn_partial) {   // res = dL/d(BN output of the producer); xhat from the pre-BN value
        const v4f rm = v4_sel(tin && own_ok, res, zero4);
        bsg += rm;
        bsgx = v4_fma(rm, (pre - mu) * is, bsgx);
      }
    }
  }
  // fold the 9 + 2 sums over the SW columns of the wave (butterfly over the column bits of the lane id: fixed order)
#pragma unroll
  for (int off = CL; off < 64; off <<= 1) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      accw[tp].x += __shfl_xor(accw[tp].x, off, 64); accw[tp].y += __shfl_xor(accw[tp].y, off, 64);
      accw[tp].z += __shfl_xor(accw[tp].z, off, 64); accw[tp].w += __shfl_xor(accw[tp].w, off, 64);
    }
    bsg.x += __shfl_xor(bsg.x, off, 64); bsg.y += __shfl_xor(bsg.y, off, 64);
    bsg.z += __shfl_xor(bsg.z, off, 64); bsg.w += __shfl_xor(bsg.w, off, 64);
    bsgx.x += __shfl_xor(bsgx.x, off, 64); bsgx.y += __shfl_xor(bsgx.y, off, 64);
    bsgx.z += __shfl_xor(bsgx.z, off, 64); bsgx.w += __shfl_xor(bsgx.w, off, 64);
  }
  if (col == 0 && chan_ok) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) v4_st(partial + (sp * 9 + tp) * C + c4 * 4, accw[tp]);
    if (bn_partial) {
      v4_st(bn_partial + (sp * 2 + 0) * C + c4 * 4, bsg);
      v4_st(bn_partial + (sp * 2 + 1) * C + c4 * 4, bsgx);
    }
  }
}

// ================================================================================================
// Strided depthwise 3x3 / SAME (keras.applications.mobilenet: DepthwiseConv2D((3,3), padding='same', strides=(2,2)),
// call site spnet/models.py:346-355).  TF SAME: out = ceil(in/s), pad_total = max((out-1)*s + 3 - in, 0), floor(half)
// before.  One thread per (pixel, channel quad); these layers are a small share of a MobileNet step, so the plain
// gather forms are used (the stride-1 layers run the LDS-tiled kernels above).
//   fwd        y[b,oh,ow,c]  = sum_k x[b, oh*s-pt+kh, ow*s-pl+kw, c] w[k,c]
//   bwd_data   dx[b,h,w,c]   = sum over the windows containing (h,w) of dy * w[tap]
//   bwd_weight dw[k,c]       = sum_{b,oh,ow} x[...] dy[b,oh,ow,c]      (per-workgroup partial rows + reduce_rows)
// ================================================================================================
__global__ __launch_bounds__(256) void dw3x3_strided_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                                float* __restrict__ y, int Bn, int H, int W, int C,
                                                                int OH, int OW, int s, int pt, int pl) {
  const int c4n = C >> 2;
  const long total = (long)Bn * OH * OW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    const int c = c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = oh * s - pt + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = ow * s - pl + kw;
        if (w < 0 || w >= W) continue;
        f4_fma(acc, *reinterpret_cast<const float4*>(x + (((long)b * H + h) * W + w) * C + c),
               *reinterpret_cast<const float4*>(wt + (long)(kh * 3 + kw) * C + c));
      }
    }
    *reinterpret_cast<float4*>(y + i * 4) = acc;
  }
}

__global__ __launch_bounds__(256) void dw3x3_strided_bwd_data_kernel(const float* __restrict__ dy,
                                                                     const float* __restrict__ wt,
                                                                     float* __restrict__ dx, int Bn, int H, int W, int C,
                                                                     int OH, int OW, int s, int pt, int pl) {
  const int c4n = C >> 2;
  const long total = (long)Bn * H * W * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    const int c = c4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int hh = h + pt - kh;                    // = oh * s
      if (hh < 0 || hh % s) continue;
      const int oh = hh / s;
      if (oh >= OH) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ww = w + pl - kw;
        if (ww < 0 || ww % s) continue;
        const int ow = ww / s;
        if (ow >= OW) continue;
        f4_fma(acc, *reinterpret_cast<const float4*>(dy + (((long)b * OH + oh) * OW + ow) * C + c),
               *reinterpret_cast<const float4*>(wt + (long)(kh * 3 + kw) * C + c));
      }
    }
    *reinterpret_cast<float4*>(dx + i * 4) = acc;
  }
}

// blockDim = (CL, 256/CL): x over channel quads, y over output pixels; one partial [9][C] row per workgroup row.
__global__ __launch_bounds__(256) void dw3x3_strided_bwd_weight_kernel(const float* __restrict__ x,
                                                                       const float* __restrict__ dy,
                                                                       float* __restrict__ partial, int Bn, int H, int W,
                                                                       int C, int OH, int OW, int s, int pt, int pl) {
  extern __shared__ __attribute__((aligned(16))) float4 red[];   // [blockDim.y][9][blockDim.x]
  const int c4n = C >> 2;
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = c4 < c4n;
  const int c = c4 * 4;
  float4 acc[9];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) acc[tp] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (active) {
    const long npix = (long)Bn * OH * OW;
    for (long u = (long)blockIdx.y * blockDim.y + threadIdx.y; u < npix; u += (long)gridDim.y * blockDim.y) {
      const int ow = (int)(u % OW);
      long t = u / OW;
      const int oh = (int)(t % OH);
      const int b = (int)(t / OH);
      const float4 g = *reinterpret_cast<const float4*>(dy + u * C + c);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int h = oh * s - pt + kh;
        if (h < 0 || h >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int w = ow * s - pl + kw;
          if (w < 0 || w >= W) continue;
          f4_fma(acc[kh * 3 + kw], *reinterpret_cast<const float4*>(x + (((long)b * H + h) * W + w) * C + c), g);
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
      float4 sm = red[tp * bx + threadIdx.x];
      for (int yy = 1; yy < by; ++yy) {
        const float4 v = red[(yy * 9 + tp) * bx + threadIdx.x];
        sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
      }
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.y * 9 + tp) * C + c) = sm;
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

// Many independent row reductions in one launch: job j folds in_j[P_j][L_j] into out_j[L_j].  jobs (device
// memory) = njobs x {in pointer, out pointer, P, L} as four 64-bit words.  Same fixed order as
// reduce_rows_kernel with one slice: 16 interleaved row groups, then the groups in order.
__global__ __launch_bounds__(256) void reduce_rows_batched_kernel(const long long* __restrict__ jobs) {
  __shared__ float red[16][16];
  const long long* jb = jobs + 4 * blockIdx.y;
  const float* __restrict__ in = reinterpret_cast<const float*>(jb[0]);
  float* __restrict__ out = reinterpret_cast<float*>(jb[1]);
  const int P = (int)jb[2], L = (int)jb[3];
  if ((int)blockIdx.x * 16 >= L) return;           // whole workgroup leaves together
  const int lane = threadIdx.x & 15, g = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + lane;
  float s = 0.f;
  if (col < L) {
#pragma unroll 4
    for (int p = g; p < P; p += 16) s += in[(long)p * L + col];
  }
  red[g][lane] = s;
  __syncthreads();
  if (g == 0 && col < L) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][lane];
    out[col] = t;
  }
}

// max_L: the largest L_j (sizes the grid).
extern "C" int spnet_reduce_rows_batched(const void* jobs, int njobs, int max_L, void* stream) {
  if (!jobs || njobs < 1 || max_L < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(reduce_rows_batched_kernel, dim3((max_L + 15) / 16, njobs), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs));
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_reduce_rows(const float* in, int P, int L, float* out, void* stream) {
  launch_reduce_rows(in, P, L, out, nullptr, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}

// The same with 32*L floats of scratch: from 129 rows on, 32 row slices are folded side by side first (a bias gradient
// over tens of thousands of pixels otherwise runs on L/16 workgroups only).
extern "C" int spnet_reduce_rows_ws(const float* in, int P, int L, float* out, float* scratch, long scratch_floats,
                                    void* stream) {
  if (scratch && scratch_floats < 32L * L) return (int)hipErrorInvalidValue;
  launch_reduce_rows(in, P, L, out, scratch, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------- strided entry points
static void same_pad(int in, int s, int* out, int* before) {
  *out = (in + s - 1) / s;
  int total = (*out - 1) * s + 3 - in;
  if (total < 0) total = 0;
  *before = total / 2;
}

static int dws_rows(int B, int OH, int OW, int C) {
  const int cl = chan_lanes(C / 4);
  const int by = 256 / cl;
  long gy = ((long)B * OH * OW + (long)by * 8 - 1) / ((long)by * 8);
  if (gy > 256) gy = 256;
  return gy < 1 ? 1 : (int)gy;
}

// floats of workspace for spnet_dwconv3x3_strided_bwd_weight
extern "C" long spnet_dwconv3x3_strided_ws(int B, int H, int W, int C, int stride) {
  int OH, OW, pt, pl;
  same_pad(H, stride, &OH, &pt);
  same_pad(W, stride, &OW, &pl);
  return (long)dws_rows(B, OH, OW, C) * 9 * C;
}

// op 0: y = dw(x, w) ; op 1: dx = dw^T(dy, w) ; op 2: dw = sum x (x) dy (workspace: spnet_dwconv3x3_strided_ws floats).
// a = x (op 0, 2) or dy (op 1); b = w (op 0, 1) or dy (op 2); H, W = INPUT extent; stride 1 or 2.
extern "C" int spnet_dwconv3x3_strided(int op, const float* a, const float* b, float* out, int B, int H, int W, int C,
                                       int stride, float* workspace, void* stream) {
  if ((C & 3) || (stride != 1 && stride != 2) || op < 0 || op > 2) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  int OH, OW, pt, pl;
  same_pad(H, stride, &OH, &pt);
  same_pad(W, stride, &OW, &pl);
  if (op == 0) {
    const long total = (long)B * OH * OW * (C / 4);
    hipLaunchKernelGGL(dw3x3_strided_fwd_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, st, a, b, out, B, H, W, C,
                       OH, OW, stride, pt, pl);
  } else if (op == 1) {
    const long total = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(dw3x3_strided_bwd_data_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, st, a, b, out, B, H,
                       W, C, OH, OW, stride, pt, pl);
  } else {
    if (!workspace) return (int)hipErrorInvalidValue;
    const int cl = chan_lanes(C / 4), by = 256 / cl;
    const int gy = dws_rows(B, OH, OW, C);
    dim3 grid((C / 4 + cl - 1) / cl, gy), block(cl, by);
    hipLaunchKernelGGL(dw3x3_strided_bwd_weight_kernel, grid, block, (size_t)by * 9 * cl * sizeof(float4), st, a, b,
                       workspace, B, H, W, C, OH, OW, stride, pt, pl);
    launch_reduce_rows(workspace, gy, 9 * C, out, nullptr, st);
  }
  SPNET_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------- tiled entry points
struct DwGeom { int cfg, th, tw, cc4, tiles_h, tiles_w, cchunks; long nblk; size_t lds_fwd, lds_bwd; };
// cfg 1: 6x8 pixel tile x 16 channel quads (exit flow, 6x8 planes); cfg 0: 12x16 tile x 8 quads;
// cfg 2 (forward only): 24x16 tile x 8 quads for the large planes (halo read overhead 1.22 instead of 1.31)
static DwGeom dw_geom(int B, int H, int W, int C, bool fwd) {
  DwGeom g;
  g.cfg = (H <= 6 && W <= 8) ? 1 : 0;   // (a 24x16 forward tile was tried: 60 KB of LDS halves residency and loses)
  // (measured and dropped, round 2: 6x16 half-height tiles for the middle flow's 12x16 planes -- twice the workgroups,
  // forward 11.1 -> 10.8 us but backward 16.5 -> 18.4 us)
  g.th = g.cfg == 1 ? 6 : (g.cfg == 2 ? 24 : 12);
  g.tw = g.cfg == 1 ? 8 : 16;
  g.cc4 = g.cfg == 1 ? 16 : 8;
  g.tiles_h = (H + g.th - 1) / g.th;
  g.tiles_w = (W + g.tw - 1) / g.tw;
  g.cchunks = (C / 4 + g.cc4 - 1) / g.cc4;
  g.nblk = (long)B * g.tiles_h * g.tiles_w * g.cchunks;
  g.lds_fwd = (size_t)(g.th + 2) * (g.tw + 2) * g.cc4 * sizeof(float4);
  g.lds_bwd = g.lds_fwd > 11 * 256 * sizeof(float2) ? g.lds_fwd : 11 * 256 * sizeof(float2);   // dz tile / reduction scratch
  return g;
}

static bool dw_planes_ok(const void* p, int B, int H, int W, int C) {
  return p && !(((uintptr_t)p) & 15) && 3 * x3t_plane_elems((long)B * H * W, C) * 2 < (1L << 31);
}

static int dw_tiled_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C, int relu_in,
                        const float* in_scale, const float* in_shift, const DwBnFinalize& fin, void* stream,
                        unsigned short* planes = nullptr) {
  if (C & 3) return (int)hipErrorInvalidValue;
  if (planes && !dw_planes_ok(planes, B, H, W, C)) return (int)hipErrorInvalidValue;
  const long plane_stride = x3t_plane_elems((long)B * H * W, C);
  const DwGeom g = dw_geom(B, H, W, C, true);
#define DW_FWD(CC4, TW, TH)                                                                                     \
  hipLaunchKernelGGL((dw3x3_tile_fwd_kernel<CC4, TW, TH>), dim3((unsigned)g.nblk), dim3(256), g.lds_fwd,        \
                     (hipStream_t)stream, x, w, y, H, W, C, relu_in, g.tiles_h, g.tiles_w, g.cchunks, in_scale, \
                     in_shift, fin, planes, plane_stride)
  if (g.cfg == 1) DW_FWD(16, 8, 6);
  else if (g.cfg == 2) DW_FWD(8, 16, 24);
  else DW_FWD(8, 16, 12);
#undef DW_FWD
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_dwconv3x3_tiled_fwd(const float* x, const float* w, float* y, int B, int H, int W,
                                         int C, int relu_in, const float* in_scale,
                                         const float* in_shift, void* stream) {
  DwBnFinalize fin = {};
  return dw_tiled_fwd(x, w, y, B, H, W, C, relu_in, in_scale, in_shift, fin, stream);
}
// ... with the output written as the bf16x3 planes of the [B*H*W][C] matrix (csrc/x3t.h; zeroed allocation of
// 3 * spnet_bf16x3_plane_elems(B*H*W, C) bf16): the A operand of spnet_gemm_bf16x3_pp / _wgrad_batched; same arithmetic.
extern "C" int spnet_dwconv3x3_tiled_fwd_x3(const float* x, const float* w, void* y_planes, int B, int H, int W,
                                            int C, int relu_in, const float* in_scale,
                                            const float* in_shift, void* stream) {
  DwBnFinalize fin = {};
  if (!y_planes) return (int)hipErrorInvalidValue;
  return dw_tiled_fwd(x, w, nullptr, B, H, W, C, relu_in, in_scale, in_shift, fin, stream, reinterpret_cast<unsigned short*>(y_planes));
}

// The same with the producer BatchNorm's forward finalize folded into the prologue (training): partial[rows][2][C]
// are the column sums of the producer's pre-normalisation output (spnet_gemm_f32_colstats), M its row count; the kernel
// applies relu?(x*scale + shift) with the scale / shift it derives, writes save_mean / save_invstd / scale_shift[2C] and
// updates the moving statistics exactly like spnet_bn_finalize_fwd.  rows <= 128.
static int dw_tiled_fwd_bnfin(const float* x, const float* w, float* y, unsigned short* planes, int B, int H, int W, int C,
                              int relu_in, const float* partial, int rows, long M,
                              const float* gamma, const float* beta, float* moving_mean,
                              float* moving_var, float* save_mean, float* save_invstd,
                              float* scale_shift, float eps, float momentum, void* stream) {
  if (!partial || rows < 1 || rows > 128 || M < 1) return (int)hipErrorInvalidValue;
  DwBnFinalize fin;
  fin.partial = partial; fin.gamma = gamma; fin.beta = beta; fin.moving_mean = moving_mean; fin.moving_var = moving_var;
  fin.save_mean = save_mean; fin.save_invstd = save_invstd; fin.scale = scale_shift; fin.shift = scale_shift + C;
  fin.rows = rows; fin.M = M;
  fin.eps = eps; fin.momentum = momentum;
  return dw_tiled_fwd(x, w, y, B, H, W, C, relu_in, nullptr, nullptr, fin, stream, planes);
}
extern "C" int spnet_dwconv3x3_tiled_fwd_bnfin(const float* x, const float* w, float* y, int B, int H, int W, int C,
                                               int relu_in, const float* partial, int rows, long M,
                                               const float* gamma, const float* beta, float* moving_mean,
                                               float* moving_var, float* save_mean, float* save_invstd,
                                               float* scale_shift, float eps, float momentum, void* stream) {
  return dw_tiled_fwd_bnfin(x, w, y, nullptr, B, H, W, C, relu_in, partial, rows, M, gamma, beta, moving_mean, moving_var,
                            save_mean, save_invstd, scale_shift, eps, momentum, stream);
}
extern "C" int spnet_dwconv3x3_tiled_fwd_bnfin_x3(const float* x, const float* w, void* y_planes, int B, int H, int W, int C,
                                                  int relu_in, const float* partial, int rows, long M,
                                                  const float* gamma, const float* beta, float* moving_mean,
                                                  float* moving_var, float* save_mean, float* save_invstd,
                                                  float* scale_shift, float eps, float momentum, void* stream) {
  if (!y_planes) return (int)hipErrorInvalidValue;
  return dw_tiled_fwd_bnfin(x, w, nullptr, reinterpret_cast<unsigned short*>(y_planes), B, H, W, C, relu_in, partial, rows, M,
                            gamma, beta, moving_mean, moving_var, save_mean, save_invstd, scale_shift, eps, momentum, stream);
}

// ---------------------------------------------------------------- streaming entry points
// Loads in flight per lane: forward 4 rows (own + edge column), backward 3 (dz own + edge, x [, add, bn_x]).  Measured
// (tools/dw_stream_sweep.py, profiles/r04_b_diag_dw_stream_sweep.txt): twice the depth (8 / 6 rows, 170 / 310+ VGPRs) is
// never faster, not even on the 12-row planes -- occupancy costs more than the extra round trips save.
constexpr int DWS_CL = 8, DWS_SW = 64 / DWS_CL, DWS_PD_FWD = 4, DWS_PD_BWD = 3;

// rows_per_seg <= 0: the library's choice.  Backward: whole columns always (a segment re-reads 2 halo rows of dz AND adds
// a partial row; measured slower on every plane).  Forward: whole columns from 8 waves per CU on; below that the rows are
// cut into segments (2 halo rows re-read per segment, served by the cache) of at least 6 rows until the launch has ~16
// waves per CU (93x125x64 at batch 32: 46 -> 39 us with 24-row segments; 12x16x728: 13.6 -> 12.5 us with 6-row ones).
static DwStreamGeom dw_stream_geom(int B, int H, int W, int C, int rows_per_seg, bool backward) {
  DwStreamGeom g;
  g.strips = (W + DWS_SW - 1) / DWS_SW;
  g.cchunks = (C / 4 + DWS_CL - 1) / DWS_CL;
  const long base = (long)B * g.strips * g.cchunks;
  if (rows_per_seg <= 0) {
    long segs = (backward || base >= 2048) ? 1 : (4096 + base - 1) / base;
    rows_per_seg = (int)((H + segs - 1) / segs);
    if (rows_per_seg < 6) rows_per_seg = 6;
  }
  if (rows_per_seg > H) rows_per_seg = H;
  g.rows_per_seg = rows_per_seg;
  g.segs = (H + rows_per_seg - 1) / rows_per_seg;
  g.waves = base * g.segs;
  return g;
}

// Which form is faster for this plane (measured on MI355X with operands beyond the Infinity Cache, tiled / streaming,
// us at batch 32; profiles/r04_b_diag_dw_stream_sweep.txt):
//   forward   93x125x64 55 / 39   93x125x128 103 / 71   47x63x128 29 / 23   47x63x256 54 / 40   24x32x256 17 / 15
//             24x32x728 44 / 37   12x16x728 14.2 / 13.6   6x8x1024 9.0 / 7.4   6x8x1536 10.0 / 8.2      -> always
//   backward  93x125x64 72 / 59   93x125x128 125 / 113  47x63x128 37 / 34   47x63x256 70 / 63   24x32x256 20.8 / 20.5
//             24x32x728 54 / 54   12x16x728 17.4 / 19.9   6x8x1536 12.3 / 13.4            -> planes of >= 2,048 pixels
// (the forward form with the folded BatchNorm finalize, spnet_dwconv3x3_tiled_fwd_bnfin, exists in the tiled form only)
extern "C" long spnet_dwconv3x3_prefers_stream(int B, int H, int W, int C, int backward) {
  (void)B; (void)C;
  if (!backward) return 1;
  return (long)H * W >= 2048 ? 1 : 0;
}

// rows of the [rows][9][C] weight-gradient / [rows][2][C] BatchNorm partial buffers the streaming backward leaves
extern "C" long spnet_dwconv3x3_stream_rows(int B, int H, int W, int C, int rows_per_seg) {
  const DwStreamGeom g = dw_stream_geom(B, H, W, C, rows_per_seg, true);
  return (long)B * g.segs * g.strips;
}

extern "C" long spnet_dwconv3x3_stream_bwd_ws(int B, int H, int W, int C, int rows_per_seg) {
  return (spnet_dwconv3x3_stream_rows(B, H, W, C, rows_per_seg) + 32) * 9 * C;      // partial rows + 32 second-level slices
}

// y = dw3x3(relu?(x * in_scale + in_shift)), the arithmetic (and bits) of spnet_dwconv3x3_tiled_fwd
static int dw_stream_fwd(const float* x, const float* w, float* y, unsigned short* planes, int B, int H, int W, int C,
                         int relu_in, const float* in_scale, const float* in_shift, int rows_per_seg, void* stream) {
  if ((C & 3) || B < 1 || H < 1 || W < 1) return (int)hipErrorInvalidValue;
  if (planes && !dw_planes_ok(planes, B, H, W, C)) return (int)hipErrorInvalidValue;
  if ((long)H * W * C * 4 >= (1L << 31)) return (int)hipErrorInvalidValue;       // 32-bit byte offsets inside an image
  const DwStreamGeom g = dw_stream_geom(B, H, W, C, rows_per_seg, false);
  if ((g.waves + 3) / 4 > 0x7fffffffL) return (int)hipErrorInvalidValue;
  if (planes)
    hipLaunchKernelGGL((dw3x3_stream_fwd_kernel<DWS_CL, DWS_PD_FWD, true>), dim3((unsigned)((g.waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, x, w, y, H, W, C, relu_in, in_scale, in_shift, g, planes,
                       x3t_plane_elems((long)B * H * W, C));
  else
    hipLaunchKernelGGL((dw3x3_stream_fwd_kernel<DWS_CL, DWS_PD_FWD, false>), dim3((unsigned)((g.waves + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, x, w, y, H, W, C, relu_in, in_scale, in_shift, g, planes, 0L);
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_dwconv3x3_stream_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C,
                                          int relu_in, const float* in_scale, const float* in_shift, int rows_per_seg,
                                          void* stream) {
  return dw_stream_fwd(x, w, y, nullptr, B, H, W, C, relu_in, in_scale, in_shift, rows_per_seg, stream);
}
// ... with the bf16x3 planes output (see spnet_dwconv3x3_tiled_fwd_x3)
extern "C" int spnet_dwconv3x3_stream_fwd_x3(const float* x, const float* w, void* y_planes, int B, int H, int W, int C,
                                             int relu_in, const float* in_scale, const float* in_shift, int rows_per_seg,
                                             void* stream) {
  if (!y_planes) return (int)hipErrorInvalidValue;
  return dw_stream_fwd(x, w, nullptr, reinterpret_cast<unsigned short*>(y_planes), B, H, W, C, relu_in, in_scale, in_shift,
                       rows_per_seg, stream);
}

// The fused backward of spnet_dwconv3x3_tiled_bwd in streaming form: same arguments and outputs; the partial buffers
// have spnet_dwconv3x3_stream_rows(B,H,W,C,rows_per_seg) rows, workspace spnet_dwconv3x3_stream_bwd_ws floats.
extern "C" int spnet_dwconv3x3_stream_bwd(const float* dy, const float* x_fwd, const float* w, float* dx, float* dw, int B,
                                          int H, int W, int C, int relu_in, const float* add, float* workspace,
                                          const float* in_scale, const float* in_shift, const float* bn_mean,
                                          const float* bn_invstd, float* bn_partial, const float* bn_x, int rows_per_seg,
                                          void* stream) {
  if (bn_partial && (!bn_mean || !bn_invstd)) return (int)hipErrorInvalidValue;
  if (bn_x && !bn_partial) return (int)hipErrorInvalidValue;
  if ((C & 3) || B < 1 || H < 1 || W < 1 || !workspace) return (int)hipErrorInvalidValue;
  if ((long)H * W * C * 4 >= (1L << 31)) return (int)hipErrorInvalidValue;       // 32-bit byte offsets inside an image
  const DwStreamGeom g = dw_stream_geom(B, H, W, C, rows_per_seg, true);
  if ((g.waves + 3) / 4 > 0x7fffffffL) return (int)hipErrorInvalidValue;
#define DWS_BWD(ADD, BNX)                                                                                              \
  hipLaunchKernelGGL((dw3x3_stream_bwd_kernel<DWS_CL, DWS_PD_BWD, ADD, BNX>), dim3((unsigned)((g.waves + 3) / 4)),     \
                     dim3(256), 0, (hipStream_t)stream, dy, x_fwd, w, dx, workspace, add, H, W, C, relu_in, in_scale,  \
                     in_shift, bn_mean, bn_invstd, bn_partial, bn_x, g)
  if (add && bn_x) DWS_BWD(true, true);
  else if (add) DWS_BWD(true, false);
  else if (bn_x) DWS_BWD(false, true);
  else DWS_BWD(false, false);
#undef DWS_BWD
  const int P = (int)((long)B * g.segs * g.strips), L = 9 * C;
  if (dw) launch_reduce_rows(workspace, P, L, dw, workspace + (long)P * L, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" long spnet_dwconv3x3_tiled_rows(int B, int H, int W, int C) {
  const DwGeom g = dw_geom(B, H, W, C, false);
  return (long)B * g.tiles_h * g.tiles_w;   // rows of the [rows][2][C] BatchNorm partial buffer
}

// floats of workspace needed by spnet_dwconv3x3_tiled_bwd
extern "C" long spnet_dwconv3x3_tiled_bwd_ws(int B, int H, int W, int C) {
  const DwGeom g = dw_geom(B, H, W, C, false);
  return ((long)B * g.tiles_h * g.tiles_w + 32) * 9 * C;      // partial rows + 32 second-level slices
}

// Fused backward: dx = dw3x3(dy, flip w) * (xin > 0 if relu_in) (+ add);  dw[3][3][C] = weight gradient,
// xin = x_fwd*in_scale + in_shift when the producer's BatchNorm affine is fused (else x_fwd).  With
// bn_partial != NULL the kernel also emits the producer BatchNorm's backward sums (sum dx, sum dx*xhat)
// as [rows][2][C] partials, rows = spnet_dwconv3x3_tiled_rows().  bn_x (or NULL = x_fwd): the tensor that
// BatchNorm normalised, when it is not this layer's input itself -- a middle block's first unit reads the
// previous block's OUTPUT y = BN(yp) + residual, and produces the sums of that BN from yp.
extern "C" int spnet_dwconv3x3_tiled_bwd(const float* dy, const float* x_fwd, const float* w, float* dx,
                                         float* dw, int B, int H, int W, int C, int relu_in,
                                         const float* add, float* workspace, const float* in_scale,
                                         const float* in_shift, const float* bn_mean,
                                         const float* bn_invstd, float* bn_partial, const float* bn_x,
                                         void* stream) {
  if (bn_partial && (!bn_mean || !bn_invstd)) return (int)hipErrorInvalidValue;
  if (bn_x && !bn_partial) return (int)hipErrorInvalidValue;
  if (C & 3) return (int)hipErrorInvalidValue;
  const DwGeom g = dw_geom(B, H, W, C, false);
  if (g.cfg == 1)
    hipLaunchKernelGGL((dw3x3_tile_bwd_kernel<16, 8, 6>), dim3((unsigned)g.nblk), dim3(256), g.lds_bwd,
                       (hipStream_t)stream, dy, x_fwd, w, dx, workspace, add, H, W, C, relu_in, g.tiles_h,
                       g.tiles_w, g.cchunks, in_scale, in_shift, bn_mean, bn_invstd, bn_partial, bn_x);
  else
    hipLaunchKernelGGL((dw3x3_tile_bwd_kernel<8, 16, 12>), dim3((unsigned)g.nblk), dim3(256), g.lds_bwd,
                       (hipStream_t)stream, dy, x_fwd, w, dx, workspace, add, H, W, C, relu_in, g.tiles_h,
                       g.tiles_w, g.cchunks, in_scale, in_shift, bn_mean, bn_invstd, bn_partial, bn_x);
  const int P = B * g.tiles_h * g.tiles_w, L = 9 * C;
  // dw == NULL: leave the [rows][9][C] partial sums in `workspace`; the caller folds them later
  // (spnet_reduce_rows_batched does it for all layers of a step in one launch, off the critical path)
  if (dw) launch_reduce_rows(workspace, P, L, dw, workspace + (long)P * L, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}
