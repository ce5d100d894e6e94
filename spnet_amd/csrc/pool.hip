// Pooling layers of the hot path, NHWC fp32.
//   * MaxPooling2D((3,3), strides 2, padding 'same') of Xception blocks 2/3/4/13, fused with the
//     residual add that follows it (keras.applications.Xception, call site spnet/models.py:357-359).
//     TF SAME: pad_total = max((out-1)*2+3-in, 0), floor(pad_total/2) before, the rest after
//     (so an even extent pads 0 before / 1 after), padding value -inf.
//   * AveragePooling2D((2,2)) of the stem (spnet/models.py:323,337): VALID, floor(in/2).
#include "common.h"

// y = maxpool(x) + residual ; idx = argmax tap (kh*3+kw), one byte per element, first max wins.
__global__ __launch_bounds__(256) void maxpool_add_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ residual,
                                                              float* __restrict__ y,
                                                              uint32_t* __restrict__ idx4, int Bn,
                                                              int H, int W, int C, int OH, int OW,
                                                              int pt, int pl,
                                                              const float* __restrict__ x_ss,
                                                              const float* __restrict__ r_ss) {
  const int c4n = C >> 2;
  const long total = (long)Bn * OH * OW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    const float ninf = -__builtin_huge_valf();
    float4 xs = make_float4(1.f, 1.f, 1.f, 1.f), xh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (x_ss) {   // BatchNorm affine of the pooled branch applied on load: [scale[C] | shift[C]]
      xs = *reinterpret_cast<const float4*>(x_ss + c4 * 4);
      xh = *reinterpret_cast<const float4*>(x_ss + C + c4 * 4);
    }
    float4 m = make_float4(ninf, ninf, ninf, ninf);
    uint32_t ix = 0, iy = 0, iz = 0, iw = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = oh * 2 - pt + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = ow * 2 - pl + kw;
        if (w < 0 || w >= W) continue;
        float4 v = *reinterpret_cast<const float4*>(x + (((long)b * H + h) * W + w) * C + c4 * 4);
        v.x = fmaf(v.x, xs.x, xh.x); v.y = fmaf(v.y, xs.y, xh.y);
        v.z = fmaf(v.z, xs.z, xh.z); v.w = fmaf(v.w, xs.w, xh.w);
        const uint32_t tap = kh * 3 + kw;
        if (v.x > m.x) { m.x = v.x; ix = tap; }
        if (v.y > m.y) { m.y = v.y; iy = tap; }
        if (v.z > m.z) { m.z = v.z; iz = tap; }
        if (v.w > m.w) { m.w = v.w; iw = tap; }
      }
    }
    if (residual) {
      float4 r = *reinterpret_cast<const float4*>(residual + i * 4);
      if (r_ss) {   // BatchNorm affine of the residual branch
        const float4 rs = *reinterpret_cast<const float4*>(r_ss + c4 * 4);
        const float4 rh = *reinterpret_cast<const float4*>(r_ss + C + c4 * 4);
        r.x = fmaf(r.x, rs.x, rh.x); r.y = fmaf(r.y, rs.y, rh.y);
        r.z = fmaf(r.z, rs.z, rh.z); r.w = fmaf(r.w, rs.w, rh.w);
      }
      m.x += r.x; m.y += r.y; m.z += r.z; m.w += r.w;
    }
    *reinterpret_cast<float4*>(y + i * 4) = m;
    if (idx4) idx4[i] = ix | (iy << 8) | (iz << 16) | (iw << 24);
  }
}

// dx[b,h,w,c] = sum over the (<=4) windows that contain (h,w) and whose argmax is (h,w) of dy.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy,
                                                          const uint32_t* __restrict__ idx4,
                                                          float* __restrict__ dx, int Bn, int H, int W,
                                                          int C, int OH, int OW, int pt, int pl) {
  const int c4n = C >> 2;
  const long total = (long)Bn * H * W * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    // windows: oh*2 - pt <= h <= oh*2 - pt + 2
    const int oh_lo = max(0, (h + pt - 2 + 1) >> 1), oh_hi = min(OH - 1, (h + pt) >> 1);
    const int ow_lo = max(0, (w + pl - 2 + 1) >> 1), ow_hi = min(OW - 1, (w + pl) >> 1);
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const uint32_t kh = h - (oh * 2 - pt);
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const uint32_t tap = kh * 3 + (w - (ow * 2 - pl));
        const long o = (((long)b * OH + oh) * OW + ow) * c4n + c4;
        const uint32_t id = idx4[o];
        const float4 g = *reinterpret_cast<const float4*>(dy + o * 4);
        if ((id & 0xffu) == tap) s.x += g.x;
        if (((id >> 8) & 0xffu) == tap) s.y += g.y;
        if (((id >> 16) & 0xffu) == tap) s.z += g.z;
        if ((id >> 24) == tap) s.w += g.w;
      }
    }
    *reinterpret_cast<float4*>(dx + i * 4) = s;
  }
}

// The same data gradient, plus the BatchNorm-backward sums of the layer whose (never materialised) output was pooled:
// dx IS dL/d(BN output), so sum dx and sum dx * xhat(yp) are accumulated while dx is produced -- one extra read of
// yp instead of a stand-alone reduction pass that reads dx AND yp again.  blockDim = (CL, 256/CL): x over channel
// quads, y over pixels; a thread keeps its channel quad, so its two float4 sums stay in registers.  One partial
// row [2][C] per grid row, fixed order (rows of the thread block in y order).
__global__ __launch_bounds__(256) void maxpool_bwd_bnsums_kernel(const float* __restrict__ dy,
                                                                 const uint32_t* __restrict__ idx4,
                                                                 float* __restrict__ dx, int Bn, int H, int W,
                                                                 int C, int OH, int OW, int pt, int pl,
                                                                 const float* __restrict__ yp,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) float4 pred4[];   // [blockDim.y][2][blockDim.x]
  const int c4n = C >> 2;
  const int c4 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = c4 < c4n;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (active) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + c4 * 4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + c4 * 4);
    const long npix = (long)Bn * H * W;
    // The windows that contain (h, w): rows oh1 = (h + pt) / 2 (tap row kh1 = (h + pt) % 2) and, when kh1 == 0, oh1 - 1
    // (tap row 2); columns likewise.  All four candidates are fetched at once -- clamped addresses, no branches -- and
    // added in maxpool_bwd_kernel's order (oh ascending, ow ascending; a candidate that does not exist adds 0): 0.317 ->
    // 0.261 ms per step over the four pooling layers (two pixels per trip measured slower: 0.280).
    struct Px {
      float4 v, g00, g01, g10, g11;
      uint32_t i00, i01, i10, i11, kh1, kw1;
      bool v00, v01, v10, v11;
    };
    auto fetch = [&](long r, Px& p) {
      const int w = (int)(r % W);
      const long t = r / W;
      const int h = (int)(t % H);
      const int b = (int)(t / H);
      p.v = *reinterpret_cast<const float4*>(yp + r * C + c4 * 4);
      const int hp = h + pt, wp = w + pl;
      const int oh1 = hp >> 1, ow1 = wp >> 1;
      p.kh1 = hp & 1;
      p.kw1 = wp & 1;
      const bool vh1 = oh1 < OH, vh0 = p.kh1 == 0 && oh1 >= 1, vw1 = ow1 < OW, vw0 = p.kw1 == 0 && ow1 >= 1;
      p.v00 = vh0 && vw0; p.v01 = vh0 && vw1; p.v10 = vh1 && vw0; p.v11 = vh1 && vw1;
      const int oh0 = max(oh1 - 1, 0), ow0 = max(ow1 - 1, 0), oh1c = min(oh1, OH - 1), ow1c = min(ow1, OW - 1);
      const long rb = (long)b * OH;
      const long o00 = ((rb + oh0) * OW + ow0) * c4n + c4, o01 = ((rb + oh0) * OW + ow1c) * c4n + c4;
      const long o10 = ((rb + oh1c) * OW + ow0) * c4n + c4, o11 = ((rb + oh1c) * OW + ow1c) * c4n + c4;
      p.i00 = idx4[o00]; p.i01 = idx4[o01]; p.i10 = idx4[o10]; p.i11 = idx4[o11];
      p.g00 = *reinterpret_cast<const float4*>(dy + o00 * 4); p.g01 = *reinterpret_cast<const float4*>(dy + o01 * 4);
      p.g10 = *reinterpret_cast<const float4*>(dy + o10 * 4); p.g11 = *reinterpret_cast<const float4*>(dy + o11 * 4);
    };
    auto sum = [&](long r, const Px& p) {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      auto take = [&](bool valid, uint32_t id, const float4 g, uint32_t tap) {
        s.x += (valid && (id & 0xffu) == tap) ? g.x : 0.f;
        s.y += (valid && ((id >> 8) & 0xffu) == tap) ? g.y : 0.f;
        s.z += (valid && ((id >> 16) & 0xffu) == tap) ? g.z : 0.f;
        s.w += (valid && (id >> 24) == tap) ? g.w : 0.f;
      };
      take(p.v00, p.i00, p.g00, 2 * 3 + 2);
      take(p.v01, p.i01, p.g01, 2 * 3 + p.kw1);
      take(p.v10, p.i10, p.g10, p.kh1 * 3 + 2);
      take(p.v11, p.i11, p.g11, p.kh1 * 3 + p.kw1);
      *reinterpret_cast<float4*>(dx + r * C + c4 * 4) = s;
      const float4 v = p.v;
      s0.x += s.x; s0.y += s.y; s0.z += s.z; s0.w += s.w;
      s1.x = fmaf(s.x, (v.x - mu.x) * is.x, s1.x); s1.y = fmaf(s.y, (v.y - mu.y) * is.y, s1.y);
      s1.z = fmaf(s.z, (v.z - mu.z) * is.z, s1.z); s1.w = fmaf(s.w, (v.w - mu.w) * is.w, s1.w);
    };
    for (long r = (long)blockIdx.y * blockDim.y + threadIdx.y; r < npix; r += (long)gridDim.y * blockDim.y) {
      Px p;
      fetch(r, p);
      sum(r, p);
    }
  }
  const int bx = blockDim.x, by = blockDim.y;
  pred4[(threadIdx.y * 2 + 0) * bx + threadIdx.x] = s0;
  pred4[(threadIdx.y * 2 + 1) * bx + threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.y == 0 && active) {
    for (int q = 0; q < 2; ++q) {
      float4 sm = pred4[q * bx + threadIdx.x];
      for (int y = 1; y < by; ++y) {
        const float4 u = pred4[(y * 2 + q) * bx + threadIdx.x];
        sm.x += u.x; sm.y += u.y; sm.z += u.z; sm.w += u.w;
      }
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.y * 2 + q) * C + c4 * 4) = sm;
    }
  }
}

// Launch shape of the fused kernel: 8 channel-quad lanes x 32 pixel rows per workgroup (a wave reads 8 rows of
// 128 contiguous bytes), grid (ceil(C/32), rows).  The number of partial rows doubles as the number of workgroup rows,
// so it decides the parallelism: at most `max_rows` (128 where the consumer is the one-launch BatchNorm backward, more
// where only per-channel coefficients are derived from the sums), at least four pixels per thread.
#define POOL_CL 8
extern "C" long spnet_maxpool3x3s2_bwd_rows(int B, int H, int W, int C, int max_rows) {
  const int by = 256 / POOL_CL;
  long gy = ((long)B * H * W + (long)by * 4 - 1) / ((long)by * 4);
  long cap = max_rows < 1 ? 1 : max_rows;
  if (gy > cap) gy = cap;
  return gy < 1 ? 1 : gy;
}

__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float* __restrict__ x,
                                                           float* __restrict__ y, int Bn, int H, int W,
                                                           int C, int OH, int OW) {
  const long total = (long)Bn * OH * OW * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    const float* p = x + (((long)b * H + 2 * oh) * W + 2 * ow) * C + c;
    y[i] = ((p[0] + p[C]) + (p[(long)W * C] + p[(long)W * C + C])) * 0.25f;
  }
}

__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ dy,
                                                           float* __restrict__ dx, int Bn, int H, int W,
                                                           int C, int OH, int OW) {
  const long total = (long)Bn * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    const int oh = h >> 1, ow = w >> 1;
    dx[i] = (oh < OH && ow < OW) ? 0.25f * dy[(((long)b * OH + oh) * OW + ow) * C + c] : 0.f;
  }
}

static void same_pool_geom(int in, int* out, int* pad_before) {
  *out = (in + 1) / 2;
  int total = (*out - 1) * 2 + 3 - in;
  if (total < 0) total = 0;
  *pad_before = total / 2;
}

// x_ss / r_ss (or NULL): [scale[C] | shift[C]] of the BatchNorm that produced x / residual, applied on
// load so that neither normalised tensor is ever written to HBM.
extern "C" int spnet_maxpool3x3s2_add_fwd(const float* x, const float* residual, float* y,
                                          uint32_t* idx4, int B, int H, int W, int C, const float* x_ss,
                                          const float* r_ss, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  int OH, OW, pt, pl;
  same_pool_geom(H, &OH, &pt);
  same_pool_geom(W, &OW, &pl);
  const long total = (long)B * OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool_add_fwd_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, residual, y, idx4, B, H, W, C, OH, OW, pt, pl, x_ss, r_ss);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_maxpool3x3s2_bwd(const float* dy, const uint32_t* idx4, float* dx, int B, int H,
                                      int W, int C, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  int OH, OW, pt, pl;
  same_pool_geom(H, &OH, &pt);
  same_pool_geom(W, &OW, &pl);
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, idx4, dx, B, H, W, C, OH, OW, pt, pl);
  SPNET_RETURN_LAUNCH_STATUS();
}

// spnet_maxpool3x3s2_bwd that also emits the backward sums (sum dx, sum dx * xhat) of the BatchNorm whose
// pre-normalisation tensor is yp [B,H,W,C] (mean / invstd saved by its forward): partial[rows][2][C],
// rows = spnet_maxpool3x3s2_bwd_rows(B,H,W,C,max_rows) (or fewer).  Replaces the TF MaxPoolGrad + the reduction half of the fused
// BatchNorm gradient behind block{2,3,4,13}_pool of keras.applications.Xception (call site spnet/models.py:357-359).
extern "C" int spnet_maxpool3x3s2_bwd_bnsums(const float* dy, const uint32_t* idx4, float* dx, int B, int H, int W,
                                             int C, const float* yp, const float* mean, const float* invstd,
                                             float* partial, int rows, void* stream) {
  if ((C & 3) || !yp || !mean || !invstd || !partial) return (int)hipErrorInvalidValue;
  int OH, OW, pt, pl;
  same_pool_geom(H, &OH, &pt);
  same_pool_geom(W, &OW, &pl);
  if (rows < 1 || rows > spnet_maxpool3x3s2_bwd_rows(B, H, W, C, rows)) return (int)hipErrorInvalidValue;
  const int cl = POOL_CL, by = 256 / cl;
  dim3 grid((C / 4 + cl - 1) / cl, (unsigned)rows), block(cl, by);
  hipLaunchKernelGGL(maxpool_bwd_bnsums_kernel, grid, block, (size_t)by * 2 * cl * sizeof(float4), (hipStream_t)stream,
                     dy, idx4, dx, B, H, W, C, OH, OW, pt, pl, yp, mean, invstd, partial);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_avgpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
  const int OH = H / 2, OW = W / 2;
  const long total = (long)B * OH * OW * C;
  hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, y, B, H, W, C, OH, OW);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
  const int OH = H / 2, OW = W / 2;
  const long total = (long)B * H * W * C;
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, dy, dx, B, H, W, C, OH, OW);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------- Inception-ResNet-v2 pooling layers
// MaxPooling2D(3, strides=2, padding='valid') (stem, mixed_6a, mixed_7a of keras InceptionResNetV2; call site
// spnet/models.py:357-359): the same kernels with no padding, no residual and OH = (H-3)/2+1.
extern "C" int spnet_maxpool3x3s2_valid_fwd(const float* x, float* y, uint32_t* idx4, int B, int H, int W, int C,
                                            void* stream) {
  if ((C & 3) || H < 3 || W < 3) return (int)hipErrorInvalidValue;
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  const long total = (long)B * OH * OW * (C / 4);
  hipLaunchKernelGGL(maxpool_add_fwd_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (const float*)nullptr, y, idx4, B, H, W, C, OH, OW, 0, 0, (const float*)nullptr, (const float*)nullptr);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_maxpool3x3s2_valid_bwd(const float* dy, const uint32_t* idx4, float* dx, int B, int H, int W, int C,
                                            void* stream) {
  if ((C & 3) || H < 3 || W < 3) return (int)hipErrorInvalidValue;
  const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, dy, idx4, dx,
                     B, H, W, C, OH, OW, 0, 0);
  SPNET_RETURN_LAUNCH_STATUS();
}

// AveragePooling2D(3, strides=1, padding='same') (mixed_5b): TensorFlow averages over the window entries that lie
// inside the image only.  bwd = 1: dx[p] = sum over the windows q containing p of dy[q] / count(q).
__global__ __launch_bounds__(256) void avgpool3x3s1_kernel(const float* __restrict__ in, float* __restrict__ out, int Bn,
                                                           int H, int W, int C, int bwd) {
  const int c4n = C >> 2;
  const long total = (long)Bn * H * W * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int n = 0;
#pragma unroll
    for (int dh = -1; dh <= 1; ++dh) {
      const int hh = h + dh;
      if (hh < 0 || hh >= H) continue;
#pragma unroll
      for (int dw = -1; dw <= 1; ++dw) {
        const int ww = w + dw;
        if (ww < 0 || ww >= W) continue;
        const float4 v = *reinterpret_cast<const float4*>(in + (((long)b * H + hh) * W + ww) * C + c4 * 4);
        float k = 1.f;
        if (bwd) {   // weight of window (hh, ww): 1 / number of its entries inside the image
          const int nh = min(hh + 1, H - 1) - max(hh - 1, 0) + 1, nw = min(ww + 1, W - 1) - max(ww - 1, 0) + 1;
          k = 1.f / (float)(nh * nw);
        }
        s.x = fmaf(v.x, k, s.x); s.y = fmaf(v.y, k, s.y); s.z = fmaf(v.z, k, s.z); s.w = fmaf(v.w, k, s.w);
        ++n;
      }
    }
    if (!bwd) {
      const float k = 1.f / (float)n;
      s.x *= k; s.y *= k; s.z *= k; s.w *= k;
    }
    *reinterpret_cast<float4*>(out + i * 4) = s;
  }
}

extern "C" int spnet_avgpool3x3s1_same(const float* in, float* out, int B, int H, int W, int C, int backward,
                                       void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(avgpool3x3s1_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H,
                     W, C, backward ? 1 : 0);
  SPNET_RETURN_LAUNCH_STATUS();
}
