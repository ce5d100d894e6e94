// General k_h x k_w convolutions of keras InceptionResNetV2 (3x3, 5x5, 1x7, 7x1, 1x3, 3x1; strides 1 | 2; 'same' |
// 'valid'; call site spnet/models.py:357-359 with cf.basemodel = 'InceptionResNetV2') as patch gather + the fp32 MFMA
// GEMM of gemm.hip, and the block glue of inception_resnet_block: x + scale*up (+ ReLU) and gradient accumulation.
// All tensors NHWC fp32 with C % 4 == 0 (the one 3-channel conv of the network is block1_conv1's shape and runs the
// direct kernel of stem.hip); K index of a patch row = (kh*KW + kw)*C + c, i.e. the flattened HWIO kernel is the GEMM's
// B operand as it stands.
#include "common.h"

// col[(b,oh,ow)][(kh,kw,c)] = x[b, oh*s - pt + kh, ow*s - pl + kw, c]  (zero outside the image)
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ x, float* __restrict__ col, int Bn, int H,
                                                     int W, int C, int KH, int KW, int s, int pt, int pl, int OH, int OW,
                                                     long ldx) {
  const int c4n = C >> 2;
  const long total = (long)Bn * OH * OW * KH * KW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int kw = (int)(t % KW);
    t /= KW;
    const int kh = (int)(t % KH);
    t /= KH;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    const int h = oh * s - pt + kh, w = ow * s - pl + kw;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h >= 0 && h < H && w >= 0 && w < W) v = *reinterpret_cast<const float4*>(x + (((long)b * H + h) * W + w) * ldx + c4 * 4);
    *reinterpret_cast<float4*>(col + i * 4) = v;
  }
}

// dx[b,h,w,c] = sum over the patches (oh,ow) and taps (kh,kw) that read (h,w) of dcol  (gather form: no atomics)
__global__ __launch_bounds__(256) void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int Bn, int H,
                                                     int W, int C, int KH, int KW, int s, int pt, int pl, int OH, int OW) {
  const int c4n = C >> 2;
  const long total = (long)Bn * H * W * c4n;
  const long K = (long)KH * KW * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < KH; ++kh) {
      const int hh = h + pt - kh;
      if (hh < 0 || hh % s) continue;
      const int oh = hh / s;
      if (oh >= OH) continue;
      for (int kw = 0; kw < KW; ++kw) {
        const int ww = w + pl - kw;
        if (ww < 0 || ww % s) continue;
        const int ow = ww / s;
        if (ow >= OW) continue;
        const float4 v = *reinterpret_cast<const float4*>(dcol + (((long)b * OH + oh) * OW + ow) * K + ((long)kh * KW + kw) * C + c4 * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(dx + i * 4) = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Gradient producers that also leave the BatchNorm-backward sums of the layer whose output they differentiate.
// A conv2d_bn output y = relu(BN(yp)) of Inception-ResNet-v2 has exactly one consumer: a k x k convolution (its gradient
// comes out of the adjoint patch gather) or a Concatenate (a column block of the concatenated gradient).  Both passes
// produce dL/dy element by element; these kernels mask it with the ReLU decision (y > 0) and accumulate, per channel,
// sum g and sum g * xhat(yp) on the way -- the reduction pass of the BatchNorm backward (read yp and g again, one more
// dependent launch on the main stream) disappears, and the BatchNorm backward runs from partial rows with no
// activation left to handle.  blockDim = (8 channel quads, 32 rows); one partial row [2][C] per grid row, fixed order.
//   MODE 0: g = col2im(dcol)      MODE 1: g = src[r * lds + c] (strided column block)
// ------------------------------------------------------------------------------------------------
// KHT x KWT > 0 (MODE 0, stride 1 only): the taps unrolled and branch-free -- an out-of-range tap loads a clamped address
// and contributes +0 -- so that a thread's KHT*KWT loads are all in flight together; with run-time loops and `continue`
// every tap waits for the one before it, and these launches (a few waves per CU) are pure load latency: 9,744 pixels x
// 32 channels x 9 taps took 33 us.  The sum keeps the tap order (a skipped tap and an added zero are the same value).
#define GS_CL 8
template <int MODE, int KHT = 0, int KWT = 0>
__global__ __launch_bounds__(256) void grad_bnsums_kernel(const float* __restrict__ src, int lds, float* __restrict__ dx,
                                                          long lddx, int Bn, int H, int W, int C, int KH, int KW, int s,
                                                          int pt, int pl, int OH, int OW, const float* __restrict__ y,
                                                          long ldy, const float* __restrict__ yp, long ldyp,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, int relu,
                                                          float* __restrict__ partial, long ldp) {
  __shared__ __attribute__((aligned(16))) float4 red4[32 * 2 * GS_CL];
  const int c4n = C >> 2;
  const int c4 = blockIdx.x * GS_CL + threadIdx.x;
  const bool active = c4 < c4n;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  if (active) {
    const float4 mu = *reinterpret_cast<const float4*>(mean + c4 * 4);
    const float4 is = *reinterpret_cast<const float4*>(invstd + c4 * 4);
    const long npix = (long)Bn * H * W;
    const long K = (long)KH * KW * C;
    for (long r = (long)blockIdx.y * 32 + threadIdx.y; r < npix; r += (long)gridDim.y * 32) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
      if (MODE == 1) {
        g = *reinterpret_cast<const float4*>(src + r * lds + c4 * 4);
      } else if constexpr (KHT > 0) {
        const int w = (int)(r % W);
        const int t = (int)(r / W);
        const int h = t % H, b = t / H;
        float4 v[KHT * KWT > 0 ? KHT * KWT : 1];
#pragma unroll
        for (int kh = 0; kh < KHT; ++kh) {
          const int oh = h + pt - kh;
          const int ohc = min(max(oh, 0), OH - 1);
#pragma unroll
          for (int kw = 0; kw < KWT; ++kw) {
            const int ow = w + pl - kw;
            const int owc = min(max(ow, 0), OW - 1);
            v[kh * KWT + kw] = *reinterpret_cast<const float4*>(src + (((long)b * OH + ohc) * OW + owc) * K +
                                                                (kh * KWT + kw) * C + c4 * 4);
          }
        }
#pragma unroll
        for (int kh = 0; kh < KHT; ++kh) {
          const bool vh = (unsigned)(h + pt - kh) < (unsigned)OH;
#pragma unroll
          for (int kw = 0; kw < KWT; ++kw) {
            const bool ok = vh && (unsigned)(w + pl - kw) < (unsigned)OW;
            const float4 u = v[kh * KWT + kw];
            g.x += ok ? u.x : 0.f; g.y += ok ? u.y : 0.f; g.z += ok ? u.z : 0.f; g.w += ok ? u.w : 0.f;
          }
        }
      } else {
        const int w = (int)(r % W);
        long t = r / W;
        const int h = (int)(t % H);
        const int b = (int)(t / H);
        for (int kh = 0; kh < KH; ++kh) {
          const int hh = h + pt - kh;
          if (hh < 0 || hh % s) continue;
          const int oh = hh / s;
          if (oh >= OH) continue;
          for (int kw = 0; kw < KW; ++kw) {
            const int ww = w + pl - kw;
            if (ww < 0 || ww % s) continue;
            const int ow = ww / s;
            if (ow >= OW) continue;
            const float4 v = *reinterpret_cast<const float4*>(src + (((long)b * OH + oh) * OW + ow) * K +
                                                              ((long)kh * KW + kw) * C + c4 * 4);
            g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
          }
        }
      }
      const float4 xv = *reinterpret_cast<const float4*>(yp + r * ldyp + c4 * 4);
      if (relu) {
        const float4 yv = *reinterpret_cast<const float4*>(y + r * ldy + c4 * 4);
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
        g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      }
      *reinterpret_cast<float4*>(dx + r * lddx + c4 * 4) = g;
      s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
      s1.x = fmaf(g.x, (xv.x - mu.x) * is.x, s1.x); s1.y = fmaf(g.y, (xv.y - mu.y) * is.y, s1.y);
      s1.z = fmaf(g.z, (xv.z - mu.z) * is.z, s1.z); s1.w = fmaf(g.w, (xv.w - mu.w) * is.w, s1.w);
    }
  }
  red4[(threadIdx.y * 2 + 0) * GS_CL + threadIdx.x] = s0;
  red4[(threadIdx.y * 2 + 1) * GS_CL + threadIdx.x] = s1;
  __syncthreads();
  if (threadIdx.y == 0 && active) {
    for (int q = 0; q < 2; ++q) {
      float4 sm = red4[q * GS_CL + threadIdx.x];
      for (int yy = 1; yy < 32; ++yy) {
        const float4 u = red4[(yy * 2 + q) * GS_CL + threadIdx.x];
        sm.x += u.x; sm.y += u.y; sm.z += u.z; sm.w += u.w;
      }
      *reinterpret_cast<float4*>(partial + ((long)blockIdx.y * 2 + q) * ldp + c4 * 4) = sm;
    }
  }
}

// partial rows (= workgroup rows) for npix pixels: at most max_rows, one pixel per thread or more (these tensors are
// small: the rows are the parallelism)
extern "C" long spnet_grad_bnsums_rows(long npix, int max_rows) {
  long gy = (npix + 31) / 32;
  long cap = max_rows < 1 ? 1 : max_rows;
  if (gy > cap) gy = cap;
  return gy < 1 ? 1 : gy;
}

static void conv_geom(int in, int k, int s, int same, int* out, int* before) {
  if (same) {
    *out = (in + s - 1) / s;
    int total = (*out - 1) * s + k - in;
    if (total < 0) total = 0;
    *before = total / 2;
  } else {
    *out = (in - k) / s + 1;
    *before = 0;
  }
}

// backward == 0: col [B*OH*OW][KH*KW*C] from x;  backward == 1: dx [B][H][W][C] from dcol (x = dcol, out = dx).
extern "C" int spnet_patches(const float* in, float* out, int B, int H, int W, int C, int KH, int KW, int stride, int same,
                             int backward, void* stream) {
  if ((C & 3) || KH < 1 || KW < 1 || (stride != 1 && stride != 2)) return (int)hipErrorInvalidValue;
  int OH, OW, pt, pl;
  conv_geom(H, KH, stride, same, &OH, &pt);
  conv_geom(W, KW, stride, same, &OW, &pl);
  if (OH < 1 || OW < 1) return (int)hipErrorInvalidValue;
  if (!backward) {
    const long total = (long)B * OH * OW * KH * KW * (C / 4);
    hipLaunchKernelGGL(im2col_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W,
                       C, KH, KW, stride, pt, pl, OH, OW, (long)C);
  } else {
    const long total = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(col2im_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W,
                       C, KH, KW, stride, pt, pl, OH, OW);
  }
  SPNET_RETURN_LAUNCH_STATUS();
}

// spnet_patches(backward = 1) that also masks the gradient with the ReLU of the layer that produced x (y = its output,
// relu != 0) and leaves that layer's BatchNorm-backward sums: partial[rows][2][C], rows <= spnet_grad_bnsums_rows(B*H*W, .).
// The `_ld` forms take row strides for every per-pixel tensor: dx (lddx), yp (ldyp) and the partial rows (ldp floats per
// [sum | sum*xhat] half-row): the differentiated layer may be ONE MEMBER of a group of sibling convolutions that share a
// concatenated pre-normalisation tensor, gradient buffer and partial-sum rows (column blocks of [.][Ct] tensors).
extern "C" int spnet_patches_bwd_bnsums_ld(const float* dcol, float* dx, long lddx, int B, int H, int W, int C, int KH, int KW,
                                           int stride, int same, const float* y, long ldy, const float* yp, long ldyp,
                                           const float* mean, const float* invstd, int relu, float* partial, long ldp,
                                           int rows, void* stream) {
  if ((C & 3) || KH < 1 || KW < 1 || (stride != 1 && stride != 2) || !yp || !mean || !invstd || !partial || (relu && !y))
    return (int)hipErrorInvalidValue;
  if (lddx < C || ldy < C || ldyp < C || ldp < C || ((lddx | ldy | ldyp | ldp) & 3)) return (int)hipErrorInvalidValue;
  int OH, OW, pt, pl;
  conv_geom(H, KH, stride, same, &OH, &pt);
  conv_geom(W, KW, stride, same, &OW, &pl);
  if (OH < 1 || OW < 1 || rows < 1 || rows > spnet_grad_bnsums_rows((long)B * H * W, rows)) return (int)hipErrorInvalidValue;
  dim3 grid((C / 4 + GS_CL - 1) / GS_CL, (unsigned)rows), block(GS_CL, 32);
#define GS_ARGS dcol, 0, dx, lddx, B, H, W, C, KH, KW, stride, pt, pl, OH, OW, y, ldy, yp, ldyp, mean, invstd, relu ? 1 : 0, partial, ldp
#define GS_CASE(KH_, KW_)                                                                                          \
  if (stride == 1 && KH == KH_ && KW == KW_) {                                                                     \
    hipLaunchKernelGGL((grad_bnsums_kernel<0, KH_, KW_>), grid, block, 0, (hipStream_t)stream, GS_ARGS);           \
    SPNET_RETURN_LAUNCH_STATUS();                                                                                  \
  }
  GS_CASE(3, 3) GS_CASE(1, 7) GS_CASE(7, 1) GS_CASE(1, 3) GS_CASE(3, 1) GS_CASE(5, 5)
  hipLaunchKernelGGL((grad_bnsums_kernel<0>), grid, block, 0, (hipStream_t)stream, GS_ARGS);
#undef GS_CASE
#undef GS_ARGS
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_patches_bwd_bnsums(const float* dcol, float* dx, int B, int H, int W, int C, int KH, int KW, int stride,
                                        int same, const float* y, const float* yp, const float* mean, const float* invstd,
                                        int relu, float* partial, int rows, void* stream) {
  return spnet_patches_bwd_bnsums_ld(dcol, dx, C, B, H, W, C, KH, KW, stride, same, y, C, yp, C, mean, invstd, relu, partial,
                                     C, rows, stream);
}

// dst[r][c] = src[r * lds + c] (* ReLU mask), c < C, dense dst -- the Concatenate backward of one branch -- with the same
// BatchNorm-backward sums of the branch's last layer.
extern "C" int spnet_copy_cols_bnsums_ld(const float* src, int lds, float* dst, long ldd, long M, int C, const float* y,
                                         long ldy, const float* yp, long ldyp, const float* mean, const float* invstd,
                                         int relu, float* partial, long ldp, int rows, void* stream) {
  if ((C & 3) || (lds & 3) || M < 1 || !yp || !mean || !invstd || !partial || (relu && !y)) return (int)hipErrorInvalidValue;
  if (ldd < C || ldy < C || ldyp < C || ldp < C || ((ldd | ldy | ldyp | ldp) & 3)) return (int)hipErrorInvalidValue;
  if (rows < 1 || rows > spnet_grad_bnsums_rows(M, rows)) return (int)hipErrorInvalidValue;
  dim3 grid((C / 4 + GS_CL - 1) / GS_CL, (unsigned)rows), block(GS_CL, 32);
  hipLaunchKernelGGL((grad_bnsums_kernel<1>), grid, block, 0, (hipStream_t)stream, src, lds, dst, ldd, 1, 1, (int)M, C, 1, 1, 1,
                     0, 0, 1, (int)M, y, ldy, yp, ldyp, mean, invstd, relu ? 1 : 0, partial, ldp);
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_copy_cols_bnsums(const float* src, int lds, float* dst, long M, int C, const float* y, long ldy,
                                      const float* yp, const float* mean, const float* invstd, int relu, float* partial,
                                      int rows, void* stream) {
  return spnet_copy_cols_bnsums_ld(src, lds, dst, C, M, C, y, ldy, yp, C, mean, invstd, relu, partial, C, rows, stream);
}

// Many strided column-block copies in one launch: job j copies rows_j x cols_j floats, dst_j[r * ldd_j + c] =
// src_j[r * lds_j + c].  jobs (device memory) = njobs x {src pointer, dst pointer, rows, cols, lds, ldd} as six 64-bit
// words; cols, lds, ldd multiples of 4.  (The kernels of sibling convolutions gathered into their concatenated GEMM
// operand after every optimizer step: one launch for the whole network.)
__global__ __launch_bounds__(256) void copy_cols_batched_kernel(const long long* __restrict__ jobs) {
  const long long* jb = jobs + 6 * blockIdx.y;
  const float* __restrict__ src = reinterpret_cast<const float*>(jb[0]);
  float* __restrict__ dst = reinterpret_cast<float*>(jb[1]);
  const long rows = jb[2];
  const int c4n = (int)(jb[3] >> 2);
  const long lds = jb[4], ldd = jb[5];
  const long total = rows * c4n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / c4n;
    const int c = (int)(i % c4n) * 4;
    *reinterpret_cast<float4*>(dst + r * ldd + c) = *reinterpret_cast<const float4*>(src + r * lds + c);
  }
}

extern "C" int spnet_copy_cols_batched(const void* jobs, int njobs, long max_elems, void* stream) {
  if (!jobs || njobs < 1 || max_elems < 4) return (int)hipErrorInvalidValue;
  long gx = (max_elems / 4 + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(copy_cols_batched_kernel, dim3((unsigned)gx, njobs), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs));
  SPNET_RETURN_LAUNCH_STATUS();
}

// The patch gather of an input whose pixels are ldx floats apart (ldx >= C): x is a column block of a wider tensor (the
// output of one member of a sibling group, _IRGroup).
extern "C" int spnet_patches_ld(const float* in, long ldx, float* out, int B, int H, int W, int C, int KH, int KW, int stride,
                                int same, void* stream) {
  if ((C & 3) || (ldx & 3) || ldx < C || KH < 1 || KW < 1 || (stride != 1 && stride != 2)) return (int)hipErrorInvalidValue;
  int OH, OW, pt, pl;
  conv_geom(H, KH, stride, same, &OH, &pt);
  conv_geom(W, KW, stride, same, &OW, &pl);
  if (OH < 1 || OW < 1) return (int)hipErrorInvalidValue;
  const long total = (long)B * OH * OW * KH * KW * (C / 4);
  hipLaunchKernelGGL(im2col_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, B, H, W, C,
                     KH, KW, stride, pt, pl, OH, OW, ldx);
  SPNET_RETURN_LAUNCH_STATUS();
}

// inception_resnet_block's Lambda + Activation: y = x + scale*up, ReLU if relu.
// backward (y = the forward OUTPUT, g = dL/dy): gm = g * (y > 0 if relu); dup = scale * gm; dx = gm (written only when
// dx != g: with dx == g and no ReLU the incoming buffer already is the x-branch gradient).
__global__ __launch_bounds__(256) void resadd_kernel(const float* __restrict__ x, const float* __restrict__ up,
                                                     float* __restrict__ y, long n4, float scale, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 a = *reinterpret_cast<const float4*>(x + i * 4), u = *reinterpret_cast<const float4*>(up + i * 4);
    float4 o = make_float4(fmaf(scale, u.x, a.x), fmaf(scale, u.y, a.y), fmaf(scale, u.z, a.z), fmaf(scale, u.w, a.w));
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *reinterpret_cast<float4*>(y + i * 4) = o;
  }
}
__global__ __launch_bounds__(256) void resadd_bwd_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                         float* __restrict__ dx, float* __restrict__ dup, long n4,
                                                         float scale, int relu) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = *reinterpret_cast<const float4*>(g + i * 4);
    if (relu) {
      const float4 o = *reinterpret_cast<const float4*>(y + i * 4);
      v.x = o.x > 0.f ? v.x : 0.f; v.y = o.y > 0.f ? v.y : 0.f; v.z = o.z > 0.f ? v.z : 0.f; v.w = o.w > 0.f ? v.w : 0.f;
    }
    *reinterpret_cast<float4*>(dx + i * 4) = v;
    *reinterpret_cast<float4*>(dup + i * 4) = make_float4(scale * v.x, scale * v.y, scale * v.z, scale * v.w);
  }
}

extern "C" int spnet_resadd(const float* x, const float* up, float* y, long n, float scale, int relu, void* stream) {
  if (n & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(resadd_kernel, dim3(spnet_ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, up, y, n / 4, scale,
                     relu);
  SPNET_RETURN_LAUNCH_STATUS();
}
extern "C" int spnet_resadd_bwd(const float* y, const float* g, float* dx, float* dup, long n, float scale, int relu,
                                void* stream) {
  if (n & 3) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(resadd_bwd_kernel, dim3(spnet_ew_grid(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, y, g, dx, dup,
                     n / 4, scale, relu);
  SPNET_RETURN_LAUNCH_STATUS();
}

// Gradient accumulation where a tensor feeds several branches, and channel-block copies for Concatenate:
// dst[r*ldd + c] (+)= src[r*lds + c], c < cols (cols, ldd, lds multiples of 4).
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst,
                                                        int ldd, long rows, int cols, int accumulate) {
  const int c4n = cols >> 2;
  const long total = rows * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / c4n;
    const int c = (int)(i % c4n) * 4;
    float4 v = *reinterpret_cast<const float4*>(src + r * lds + c);
    float4* d = reinterpret_cast<float4*>(dst + r * ldd + c);
    if (accumulate) {
      const float4 o = *d;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *d = v;
  }
}

extern "C" int spnet_copy_cols(const float* src, int lds, float* dst, int ldd, long rows, int cols, int accumulate,
                               void* stream) {
  if ((cols & 3) || (lds & 3) || (ldd & 3) || rows < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(copy_cols_kernel, dim3(spnet_ew_grid(rows * (cols / 4), 256)), dim3(256), 0, (hipStream_t)stream, src,
                     lds, dst, ldd, rows, cols, accumulate ? 1 : 0);
  SPNET_RETURN_LAUNCH_STATUS();
}
