// Train-time augmentation (AugmentOnTheFly, spnet/callbacks.py:272-341) and the offline warps
// (spnet/augmentation.py:82-239) on device-resident frames.
//
// Parameter-driven form: the HOST draws the random parameters in exactly the reference's RNG call
// order (cutout_inplace augmentation.py:117-134, salt_n_pepa_inplace :157-180, blur gate
// callbacks.py:306-309); the device applies them to a pristine copy of the frames.  This keeps the
// result bit-identical to the reference's numpy code while the per-pixel work runs at HBM speed and
// the reference's X_orig.copy() host-RAM doubling (callbacks.py:291) disappears.
//
// Frames are [N][H][W] fp32 (single channel, values in [-1,1]).
#include "common.h"

#define MAX_RECTS 6

// per-image min / max -> mm[n][2]; stage 1: MM_CHUNKS workgroups per image, stage 2: combine
#define MM_CHUNKS 16
__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, long hw,
                                                             float* __restrict__ part) {
  __shared__ float rlo[4], rhi[4];
  const float* p = x + (long)blockIdx.y * hw;
  float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const float v = p[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0) { rlo[threadIdx.x >> 6] = lo; rhi[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = part + ((long)blockIdx.y * MM_CHUNKS + blockIdx.x) * 2;
    o[0] = fminf(fminf(rlo[0], rlo[1]), fminf(rlo[2], rlo[3]));
    o[1] = fmaxf(fmaxf(rhi[0], rhi[1]), fmaxf(rhi[2], rhi[3]));
  }
}
__global__ __launch_bounds__(64) void minmax_combine_kernel(const float* __restrict__ part, int N,
                                                            float* __restrict__ mm) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
  for (int c = 0; c < MM_CHUNKS; ++c) {
    lo = fminf(lo, part[((long)n * MM_CHUNKS + c) * 2]);
    hi = fmaxf(hi, part[((long)n * MM_CHUNKS + c) * 2 + 1]);
  }
  mm[n * 2] = lo;
  mm[n * 2 + 1] = hi;
}

// dst[n] = src[src_index[n]] with the image's rectangles painted in order (later ones win).
// rects[n][MAX_RECTS][4] = (r0, r1, c0, c1) half-open; vals[n][MAX_RECTS]; nrect[n].
__global__ __launch_bounds__(256) void cutout_kernel(const float* __restrict__ src,
                                                     const int* __restrict__ src_index,
                                                     float* __restrict__ dst, int H, int W,
                                                     const int* __restrict__ rects,
                                                     const float* __restrict__ vals,
                                                     const int* __restrict__ nrect) {
  const int n = blockIdx.y;
  const long hw = (long)H * W;
  const float* s = src + (long)(src_index ? src_index[n] : n) * hw;
  float* d = dst + (long)n * hw;
  const int nr = nrect[n];
  int r0[MAX_RECTS], r1[MAX_RECTS], c0[MAX_RECTS], c1[MAX_RECTS];
  float val[MAX_RECTS];
#pragma unroll
  for (int k = 0; k < MAX_RECTS; ++k) {
    const int* q = rects + ((long)n * MAX_RECTS + k) * 4;
    r0[k] = q[0]; r1[k] = q[1]; c0[k] = q[2]; c1[k] = q[3];
    val[k] = vals[(long)n * MAX_RECTS + k];
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / W), c = (int)(i % W);
    float v = s[i];
#pragma unroll
    for (int k = 0; k < MAX_RECTS; ++k)
      if (k < nr && r >= r0[k] && r < r1[k] && c >= c0[k] && c < c1[k]) v = val[k];
    d[i] = v;
  }
}

// One workgroup per image: salt (= image max) first, then pepper (= image min) over it, as the
// reference's two fancy-index assignments do.  coords[n][2][npts] = rows then cols (salt points
// first, then pepper points); flag[n] = 0 skips the image.
__global__ __launch_bounds__(256) void saltpepper_kernel(float* __restrict__ x, int H, int W,
                                                         const int* __restrict__ coords, int n_salt,
                                                         int n_pepper, const int* __restrict__ flag,
                                                         const float* __restrict__ mm) {
  const int n = blockIdx.x;
  if (!flag[n]) return;
  float* d = x + (long)n * H * W;
  const int npts = n_salt + n_pepper;
  const int* rows = coords + (long)n * 2 * npts;
  const int* cols = rows + npts;
  const float lo = mm[n * 2 + 0], hi = mm[n * 2 + 1];
  for (int i = threadIdx.x; i < n_salt; i += blockDim.x) d[(long)rows[i] * W + cols[i]] = hi;
  __threadfence_block();
  __syncthreads();
  for (int i = n_salt + threadIdx.x; i < npts; i += blockDim.x) d[(long)rows[i] * W + cols[i]] = lo;
}

// Inverse-affine bilinear gather with zero border: covers flip, rotate-about-centre and integer
// translate (spnet/augmentation.py:82-112,184-207,216-239 through cv2.flip / cv2.warpAffine).
// minv[n][6] maps DESTINATION pixel (x,y) to SOURCE coordinates: sx = a*x + b*y + c, sy = d*x + e*y + f.
// Images are [N][H][W][C] fp32; weights are exact floats (OpenCV quantises them to 1/32 pixel).
__global__ __launch_bounds__(256) void warp_affine_kernel(const float* __restrict__ src,
                                                          float* __restrict__ dst, int H, int W, int C,
                                                          const float* __restrict__ minv) {
  const int n = blockIdx.y;
  const float* m = minv + (long)n * 6;
  const float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5];
  const long hw = (long)H * W;
  const float* s = src + (long)n * hw * C;
  float* o = dst + (long)n * hw * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i % W);
    const float sx = a * x + b * y + c, sy = d * x + e * y + f;
    const float fx = floorf(sx), fy = floorf(sy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float wx = sx - fx, wy = sy - fy;
    for (int ch = 0; ch < C; ++ch) {
      float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
      if (y0 >= 0 && y0 < H) {
        if (x0 >= 0 && x0 < W) v00 = s[((long)y0 * W + x0) * C + ch];
        if (x0 + 1 >= 0 && x0 + 1 < W) v01 = s[((long)y0 * W + x0 + 1) * C + ch];
      }
      if (y0 + 1 >= 0 && y0 + 1 < H) {
        if (x0 >= 0 && x0 < W) v10 = s[((long)(y0 + 1) * W + x0) * C + ch];
        if (x0 + 1 >= 0 && x0 + 1 < W) v11 = s[((long)(y0 + 1) * W + x0 + 1) * C + ch];
      }
      const float top = v00 + wx * (v01 - v00), bot = v10 + wx * (v11 - v10);
      o[i * C + ch] = top + wy * (bot - top);
    }
  }
}

// Dropout(0.1) of the stem (spnet/models.py:340): y = x * keep/(1-rate) with a counter-based hash RNG
// so that the mask is a pure function of (seed, element index) and can be regenerated in backward.
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x,
                                                      float* __restrict__ y, long n, uint32_t seed,
                                                      uint32_t thresh, float scale,
                                                      const uint32_t* __restrict__ seed_dev) {
  if (seed_dev) seed = seed_dev[0];   // per-step seed read from device memory (hipGraph replay)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const uint32_t h = hash_u32((uint32_t)i * 0x9e3779b9U + seed);
    y[i] = (h >= thresh) ? x[i] * scale : 0.f;
  }
}

// scratch: N * 32 floats
extern "C" int spnet_minmax(const float* x, int N, long hw, float* mm, float* scratch, void* stream) {
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(MM_CHUNKS, N), dim3(256), 0, (hipStream_t)stream, x, hw, scratch);
  hipLaunchKernelGGL(minmax_combine_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, scratch, N, mm);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_cutout(const float* src, const int* src_index, float* dst, int N, int H, int W,
                            const int* rects, const float* vals, const int* nrect, void* stream) {
  const long hw = (long)H * W;
  int gx = (int)((hw + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(cutout_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, src, src_index, dst,
                     H, W, rects, vals, nrect);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_saltpepper(float* x, int N, int H, int W, const int* coords, int n_salt,
                                int n_pepper, const int* flag, const float* mm, void* stream) {
  hipLaunchKernelGGL(saltpepper_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, x, H, W, coords,
                     n_salt, n_pepper, flag, mm);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_warp_affine(const float* src, float* dst, int N, int H, int W, int C,
                                 const float* minv, void* stream) {
  const long hw = (long)H * W;
  int gx = (int)((hw + 255) / 256);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(warp_affine_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, src, dst, H, W,
                     C, minv);
  SPNET_RETURN_LAUNCH_STATUS();
}

// rate in [0,1): elements with hash < rate*2^32 are dropped, survivors scaled by 1/(1-rate).
extern "C" int spnet_dropout(const float* x, float* y, long n, unsigned seed, float rate,
                             const unsigned* seed_dev, void* stream) {
  const double t = (double)rate * 4294967296.0;
  const uint32_t thresh = (uint32_t)(t > 4294967295.0 ? 4294967295.0 : t);
  hipLaunchKernelGGL(dropout_kernel, dim3(spnet_ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     y, n, seed, thresh, 1.0f / (1.0f - rate), seed_dev);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------- Gaussian blur (the reference's intended augmentation)
// blur_inplace (spnet/augmentation.py:66-70) calls cv2.GaussianBlur(img, (k,k), 0) with k in {3,7} and DISCARDS the
// result, so the reference's blur is a no-op (the default here too); this kernel is what that call computes, for runs
// that want the augmentation the author meant (DeviceAugmenter(real_blur=True)).  OpenCV semantics: sigma = 0 with
// k <= 7 selects the fixed binomial-like kernels below (getGaussianKernel's small_gaussian_tab), borders are
// BORDER_REFLECT_101, arithmetic in float.  ksize[n] in {0 (copy), 3, 5, 7} per frame.
__constant__ float kGauss[4][7] = {{1.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
                                   {0.25f, 0.5f, 0.25f, 0.f, 0.f, 0.f, 0.f},
                                   {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f, 0.f, 0.f},
                                   {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};

__device__ __forceinline__ int reflect101(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return min(max(i, 0), n - 1);
}

__global__ __launch_bounds__(256) void gaussian_blur_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            int H, int W, const int* __restrict__ ksize) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, n = blockIdx.z;
  if (x >= W) return;
  const long base = (long)n * H * W;
  const int k = ksize[n];
  if (k <= 1) {
    dst[base + (long)y * W + x] = src[base + (long)y * W + x];
    return;
  }
  const int r = k >> 1;
  const float* kw = kGauss[r];
  float acc = 0.f;
  for (int i = -r; i <= r; ++i) {        // columns first within a row (cv2: row filter, then column filter)
    const float* row = src + base + (long)reflect101(y + i, H) * W;
    float s = 0.f;
    for (int j = -r; j <= r; ++j) s = fmaf(kw[j + r], row[reflect101(x + j, W)], s);
    acc = fmaf(kw[i + r], s, acc);
  }
  dst[base + (long)y * W + x] = acc;
}

extern "C" int spnet_gaussian_blur(const float* src, float* dst, int N, int H, int W, const int* ksize, void* stream) {
  if (N < 1 || H < 4 || W < 4 || !src || !dst || src == dst || !ksize) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gaussian_blur_kernel, dim3((W + 255) / 256, H, N), dim3(256), 0, (hipStream_t)stream, src, dst, H, W,
                     ksize);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------- cv2.warpAffine, fixed-point form (8-bit images)
// What OpenCV 3.4 computes for cv2.warpAffine(uint8 image, M, (w,h)) with the defaults the reference uses
// (INTER_LINEAR, BORDER_CONSTANT 0; spnet/augmentation.py:193-194, 232): source coordinates in 1/32-pixel fixed
// point, X = (X0[y] + adelta[x]) >> 5 with X0 / adelta the host-rounded (cvRound, 10 fractional bits) row and column
// terms of the inverted matrix, bilinear weights (32-fy)(32-fx)*32 of 2^15 and a rounded 15-bit shift.  Integer
// arithmetic end to end: the result is exactly the integer OpenCV's algorithm yields, not a float approximation of it.
// src / dst hold 8-bit values as fp32 [N][H][W][C]; xrow [N][H][2] = (X0, Y0), xcol [N][W][2] = (adelta, bdelta).
__global__ __launch_bounds__(256) void warp_affine_fixed_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                int H, int W, int C, const int* __restrict__ xrow,
                                                                const int* __restrict__ xcol) {
  const int n = blockIdx.y;
  const long hw = (long)H * W;
  const float* s = src + (long)n * hw * C;
  float* o = dst + (long)n * hw * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / W), x = (int)(i % W);
    const int X = (xrow[((long)n * H + y) * 2] + xcol[((long)n * W + x) * 2]) >> 5;
    const int Y = (xrow[((long)n * H + y) * 2 + 1] + xcol[((long)n * W + x) * 2 + 1]) >> 5;
    const int sx = X >> 5, sy = Y >> 5, fx = X & 31, fy = Y & 31;
    const int w00 = min((32 - fy) * (32 - fx) * 32, 32767), w01 = (32 - fy) * fx * 32;   // int16 table entries
    const int w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
    for (int ch = 0; ch < C; ++ch) {
      const int p00 = (y0 && x0) ? (int)s[((long)sy * W + sx) * C + ch] : 0;
      const int p01 = (y0 && x1) ? (int)s[((long)sy * W + sx + 1) * C + ch] : 0;
      const int p10 = (y1 && x0) ? (int)s[((long)(sy + 1) * W + sx) * C + ch] : 0;
      const int p11 = (y1 && x1) ? (int)s[((long)(sy + 1) * W + sx + 1) * C + ch] : 0;
      const int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15;
      o[i * C + ch] = (float)min(max(v, 0), 255);
    }
  }
}

extern "C" int spnet_warp_affine_fixed(const float* src, float* dst, int N, int H, int W, int C, const int* xrow,
                                       const int* xcol, void* stream) {
  if (N < 1 || !src || !dst || src == dst || !xrow || !xcol) return (int)hipErrorInvalidValue;
  const long hw = (long)H * W;
  int gx = (int)((hw + 255) / 256);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(warp_affine_fixed_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, C, xrow,
                     xcol);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Input codec on the device: uint8 grey levels -> the network's float32 input, x = (v / 255 - 0.5) * 2 with every step
// rounded as numpy rounds it in load_X_one_proc (spnet/utils.py:340-342: `img /= 255.0; img -= 0.5; img *= 2.0` on
// float32), so a frame converted here is bit-identical to one converted on the host.  Frames then cross PCIe as one
// byte per pixel instead of four; the pass replaces the device-to-device copy into the plan's input buffer.
// 16 pixels per thread step: one 16-byte load, four 16-byte stores (+ a scalar tail); both pointers 16-byte aligned.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float u8_to_input(unsigned v) {
  return __fmul_rn(__fsub_rn(__fdiv_rn((float)v, 255.f), 0.5f), 2.f);
}

__global__ __launch_bounds__(256) void u8_to_input_kernel(const unsigned char* __restrict__ src8, float* __restrict__ dstf,
                                                          long n) {
  const uint4* __restrict__ src = reinterpret_cast<const uint4*>(src8);
  float4* __restrict__ dst = reinterpret_cast<float4*>(dstf);
  const long n16 = n / 16;
  const long gtid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (long i = gtid; i < n16; i += (long)gridDim.x * blockDim.x) {
    const uint4 q = src[i];
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      dst[i * 4 + k] = make_float4(u8_to_input(w[k] & 0xffu), u8_to_input((w[k] >> 8) & 0xffu),
                                   u8_to_input((w[k] >> 16) & 0xffu), u8_to_input(w[k] >> 24));
  }
  const long t = n16 * 16 + gtid;                    // the < 16 pixels past the last whole vector
  if (t < n) dstf[t] = u8_to_input(src8[t]);
}

extern "C" int spnet_u8_to_input(const unsigned char* src, float* dst, long n, void* stream) {
  if (n < 0 || !src || !dst || (((uintptr_t)src | (uintptr_t)dst) & 15)) return (int)hipErrorInvalidValue;
  if (n == 0) return 0;
  hipLaunchKernelGGL(u8_to_input_kernel, dim3(spnet_ew_grid((n + 15) / 16, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ------------------------------------------------------------------------------------------------
// Batch assembly: dst[i][:] = src[index[i]][:] for n rows of L floats -- the minibatch of frames / targets picked out of
// the resident training set by the epoch's shuffled index (Keras' fit does this on the host: train_spnet.py:75,81).
// index: int32 or int64 (idx_bytes 4 | 8) on the device.  Pure copy: 16 bytes per lane where rows are 16-byte multiples
// (L % 4 == 0), 4 bytes per lane otherwise (the reference layout's 331 x 331 frames: 109,561 floats per row).
// ------------------------------------------------------------------------------------------------
template <class IT, class VT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const VT* __restrict__ src, const IT* __restrict__ index,
                                                          VT* __restrict__ dst, int n, long lv, long src_rows) {
  const long total = (long)n * lv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / lv, c = i - r * lv;
    long s = (long)index[r];
    s = s < 0 ? 0 : (s >= src_rows ? src_rows - 1 : s);        // (an out-of-range index must not fault the device)
    dst[i] = src[s * lv + c];
  }
}

extern "C" int spnet_gather_rows(const float* src, long src_rows, const void* index, int idx_bytes, float* dst, int n, long L,
                                 void* stream) {
  if (!src || !index || !dst || n < 0 || L < 1 || src_rows < 1 || (idx_bytes != 4 && idx_bytes != 8) ||
      (((uintptr_t)src | (uintptr_t)dst) & 3))
    return (int)hipErrorInvalidValue;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = (L & 3) == 0 && ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0);
  const long lv = vec ? L / 4 : L;
  const dim3 grid(spnet_ew_grid((long)n * lv, 256));
#define GR(IT, VT) hipLaunchKernelGGL((gather_rows_kernel<IT, VT>), grid, dim3(256), 0, st, (const VT*)src, (const IT*)index, \
                                      (VT*)dst, n, lv, src_rows)
  if (vec) { if (idx_bytes == 4) GR(int, float4); else GR(long long, float4); }
  else { if (idx_bytes == 4) GR(int, float); else GR(long long, float); }
#undef GR
  SPNET_RETURN_LAUNCH_STATUS();
}
