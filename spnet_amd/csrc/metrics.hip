// Raster IoU of predicted vs true ellipses for the mAP metric (spnet/diagnostics.py:64-161 of the reference:
// compute_iou / precision / calc_map rasterise both ellipses with cv2.ellipse on a 512x384 canvas and count
// AND / OR pixels -- 72 pairs per image, minutes on the CPU for a validation set).
//
// One workgroup per (image, predictor) pair.  The inside test is the analytic one of
// spnet_amd/diagnostics.py:create_ellipse_image evaluated at the pixel centres, in the same arithmetic
// (angle and its sine / cosine in fp32, everything else in fp64), so device and host counts agree except
// where fp32 sin/cos differ in the last place between the two math libraries (a boundary pixel or two).
#include "common.h"

struct EllipseD {
  bool active;
  double cx, cy, a, b, cs, sn;
  int x0, x1, y0, y1;
};

__device__ __forceinline__ EllipseD make_ellipse(const float* __restrict__ v, int nx, int ny) {
  EllipseD e;
  const float a = v[2], b = v[3], noobj = v[6];
  e.active = (noobj < 0.5f) && a > 0.f && b > 0.f;
  e.cx = (double)v[0]; e.cy = (double)v[1]; e.a = (double)a; e.b = (double)b;
  const float th = -(atan2f(v[5], v[4]) / 2.0f);        // the reference draws with -angle (image y points down)
  e.cs = (double)cosf(th);
  e.sn = (double)sinf(th);
  const double r = (double)fmaxf(a, b) + 1.0;
  e.x0 = (int)fmax(0.0, floor(e.cx - r));
  e.x1 = (int)fmin((double)nx, ceil(e.cx + r) + 1.0);
  e.y0 = (int)fmax(0.0, floor(e.cy - r));
  e.y1 = (int)fmin((double)ny, ceil(e.cy + r) + 1.0);
  if (e.x1 <= e.x0 || e.y1 <= e.y0) e.active = false;
  return e;
}

__device__ __forceinline__ bool inside(const EllipseD& e, int x, int y) {
  if (!e.active || x < e.x0 || x >= e.x1 || y < e.y0 || y >= e.y1) return false;
  const double dx = (double)x - e.cx, dy = (double)y - e.cy;
  const double u = dx * e.cs + dy * e.sn, w = -dx * e.sn + dy * e.cs;
  const double p = u / e.a, q = w / e.b;
  return p * p + q * q <= 1.0;
}

// iou[pair] = |P & T| / |P | T|, or -1 when the true slot is empty (noobj > 0.99) or both rasters are empty.
__global__ __launch_bounds__(256) void ellipse_iou_kernel(const float* __restrict__ yp,
                                                          const float* __restrict__ yt, int nx, int ny,
                                                          double* __restrict__ iou) {
  __shared__ int red[2][4];
  const long pair = blockIdx.x;
  const float* p = yp + pair * 8;
  const float* t = yt + pair * 8;
  if (t[6] > 0.99f) {
    if (threadIdx.x == 0) iou[pair] = -1.0;
    return;
  }
  const EllipseD ep = make_ellipse(p, nx, ny), et = make_ellipse(t, nx, ny);
  int ci = 0, cu = 0;
  if (ep.active || et.active) {
    const int x0 = ep.active ? (et.active ? min(ep.x0, et.x0) : ep.x0) : et.x0;
    const int x1 = ep.active ? (et.active ? max(ep.x1, et.x1) : ep.x1) : et.x1;
    const int y0 = ep.active ? (et.active ? min(ep.y0, et.y0) : ep.y0) : et.y0;
    const int y1 = ep.active ? (et.active ? max(ep.y1, et.y1) : ep.y1) : et.y1;
    const int bw = x1 - x0, n = bw * (y1 - y0);
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      const int x = x0 + i % bw, y = y0 + i / bw;
      const bool a = inside(ep, x, y), b = inside(et, x, y);
      ci += (a && b) ? 1 : 0;
      cu += (a || b) ? 1 : 0;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    ci += __shfl_xor(ci, off, 64);
    cu += __shfl_xor(cu, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ci; red[1][threadIdx.x >> 6] = cu; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int ti = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const int tu = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    iou[pair] = (ti == 0 && tu == 0) ? -1.0 : (double)ti / (double)tu;
  }
}

// yp / yt: [npairs][8] DENORMALISED predictor rows (cx, cy, a, b, cos2t, sin2t, noobj, rings); iou: npairs doubles.
extern "C" int spnet_ellipse_iou(const float* yp, const float* yt, long npairs, int nx, int ny, double* iou,
                                 void* stream) {
  if (!yp || !yt || !iou || npairs < 1 || nx < 1 || ny < 1 || npairs > 0x7fffffffL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(ellipse_iou_kernel, dim3((unsigned)npairs), dim3(256), 0, (hipStream_t)stream, yp, yt, nx, ny, iou);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------- count metrics (spnet/diagnostics.py:13-59: calc_errors)
// Yp / Yt: [N][ncols] de-normalised grids, 8 variables per predictor (noobj at 6, rings at 7).  counts[7] (int, zeroed
// by the caller) = ring_miscounts, ring_truecounts, total_obj, false_obj_pos, false_obj_neg, true_obj_pos,
// true_obj_neg; pix_err[N] = centre distance of the FIRST predictor of each row (the reference's diff[:,0], diff[:,1]).
// int(round(x)) in the reference is round-half-even = rintf.  Integer atomics: the result does not depend on order.
__global__ __launch_bounds__(256) void calc_errors_kernel(const float* __restrict__ yp, const float* __restrict__ yt,
                                                          long npairs, int npred, int* __restrict__ counts,
                                                          float* __restrict__ pix_err) {
  __shared__ int sc[7];
  if (threadIdx.x < 7) sc[threadIdx.x] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (long)gridDim.x * blockDim.x) {
    const float* p = yp + i * 8;
    const float* t = yt + i * 8;
    const bool there = (int)rintf(t[6]) == 0, predicted = (int)rintf(p[6]) == 0;
    if (there) {
      atomicAdd(&sc[2], 1);
      if (predicted) {
        atomicAdd(&sc[5], 1);
        atomicAdd(fabsf(t[7] - p[7]) > 0.5f ? &sc[0] : &sc[1], 1);
      } else {
        atomicAdd(&sc[4], 1);
      }
    } else {
      atomicAdd(predicted ? &sc[3] : &sc[6], 1);
    }
    if (i % npred == 0) {
      const float dx = p[0] - t[0], dy = p[1] - t[1];
      pix_err[i / npred] = sqrtf(dx * dx + dy * dy);
    }
  }
  __syncthreads();
  if (threadIdx.x < 7 && sc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sc[threadIdx.x]);
}

extern "C" int spnet_calc_errors(const float* yp, const float* yt, long N, int ncols, int* counts, float* pix_err,
                                 void* stream) {
  if (N < 1 || ncols < 8 || (ncols & 7) || !counts || !pix_err) return (int)hipErrorInvalidValue;
  const long npairs = N * (ncols / 8);
  hipLaunchKernelGGL(calc_errors_kernel, dim3(spnet_ew_grid(npairs, 256)), dim3(256), 0, (hipStream_t)stream, yp, yt,
                     npairs, ncols / 8, counts, pix_err);
  SPNET_RETURN_LAUNCH_STATUS();
}
