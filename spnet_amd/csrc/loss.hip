// Grid-cell ellipse loss (forward + closed-form gradient) and prediction decode.
//
//   custom_loss / my_loss       spnet/models.py:557-633   (lambda-weighted MSE, optional BCE-with-logits
//                                                          on the noobj column when loss_type != 'same')
//   denorm_Y + cleanup          spnet/utils.py:186-188, 56-64; evaluate_spnet.py:70-73
//
// One 64-lane wavefront per sample: each lane owns predictors lane, lane+64, ... (8 variables each
// = two 16-byte loads per tensor), the five loss terms are reduced with a shuffle butterfly.
#include "common.h"

#define LAMBDA_CENTER 2.0f
#define LAMBDA_SIZE 1.0f
#define LAMBDA_ANGLE 3.0f
#define LAMBDA_NOOBJ 0.3f
#define LAMBDA_CLASS 5.0f

// parts[b][5] = per-sample (center, size, angle, noobj, class) sums (not yet divided by ncols);
// grad[b][ncols] = d mean_b(L_b / ncols) / d y_pred  (nullptr to skip).
__global__ __launch_bounds__(64) void ellipse_loss_kernel(const float* __restrict__ y_true,
                                                          const float* __restrict__ y_pred,
                                                          float* __restrict__ grad,
                                                          float* __restrict__ parts, int Bn, int ncols,
                                                          int hybrid) {
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int npred = ncols >> 3;
  const float inv = 1.0f / ((float)ncols * (float)Bn);
  float s_center = 0.f, s_size = 0.f, s_angle = 0.f, s_noobj = 0.f, s_class = 0.f;
  for (int p = lane; p < npred; p += 64) {
    const long o = (long)b * ncols + p * 8;
    const float4 t0 = *reinterpret_cast<const float4*>(y_true + o);
    const float4 t1 = *reinterpret_cast<const float4*>(y_true + o + 4);
    const float4 y0 = *reinterpret_cast<const float4*>(y_pred + o);
    const float4 y1 = *reinterpret_cast<const float4*>(y_pred + o + 4);
    const float obj = 1.f - t1.z;                 // 1 - noobj_true
    const float d = t0.z - t0.w;                  // a - b (true)
    const float d2 = d * d;
    const float r0 = t0.x - y0.x, r1 = t0.y - y0.y, r2 = t0.z - y0.z, r3 = t0.w - y0.w;
    const float r4 = t1.x - y1.x, r5 = t1.y - y1.y, r6 = t1.z - y1.z, r7 = t1.w - y1.w;
    s_center += LAMBDA_CENTER * obj * (r0 * r0 + r1 * r1);
    s_size += LAMBDA_SIZE * obj * (r2 * r2 + r3 * r3);
    s_angle += LAMBDA_ANGLE * obj * d2 * (r4 * r4 + r5 * r5);
    s_class += LAMBDA_CLASS * obj * r7 * r7;
    float g6;
    if (hybrid) {
      const float z = y1.z;
      s_noobj += LAMBDA_NOOBJ * (fmaxf(z, 0.f) - z * t1.z + log1pf(expf(-fabsf(z))));
      g6 = LAMBDA_NOOBJ * (1.f / (1.f + expf(-z)) - t1.z) * inv;
    } else {
      s_noobj += LAMBDA_NOOBJ * r6 * r6;
      g6 = -2.f * LAMBDA_NOOBJ * r6 * inv;
    }
    if (grad) {
      const float k = -2.f * obj * inv;
      float4 g0, g1;
      g0.x = k * LAMBDA_CENTER * r0;
      g0.y = k * LAMBDA_CENTER * r1;
      g0.z = k * LAMBDA_SIZE * r2;
      g0.w = k * LAMBDA_SIZE * r3;
      g1.x = k * LAMBDA_ANGLE * d2 * r4;
      g1.y = k * LAMBDA_ANGLE * d2 * r5;
      g1.z = g6;
      g1.w = k * LAMBDA_CLASS * r7;
      *reinterpret_cast<float4*>(grad + o) = g0;
      *reinterpret_cast<float4*>(grad + o + 4) = g1;
    }
  }
  s_center = wave_sum(s_center);
  s_size = wave_sum(s_size);
  s_angle = wave_sum(s_angle);
  s_noobj = wave_sum(s_noobj);
  s_class = wave_sum(s_class);
  if (lane == 0) {
    float* pp = parts + (long)b * 5;
    pp[0] = s_center; pp[1] = s_size; pp[2] = s_angle; pp[3] = s_noobj; pp[4] = s_class;
  }
}

// out[0..4] = batch-mean of each term / ncols, out[5] = their sum (= custom_loss value).
__global__ __launch_bounds__(64) void ellipse_loss_finalize_kernel(const float* __restrict__ parts,
                                                                   int Bn, int ncols,
                                                                   float* __restrict__ out) {
  const int lane = threadIdx.x;
  float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int b = lane; b < Bn; b += 64)
#pragma unroll
    for (int j = 0; j < 5; ++j) s[j] += parts[(long)b * 5 + j];
  float tot = 0.f;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    s[j] = wave_sum(s[j]) / ((float)Bn * (float)ncols);
    tot += s[j];
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < 5; ++j) out[j] = s[j];
    out[5] = tot;
  }
}

// out[b][p][7] = (cx, cy, a, b, angle_deg in (0,180], noobj, rings) in image units.
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ y_norm,
                                                     const float* __restrict__ means,
                                                     const float* __restrict__ ranges,
                                                     float* __restrict__ out, int Bn, int ncols,
                                                     int sigmoid_noobj) {
  const int npred = ncols >> 3;
  const long total = (long)Bn * npred;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % npred);
    const long o = i * 8;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float yn = y_norm[o + j];
      if (j == 6 && sigmoid_noobj) yn = 1.f / (1.f + expf(-yn));
      v[j] = yn * ranges[p * 8 + j] + means[p * 8 + j];
    }
    float ang = atan2f(v[5], v[4]) * 0.5f * 57.29577951308232f;
    if (!(ang > 0.f)) ang += 180.f;
    float* q = out + i * 7;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2]; q[3] = v[3]; q[4] = ang; q[5] = v[6]; q[6] = v[7];
  }
}

// loss_out: 6 floats (center,size,angle,noobj,class,total).  parts: B*5 floats of scratch.
extern "C" int spnet_ellipse_loss(const float* y_true, const float* y_pred, float* grad, float* parts,
                                  float* loss_out, int B, int ncols, int hybrid, void* stream) {
  if (ncols & 7) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ellipse_loss_kernel, dim3(B), dim3(64), 0, st, y_true, y_pred, grad, parts, B,
                     ncols, hybrid);
  hipLaunchKernelGGL(ellipse_loss_finalize_kernel, dim3(1), dim3(64), 0, st, parts, B, ncols, loss_out);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_decode(const float* y_norm, const float* means, const float* ranges, float* out,
                            int B, int ncols, int sigmoid_noobj, void* stream) {
  if (ncols & 7) return (int)hipErrorInvalidValue;
  const long total = (long)B * (ncols / 8);
  hipLaunchKernelGGL(decode_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     y_norm, means, ranges, out, B, ncols, sigmoid_noobj);
  SPNET_RETURN_LAUNCH_STATUS();
}

// ---------------------------------------------------------------- sigmoid output columns
// SelectiveSigmoid (spnet/models.py:277-298) and the 'compound' head (models.py:379-386: Dense(n_preds, sigmoid)
// + Dense(rest) concatenated and re-ordered by InterleaveColumns, i.e. ONE dense layer whose columns start::step
// pass through a sigmoid).  mode 0: y[:, start::step] = sigmoid(y[:, start::step]) in place;
// mode 1: g[:, start::step] *= y (1 - y) with y the POST-sigmoid output (backward).
__global__ __launch_bounds__(256) void selective_sigmoid_kernel(float* __restrict__ y, float* __restrict__ g, int Bn,
                                                                int ncols, int start, int step, int mode) {
  const int per = (ncols - start + step - 1) / step;
  const long total = (long)Bn * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long o = (i / per) * ncols + start + (i % per) * step;
    if (mode == 0) {
      y[o] = 1.f / (1.f + expf(-y[o]));
    } else {
      const float s = y[o];
      g[o] *= s * (1.f - s);
    }
  }
}

extern "C" int spnet_selective_sigmoid(float* y, float* grad, int B, int ncols, int start, int step, int backward,
                                       void* stream) {
  if (B < 1 || ncols < 1 || start < 0 || start >= ncols || step < 1 || (backward && !grad)) return (int)hipErrorInvalidValue;
  const long total = (long)B * ((ncols - start + step - 1) / step);
  hipLaunchKernelGGL(selective_sigmoid_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, y,
                     grad, B, ncols, start, step, backward ? 1 : 0);
  SPNET_RETURN_LAUNCH_STATUS();
}
