// bf16x3 operands in HBM: the three bf16 pieces of an fp32 matrix X[R][K] ("planes": x = h + m + l exactly), stored the
// way the GEMM kernels' LDS image wants them (round 5).
//
//   plane p (0 = high .. 2 = low), element (r, k) at
//     p * PS + ((r / 16) * nk + k / 32) * 512 + (r % 16) * 32 + ((((k % 32) / 8) ^ -((r % 16) / 4)) & 3) * 8 + k % 8
//   nk = ceil(K / 32), PS = ceil(R / 16) * nk * 512 elements (spnet_bf16x3_plane_elems); rows past R and columns past K
//   are zero (the buffers are allocated zeroed and the producers only write valid elements).
//
// A "piece" = 16 rows x 32 columns of one plane = 1 KiB of CONTIGUOUS memory, its four 16-byte chunks per row already
// XOR-swizzled for conflict-free ds_read_b128 / ds_read_b64_tr_b16: one wave instruction of LDS-DMA
// (buffer_load_dwordx4 ... lds) moves a piece global -> LDS as eight full 128-byte lines.  With row-major planes the same
// instruction read 16 rows x 64 bytes -- half lines, every line touched again by the next K step -- and the L2 -> LDS path,
// not the matrix pipe, set the GEMM's time (tools/x3pp_probe.py: LDS-DMA alone 1.80 us per K step of a 96 x 96 tile pair
// against 0.86 us of MFMAs; 1.29 us with pieces).  The same image serves all three products of a pointwise convolution:
// forward and data gradient read it row-wise (k along the row), the weight gradient column-wise (ds_read_b64_tr_b16).
// Producers write it directly: the depthwise forward kernels (z), the BatchNorm backward kernels (dy), the per-step
// weight split (W, W^T).
#pragma once
#include "common.h"

typedef __bf16 x3_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float x3_f32x2_t __attribute__((ext_vector_type(2)));

// Two fp32 -> two bf16 (round to nearest even, NaN stays NaN) in one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned x3_cvt_pk(float a, float b) {
  x3_f32x2_t v = {a, b};
  x3_bf16x2_t r = __builtin_convertvector(v, x3_bf16x2_t);
  return __builtin_bit_cast(unsigned, r);
}
// (x0, x1) -> packed (h, m, l) pairs: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); both differences are exact
__device__ __forceinline__ void x3_split_pk(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = x3_cvt_pk(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
  m = x3_cvt_pk(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
  l = x3_cvt_pk(s0, s1);
}

static inline long x3t_plane_elems(long R, int K) { return ((R + 15) / 16) * (long)((K + 31) / 32) * 512; }

// element offset of (row, col) inside one plane
__device__ __forceinline__ long x3t_off(long row, int col, int nk) {
  const int r16 = (int)(row & 15);
  return ((row >> 4) * nk + (col >> 5)) * 512 + r16 * 32 + ((((col & 31) >> 3) ^ (0 - (r16 >> 2))) & 3) * 8 + (col & 7);
}
// the same in 32-bit arithmetic, as a BYTE offset (plane sets are at most 2 GiB)
__device__ __forceinline__ int x3t_off32_bytes(int row, int col, int nk) {
  const int r16 = row & 15;
  return (((row >> 4) * nk + (col >> 5)) << 10) + (r16 << 6) + (((((col & 31) >> 3) ^ (0 - (r16 >> 2))) & 3) << 4) + ((col & 7) << 1);
}
// four consecutive columns (col % 4 == 0) of one row: 8 bytes per plane
__device__ __forceinline__ void x3t_store4(unsigned short* __restrict__ planes, long plane_stride, long off, float a, float b,
                                           float c, float d) {
  unsigned h0, m0, l0, h1, m1, l1;
  x3_split_pk(a, b, h0, m0, l0);
  x3_split_pk(c, d, h1, m1, l1);
  *reinterpret_cast<uint2*>(planes + off) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(planes + plane_stride + off) = make_uint2(m0, m1);
  *reinterpret_cast<uint2*>(planes + 2 * plane_stride + off) = make_uint2(l0, l1);
}
// two consecutive columns (col % 2 == 0): 4 bytes per plane
__device__ __forceinline__ void x3t_store2(unsigned short* __restrict__ planes, long plane_stride, long off, float a, float b) {
  unsigned h, m, l;
  x3_split_pk(a, b, h, m, l);
  *reinterpret_cast<unsigned*>(planes + off) = h;
  *reinterpret_cast<unsigned*>(planes + plane_stride + off) = m;
  *reinterpret_cast<unsigned*>(planes + 2 * plane_stride + off) = l;
}
