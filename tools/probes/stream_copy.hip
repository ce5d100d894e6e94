// Probe (not product code): what does a pure streaming pass reach on this chip, as a function of how the bytes in flight
// are organised?  y = x over N float4, four organisations:
//   mode 0  grid-stride, U independent 16-byte loads per lane issued before the first store (registers hold the bytes in flight)
//   mode 1  the same with nontemporal loads and stores
//   mode 2  wave-private LDS ring: global_load_lds_dwordx4 (LDS-DMA, no VGPRs) D pieces of 1 KiB ahead, ds_read_b128 +
//           global store behind it -- bytes in flight bounded by LDS (160 KB / CU), not by registers
// tools/stream_copy_probe.py times them (rotating buffers beyond the Infinity Cache).
#include <hip/hip_runtime.h>

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load(const float4* p) {
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store(const float4 a, float4* p) {
  v4f v = {a.x, a.y, a.z, a.w};
  __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_regs_kernel(const float4* __restrict__ x, float4* __restrict__ y, long n4) {
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? nt_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) nt_store(v[u], y + i + u * stride);
      else y[i + u * stride] = v[u];
    }
  }
  for (; i < n4; i += stride) y[i] = x[i];
}

// wave-private ring of D slots x 1 KiB in LDS; a wave owns a contiguous run of `per_wave` KiB-pieces
template <int D>
__global__ __launch_bounds__(256) void copy_ldsring_kernel(const float4* __restrict__ x, float4* __restrict__ y, long pieces,
                                                           long per_wave) {
  extern __shared__ __attribute__((aligned(16))) float4 ring[];      // [4 waves][D][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * 4 + wave;
  const long p0 = gw * per_wave;
  long p1 = p0 + per_wave;
  if (p1 > pieces) p1 = pieces;
  if (p0 >= p1) return;
  float4* my = ring + (long)wave * D * 64;
  const long n = p1 - p0;
  // prologue: D pieces in flight
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if (d < n) {
      const float4* src = x + (p0 + d) * 64 + lane;
      __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(my + d * 64), 16, 0, 0);
    }
  }
  for (long p = 0; p < n; p += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (p + d >= n) break;
      // wait until piece (p + d) has landed.  Younger than its DMA: in the steady state (D-1) stores + (D-1) DMAs; in
      // the first pass only the prologue's later DMAs and the stores / DMAs of this pass: D-1+d >= D-1
      if (p == 0) {
        if (D == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (D == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (D == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      } else {
        if (D == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (D == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (D == 8) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
      }
      const float4 v = my[d * 64 + lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      y[(p0 + p + d) * 64 + lane] = v;
      if (p + d + D < n) {
        const float4* src = x + (p0 + p + d + D) * 64 + lane;
        __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(my + d * 64), 16, 0, 0);
      }
    }
  }
}

extern "C" int probe_copy(const void* x, void* y, long n4, int mode, int depth, int blocks, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const float4* xs = (const float4*)x;
  float4* ys = (float4*)y;
#define REGS(U, NT) hipLaunchKernelGGL((copy_regs_kernel<U, NT>), dim3(blocks), dim3(256), 0, st, xs, ys, n4)
  if (mode == 0 || mode == 1) {
    const bool nt = mode == 1;
    if (depth == 1) { if (nt) REGS(1, true); else REGS(1, false); }
    else if (depth == 2) { if (nt) REGS(2, true); else REGS(2, false); }
    else if (depth == 4) { if (nt) REGS(4, true); else REGS(4, false); }
    else if (depth == 8) { if (nt) REGS(8, true); else REGS(8, false); }
    else if (depth == 16) { if (nt) REGS(16, true); else REGS(16, false); }
    else return 1;
  } else if (mode == 2) {
    const long pieces = n4 / 64;
    const long waves = (long)blocks * 4;
    const long per_wave = (pieces + waves - 1) / waves;
#define RING(D) hipLaunchKernelGGL((copy_ldsring_kernel<D>), dim3(blocks), dim3(256), 4 * D * 1024, st, xs, ys, pieces, per_wave)
    if (depth == 2) RING(2);
    else if (depth == 4) RING(4);
    else if (depth == 8) RING(8);
    else if (depth == 16) RING(16);
    else return 1;
  } else return 1;
  return (int)hipGetLastError();
}
