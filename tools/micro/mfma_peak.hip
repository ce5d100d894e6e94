// Dev microbenchmark: what a bare v_mfma_f32_16x16x4_f32 loop reaches at the GEMM kernels' occupancy
// (256-thread workgroups, 2 per CU) for launch durations from ~20 us to ~1 ms.  Build on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grids[] = {256, 512, 1024};
  const int iters_list[] = {64, 128, 256, 512, 2048, 8192};
  for (int g : grids)
    for (int iters : iters_list) {
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop<16>, dim3(g), dim3(256), 0, 0, out, iters, 1.f, 2.f);
      hipDeviceSynchronize();
      const int reps = 20;
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<16>, dim3(g), dim3(256), 0, 0, out, iters, 1.f, 2.f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps;
      const double flop = (double)g * 4 * iters * 16 * 2048.0;
      printf("grid %4d iters %5d: %8.1f us  %6.1f TFLOP/s\n", g, iters, us, flop / us / 1e6);
    }
  return 0;
}
