// Probe (not product code): a stand-in for RCCL's channel kernels on ONE GPU.  N workgroups of 256 threads occupy N CUs
// for a given time (spinning on the 100 MHz real-time counter, a few dependent loads from a small buffer per poll so
// that the memory path sees some traffic too), launched on a side stream where the gradient all-reduce of the Dense-head
// bucket would start.  tools/channel_hog.py measures what the train step loses to them.
// Exit condition every wave reaches: elapsed ticks >= `ticks`, or `max_polls` polls (a hard cap, ~10x the time).
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void hog_kernel(unsigned long long ticks, unsigned long long max_polls,
                                                  const float* __restrict__ buf, int buf_floats, float* __restrict__ sink) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  unsigned idx = threadIdx.x + blockIdx.x * 256u;
  for (unsigned long long p = 0; p < max_polls; ++p) {
    acc += buf[idx % (unsigned)buf_floats];
    idx = idx * 1664525u + 1013904223u;
    if (__builtin_amdgcn_s_memrealtime() - t0 >= ticks) break;
  }
  if (acc == 1234.5f) sink[0] = acc;      // (keeps the loads alive)
}

extern "C" int probe_hog(int workgroups, double microseconds, const void* buf, int buf_floats, void* sink, void* stream) {
  if (workgroups < 1) return 0;
  const unsigned long long ticks = (unsigned long long)(microseconds * 100.0);        // 100 MHz
  const unsigned long long max_polls = (unsigned long long)(microseconds * 100.0) + 1000;   // >= 0.1 us per poll
  hipLaunchKernelGGL(hog_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, ticks, max_polls, (const float*)buf,
                     buf_floats, (float*)sink);
  return (int)hipGetLastError();
}
