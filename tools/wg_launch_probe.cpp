// How long does the chip need merely to START n one-wave workgroups (64 threads, some LDS each) that do nothing?  The detection and
// search kernels launch 10^5 such workgroups per dispatch; if this alone takes a good part of their time, they are paced by the
// workgroup dispatcher, not by their instructions.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wg_launch_probe tools/wg_launch_probe.cpp && /tmp/wg_launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int kLdsWords>
__global__ __launch_bounds__(64) void nothing(int *sink, int never) {
  __shared__ int s[kLdsWords];
  s[threadIdx.x] = threadIdx.x;
  if (never) sink[blockIdx.x] = s[(threadIdx.x + 1) & 63];
}
template <int kLdsWords>
__global__ __launch_bounds__(64) void busy(int *sink, int never, int spins) {  // ~spins x 4 dependent VALU instructions per wave
  __shared__ int s[kLdsWords];
  s[threadIdx.x] = threadIdx.x;
  int v = threadIdx.x;
  for (int i = 0; i < spins; i++) { v = v * 3 + 1; v ^= v >> 3; v += i; v *= 5; }
  if (never || v == 0x7fffffff) sink[blockIdx.x] = s[(threadIdx.x + 1) & 63] + v;
}
template <typename F>
static float time_us(F launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.f / reps;
}
int main() {
  int *sink;
  hipMalloc(&sink, 1 << 22);
  for (int n : {8192, 32768, 106496, 425984}) {
    const float t0 = time_us([&] { hipLaunchKernelGGL(nothing<64>, dim3(n), dim3(64), 0, 0, sink, 0); }, 20);
    const float t1 = time_us([&] { hipLaunchKernelGGL(nothing<1440>, dim3(n), dim3(64), 0, 0, sink, 0); }, 20);
    const float t2 = time_us([&] { hipLaunchKernelGGL(busy<1440>, dim3(n), dim3(64), 0, 0, sink, 0, 256); }, 20);
    printf("%7d one-wave workgroups: empty, 256 B LDS %.1f us; empty, 5760 B LDS %.1f us; ~1000 VALU instructions each, 5760 B LDS %.1f us\n", n, t0, t1, t2);
  }
  return 0;
}
