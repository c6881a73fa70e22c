// Does instruction-cache pressure from OTHER kernels slow a kernel whose own code fits?  
//   victim    : many one-wave workgroups, each runs ~20 KB of straight-line vector code once (the shape of fast_cells / search_points)
//   aggressor : ONE wave per CU looping over ~N KB of straight-line code (N = 32 .. 256): a few % of the vector pipe, all of the I-cache
// Prints the victim's time alone and with each aggressor running on a second stream.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/icache_probe tools/icache_probe.cpp && /tmp/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMA asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define R4(X) X X X X
#define R16(X) R4(R4(X))
#define R64(X) R4(R16(X))
#define R256(X) R4(R64(X))
#define R1024(X) R4(R256(X))

__global__ __launch_bounds__(64) void victim(float *out, float a, float b) {  // 2560 x 8 B = 20 KB
  float x = threadIdx.x;
  R1024(FMA) R1024(FMA) R256(FMA) R256(FMA)
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

template <int kKB>
__global__ __launch_bounds__(64) void aggressor(float *out, float a, float b, const int *stop) {  // kKB of code per loop trip
  float x = threadIdx.x;
  for (int it = 0; it < 400000 / (kKB / 8); it++) {  // at most ~0.8 s even if nobody raises the flag
    if (__builtin_amdgcn_readfirstlane(*(volatile const int *)stop)) break;   // every wave reaches this: the grid drains when asked
#pragma unroll
    for (int k = 0; k < kKB / 8; k++) { R1024(FMA) }  // 1024 x 8 B = 8 KB per repetition
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int kKB>
static int run_with(hipStream_t sv, hipStream_t sa, float *dv, float *da, int *dstop, int *hstop, int n_victim, float alone_ms) {
  *hstop = 0;
  hipLaunchKernelGGL(aggressor<kKB>, dim3(256), dim3(64), 0, sa, da, 1.0001f, 0.5f, dstop);
  CHECK(hipGetLastError());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(victim, dim3(n_victim), dim3(64), 0, sv, dv, 1.0001f, 0.5f);
  CHECK(hipEventRecord(e0, sv));
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL(victim, dim3(n_victim), dim3(64), 0, sv, dv, 1.0001f, 0.5f);
  CHECK(hipEventRecord(e1, sv));
  CHECK(hipStreamSynchronize(sv));
  *hstop = 1;  // host-coherent flag: the aggressor's waves leave their loop
  CHECK(hipStreamSynchronize(sa));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("victim with a %3d-KB aggressor wave on every CU: %.3f ms per launch (x%.2f)\n", kKB, ms / 10, ms / 10 / alone_ms);
  return 0;
}

int main() {
  const int n_victim = 100000;  // one-wave workgroups
  float *dv, *da;
  int *hstop, *dstop;
  CHECK(hipMalloc(&dv, sizeof(float) * 64 * n_victim));
  CHECK(hipMalloc(&da, sizeof(float) * 64 * 256));
  CHECK(hipHostMalloc(&hstop, sizeof(int), hipHostMallocCoherent));
  CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&dstop), hstop, 0));
  hipStream_t sv, sa;
  CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(victim, dim3(n_victim), dim3(64), 0, sv, dv, 1.0001f, 0.5f);
  CHECK(hipEventRecord(e0, sv));
  for (int r = 0; r < 10; r++) hipLaunchKernelGGL(victim, dim3(n_victim), dim3(64), 0, sv, dv, 1.0001f, 0.5f);
  CHECK(hipEventRecord(e1, sv));
  CHECK(hipStreamSynchronize(sv));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const float alone = ms / 10;
  printf("victim alone (100000 waves x 2560 instructions, 20 KB of code): %.3f ms per launch = %.2e wave instructions/s\n", alone, 100000.0 * 2560 / (alone * 1e-3));
  if (run_with<8>(sv, sa, dv, da, dstop, hstop, n_victim, alone)) return 1;
  if (run_with<32>(sv, sa, dv, da, dstop, hstop, n_victim, alone)) return 1;
  if (run_with<64>(sv, sa, dv, da, dstop, hstop, n_victim, alone)) return 1;
  if (run_with<128>(sv, sa, dv, da, dstop, hstop, n_victim, alone)) return 1;
  if (run_with<256>(sv, sa, dv, da, dstop, hstop, n_victim, alone)) return 1;
  return 0;
}
