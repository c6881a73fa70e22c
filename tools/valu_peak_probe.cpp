// What is the chip's VALU issue rate in WAVE instructions per second — the roof bench.py prices `roofline.valu` against?
// Every wave runs a long unrolled stream of independent instructions of ONE kind (inline assembly, so the compiler cannot fuse or
// pack them); 1 to 8 one-wave workgroups per SIMD.  The nominal figure (256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles) is reached by
// nothing; plain 32-bit instructions come to about two thirds of it, packed / permute / 64-bit ones to about 0.42.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_peak_probe tools/valu_peak_probe.cpp && /tmp/valu_peak_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kUnroll = 16;  // independent destination registers per lane

enum Kind { kFmaF32, kAddU32, kPkFmaF32, kPkFmaF16, kPkMin3F16, kPermB32, kMadU32U24, kMadU64U32, kFmaF64, kMin3F32 };

template <int kKind>
__global__ __launch_bounds__(64) void stream(float *sink, int iters, float seed) {
  float a[kUnroll];
  double d[kUnroll];
#pragma unroll
  for (int k = 0; k < kUnroll; k++) { a[k] = seed + k + threadIdx.x; d[k] = seed + k; }
  const float c1 = seed * 1.0001f, c2 = seed * 0.5f;
  const double d1 = seed * 1.0000001, d2 = 0.5;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      if (kKind == kFmaF32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c1), "v"(c2));
      if (kKind == kAddU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(c1));
      if (kKind == kPkFmaF16) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c1), "v"(c2));
      if (kKind == kPkMin3F16) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c1), "v"(c2));
      if (kKind == kPermB32) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c1), "v"(c2));
      if (kKind == kMadU32U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c1), "v"(c2));
      if (kKind == kMin3F32) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(c1), "v"(c2));
      if (kKind == kFmaF64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(d1), "v"(d2));
      if (kKind == kPkFmaF32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[k]) : "v"(d1), "v"(d2));
      if (kKind == kMadU64U32) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[k]) : "v"(c1), "v"(c2) : "vcc");
    }
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < kUnroll; k++) s += a[k] + static_cast<float>(d[k]);
  if (s == 12345.678f) sink[0] = s;
}

template <int kKind>
static int run(const char *what, float *sink) {
  const int iters = 4096;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int waves_per_simd : {1, 2, 4, 8}) {
    const int grid = 256 * 4 * waves_per_simd;  // one-wave workgroups: waves_per_simd per SIMD when the dispatcher spreads them evenly
    hipLaunchKernelGGL(stream<kKind>, dim3(grid), dim3(64), 0, 0, sink, 16, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(stream<kKind>, dim3(grid), dim3(64), 0, 0, sink, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = static_cast<double>(grid) * iters * kUnroll;
    printf("%-20s %d wave(s) per SIMD: %7.3f ms  %.3e wave instructions/s  (%.2f per SIMD and cycle at 2.4 GHz)\n", what, waves_per_simd, ms, insts / (ms * 1e-3),
           insts / (ms * 1e-3) / (1024.0 * 2.4e9));
  }
  return 0;
}

int main() {
  float *sink;
  CK(hipMalloc(&sink, 4096));
  run<kFmaF32>("v_fma_f32", sink);
  run<kAddU32>("v_add_u32", sink);
  run<kMin3F32>("v_min3_f32", sink);
  run<kMadU32U24>("v_mad_u32_u24", sink);
  run<kPermB32>("v_perm_b32", sink);
  run<kPkFmaF16>("v_pk_fma_f16", sink);
  run<kPkMin3F16>("v_pk_minimum3_f16", sink);
  run<kPkFmaF32>("v_pk_fma_f32", sink);
  run<kMadU64U32>("v_mad_u64_u32", sink);
  run<kFmaF64>("v_fma_f64", sink);
  return 0;
}
