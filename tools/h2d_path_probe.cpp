// Which engine carries a large pinned H2D hipMemcpyAsync: SDMA or ROCclr's copy kernel (__amd_rocclr_copyBuffer)?  Runs the same
// 79 MB transfer in several situations and prints the rate of each; under `rocprofv3 --kernel-trace --memory-copy-trace` the
// kernel trace shows a copy-kernel dispatch for every transfer that did NOT go through SDMA.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/h2d_path_probe tools/h2d_path_probe.cpp && /tmp/h2d_path_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_busy(int *sink, int spins) {
  int v = threadIdx.x + blockIdx.x;
  for (int i = 0; i < spins; i++) { v = v * 3 + 1; v ^= v >> 3; v += i; v *= 5; }
  if (v == 0x7fffffff) sink[0] = v;
}
__global__ void k_mark(int *sink, int tag) { if (tag < 0) sink[0] = tag; }  // shows up in the kernel trace between the cases
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 256u * 640 * 480;
  hipStream_t sc, sk;
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
  void *h, *d0, *d1; int *sink;
  CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
  memset(h, 5, bytes);
  CK(hipMalloc(&d0, bytes)); CK(hipMalloc(&d1, bytes)); CK(hipMalloc(&sink, 4096));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  auto rate = [&](const char *what, int n, auto body) {
    CK(hipDeviceSynchronize());
    const double t0 = now_us();
    body();
    CK(hipStreamSynchronize(sc));
    const double dt = now_us() - t0;
    printf("%-78s %6.1f GB/s\n", what, n * (double)bytes / dt / 1e3);
    CK(hipDeviceSynchronize());
    return 0;
  };
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, sk, sink, 1);
  rate("1 transfer, idle device", 1, [&] { (void)hipMemcpyAsync(d0, h, bytes, hipMemcpyHostToDevice, sc); });
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, sk, sink, 2);
  rate("4 transfers queued back to back, idle device", 4, [&] { for (int i = 0; i < 4; i++) (void)hipMemcpyAsync(i & 1 ? d1 : d0, h, bytes, hipMemcpyHostToDevice, sc); });
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, sk, sink, 3);
  rate("4 transfers, an event recorded after each (as the feeder does)", 4, [&] {
    for (int i = 0; i < 4; i++) { (void)hipMemcpyAsync(i & 1 ? d1 : d0, h, bytes, hipMemcpyHostToDevice, sc); (void)hipEventRecord(ev, sc); }
  });
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, sk, sink, 4);
  rate("4 transfers, each behind a hipStreamWaitEvent on a compute stream's event", 4, [&] {
    for (int i = 0; i < 4; i++) {
      hipLaunchKernelGGL(k_busy, dim3(4096), dim3(64), 0, sk, sink, 2000);
      (void)hipEventRecord(ev, sk);
      (void)hipStreamWaitEvent(sc, ev, 0);
      (void)hipMemcpyAsync(i & 1 ? d1 : d0, h, bytes, hipMemcpyHostToDevice, sc);
    }
  });
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, sk, sink, 5);
  rate("4 transfers while kernels run on another stream", 4, [&] {
    for (int i = 0; i < 60; i++) hipLaunchKernelGGL(k_busy, dim3(4096), dim3(64), 0, sk, sink, 2000);
    for (int i = 0; i < 4; i++) (void)hipMemcpyAsync(i & 1 ? d1 : d0, h, bytes, hipMemcpyHostToDevice, sc);
  });
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, sk, sink, 6);
  rate("4 transfers, a kernel on the COPY stream before each", 4, [&] {
    for (int i = 0; i < 4; i++) {
      hipLaunchKernelGGL(k_busy, dim3(64), dim3(64), 0, sc, sink, 200);
      (void)hipMemcpyAsync(i & 1 ? d1 : d0, h, bytes, hipMemcpyHostToDevice, sc);
    }
  });
  hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, sk, sink, 7);
  CK(hipDeviceSynchronize());
  return 0;
}
