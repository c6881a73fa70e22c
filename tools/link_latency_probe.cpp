// What does a saturated H2D transfer stream do to the SMALL traffic of a tracking step?  One compute stream runs chains shaped like
// a group-step's (dependent kernels, a few of them reading job records from pinned host memory, the last one writing a sequence
// number into pinned host memory that the host polls); a second thread keeps 64 MB H2D copies in flight on its own stream.
// Printed: host-side latency of a chain, idle and under the copies, for four kinds of chain.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/link_latency_probe tools/link_latency_probe.cpp -lpthread && /tmp/link_latency_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_nothing(int *sink, int never) { if (never) sink[threadIdx.x] = 1; }
__global__ void k_pull(const uint4 *host_src, uint4 *dst, int n16) {  // like stage_push_kernel: a few KB over the bus
  for (int i = threadIdx.x; i < n16; i += blockDim.x) dst[i] = host_src[i];
}
__global__ void k_flag(volatile unsigned *host_flag, unsigned v) { if (threadIdx.x == 0) *host_flag = v; }
__global__ void k_busy(int *sink, int spins) {  // ~100 us of one-wave workgroups
  int v = threadIdx.x + blockIdx.x;
  for (int i = 0; i < spins; i++) { v = v * 3 + 1; v ^= v >> 3; v += i; v *= 5; }
  if (v == 0x7fffffff) sink[0] = v;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t s, sc;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  int *sink; CK(hipMalloc(&sink, 1 << 20));
  uint4 *d_rec; CK(hipMalloc(&d_rec, 4 << 20));
  uint4 *h_rec; CK(hipHostMalloc(&h_rec, 1 << 16, hipHostMallocDefault));
  memset(h_rec, 1, 1 << 16);
  volatile unsigned *h_flag; CK(hipHostMalloc((void **)&h_flag, 4096, hipHostMallocDefault));
  *h_flag = 0;
  const size_t piece = 64u << 20;
  void *h_big, *d_big;
  CK(hipHostMalloc(&h_big, piece, hipHostMallocDefault));
  memset(h_big, 3, piece);
  CK(hipMalloc(&d_big, piece));
  std::atomic<int> copy_on{0}, quit{0};
  std::atomic<long> pieces{0};
  std::thread copier([&] {
    while (!quit.load()) {
      if (!copy_on.load()) { std::this_thread::sleep_for(std::chrono::microseconds(200)); continue; }
      for (int i = 0; i < 4; i++) (void)hipMemcpyAsync(d_big, h_big, piece, hipMemcpyHostToDevice, sc);
      (void)hipStreamSynchronize(sc);
      pieces += 4;
    }
  });
  unsigned seq = 0;
  auto wait_flag = [&](unsigned v) { while (*h_flag != v) { } };
  struct Kind { const char *name; int pulls; int busy; };
  const Kind kinds[] = {{"20 empty kernels + flag", 0, 0}, {"20 kernels, each pulling 4 KB of pinned host memory, + flag", 20, 0},
                        {"20 kernels of ~50 us + flag", 0, 1}, {"20 kernels of ~50 us, 6 of them pulling 4 KB, + flag", 6, 1}};
  for (int load = 0; load < 2; load++) {
    copy_on = load;
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    const long p0 = pieces.load();
    const double tl0 = now_us();
    for (const Kind &k : kinds) {
      const int reps = 200;
      double sum = 0, worst = 0;
      for (int r = 0; r < reps + 5; r++) {
        const double t0 = now_us();
        for (int i = 0; i < 20; i++) {
          if (i < k.pulls) hipLaunchKernelGGL(k_pull, dim3(1), dim3(256), 0, s, h_rec, d_rec, 256);
          else if (k.busy) hipLaunchKernelGGL(k_busy, dim3(4096), dim3(64), 0, s, sink, 3000);
          else hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, s, sink, 0);
        }
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, h_flag, ++seq);
        wait_flag(seq);
        const double dt = now_us() - t0;
        if (r >= 5) { sum += dt; worst = dt > worst ? dt : worst; }
      }
      printf("%s  %-62s mean %8.1f us  worst %8.1f us\n", load ? "H2D copies in flight:" : "link idle:           ", k.name, sum / reps, worst);
    }
    {  // small DMA copies on the compute stream: do they queue behind the big transfers of the other stream?
      void *h_small; if (hipHostMalloc(&h_small, 4 << 20, hipHostMallocDefault) != hipSuccess) return 1;
      for (int dir = 0; dir < 2; dir++)
        for (size_t bytes : {size_t(4096), size_t(65536), size_t(1) << 20, size_t(4) << 20}) {
          double sum = 0, worst = 0;
          const int reps = 100;
          for (int r = 0; r < reps + 3; r++) {
            hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, s, sink, 0);
            const double t0 = now_us();
            if (dir == 0) (void)hipMemcpyAsync(h_small, d_rec, bytes, hipMemcpyDeviceToHost, s);
            else (void)hipMemcpyAsync(d_rec, h_small, bytes, hipMemcpyHostToDevice, s);
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(64), 0, s, h_flag, ++seq);
            wait_flag(seq);
            const double dt = now_us() - t0;
            if (r >= 3) { sum += dt; worst = dt > worst ? dt : worst; }
          }
          printf("%s  %s hipMemcpyAsync of %7zu B on the compute stream + flag%*s mean %8.1f us  worst %8.1f us\n", load ? "H2D copies in flight:" : "link idle:           ",
                 dir == 0 ? "D2H" : "H2D", bytes, 14, "", sum / reps, worst);
        }
      (void)hipHostFree(h_small);
    }
    if (load) printf("  (copies ran at %.1f GB/s meanwhile)\n", (pieces.load() - p0) * (double)piece / (now_us() - tl0) / 1e3);
  }
  quit = 1;
  copier.join();
  return 0;
}
