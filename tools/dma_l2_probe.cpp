// Does a kernel see the bytes an H2D DMA (hipMemcpyAsync from pinned memory, on ANOTHER stream, ordered by an event) has just
// written into a device buffer that earlier kernels read — or can stale lines of an XCD's L2 answer?  Prints mismatching bytes per round.
//   hipcc --offload-arch=gfx950 -O2 -o dma_l2_probe tools/dma_l2_probe.cpp && ./dma_l2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void count_mismatch(const unsigned char *buf, size_t n, unsigned char want_base, unsigned long long *bad) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
  unsigned long long b = 0;
  for (; i < n; i += step) b += buf[i] != (unsigned char)(want_base + (i & 63)) ? 1 : 0;
  if (b) atomicAdd(bad, b);
}
int main() {
  const size_t n = 12 * 307200;
  unsigned char *h[2], *d; unsigned long long *bad, hb;
  CK(hipHostMalloc((void **)&h[0], n)); CK(hipHostMalloc((void **)&h[1], n));
  CK(hipMalloc((void **)&d, n)); CK(hipMalloc((void **)&bad, 8));
  hipStream_t comp, copy; hipEvent_t ev, ev2;
  CK(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
  unsigned long long total = 0;
  for (int round = 0; round < 200; round++) {
    unsigned char base = (unsigned char)(round * 7 + 1);
    unsigned char *src = h[round & 1];
    for (size_t i = 0; i < n; i++) src[i] = (unsigned char)(base + (i & 63));
    CK(hipEventRecord(ev2, comp)); CK(hipStreamWaitEvent(copy, ev2, 0));       // the copy starts behind the readers queued so far
    CK(hipMemcpyAsync(d, src, n, hipMemcpyHostToDevice, copy));
    CK(hipEventRecord(ev, copy)); CK(hipStreamWaitEvent(comp, ev, 0));          // the readers start behind the copy
    CK(hipMemsetAsync(bad, 0, 8, comp));
    for (int k = 0; k < 3; k++) hipLaunchKernelGGL(count_mismatch, dim3(2048), dim3(256), 0, comp, d, n, base, bad);
    CK(hipMemcpyAsync(&hb, bad, 8, hipMemcpyDeviceToHost, comp));
    CK(hipStreamSynchronize(comp));
    if (hb) printf("round %d: %llu stale bytes (3 passes)\n", round, hb);
    total += hb;
  }
  printf("total stale bytes over 200 rounds: %llu\n", total);
  return 0;
}
