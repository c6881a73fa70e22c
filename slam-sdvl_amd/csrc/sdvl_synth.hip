// sdvl_synth.hip — device build of the synthetic plane-scene generator (sdvl_synth.h): renders n views straight
// into HBM so that bench.py starts its timed region with every input frame already resident.
#include "sdvl_internal.h"
#include "sdvl_synth.h"

namespace {
__global__ __launch_bounds__(256) void synth_render_kernel(const sdvl_synth_view *__restrict__ views, int width, int height,
                                                           uint8_t *__restrict__ out, long long frame_bytes) {
  const sdvl_synth_view view = views[blockIdx.z];
  const int u = blockIdx.x * 64 + (threadIdx.x & 63);
  const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (u >= width || v >= height) return;
  out[static_cast<size_t>(blockIdx.z) * frame_bytes + static_cast<size_t>(v) * width + u] = sdvl_synth_pixel(&view, u, v);
}
}  // namespace

extern "C" int sdvl_synth_render(sdvl_ctx *ctx, int n, const sdvl_synth_view *views, int width, int height, void *dev_out,
                                 int64_t frame_bytes) {
  if (!ctx || n < 0 || (n > 0 && (!views || !dev_out))) return SDVL_ERR_INVALID;
  if (n == 0) return SDVL_OK;
  SDVL_REQUIRE(ctx, width > 0 && height > 0 && frame_bytes >= static_cast<int64_t>(width) * height, "bad frame geometry");
  SDVL_REQUIRE(ctx, n <= 65535, "at most 65535 views per call");
  const size_t bytes = sizeof(sdvl_synth_view) * n;
  void *hs = nullptr, *dsx = nullptr;
  int rc = sdvl_stage_alloc(ctx, bytes, &hs, &dsx);
  if (rc) return rc;
  memcpy(hs, views, bytes);
  SDVL_HIP_CHECK(ctx, sdvl_push(ctx, dsx, hs, bytes));
  hipLaunchKernelGGL(synth_render_kernel, dim3((width + 63) / 64, (height + 3) / 4, n), dim3(256), 0, ctx->stream,
                     static_cast<const sdvl_synth_view *>(dsx), width, height, static_cast<uint8_t *>(dev_out),
                     static_cast<long long>(frame_bytes));
  SDVL_HIP_CHECK(ctx, hipGetLastError());
  SDVL_HIP_CHECK(ctx, sdvl_stream_wait(ctx));
  return SDVL_OK;
}
