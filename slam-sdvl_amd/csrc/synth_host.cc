// Host build of the synthetic sequence generator (sdvl_synth.h) -> libsdvl_synth.so, used by the CPU tests.
#include "sdvl_synth.h"

extern "C" int sdvl_synth_render_host(const sdvl_synth_view *view, int width, int height, uint8_t *out, int stride) {
  for (int v = 0; v < height; v++)
    for (int u = 0; u < width; u++) out[(long)v * stride + u] = sdvl_synth_pixel(view, u, v);
  return 0;
}
