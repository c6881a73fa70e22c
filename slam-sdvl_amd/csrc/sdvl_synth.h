// Seeded synthetic sequence generator (SURVEY §8d "Synthetic inputs"): a textured plane n.X = d seen by a
// pinhole camera with pose T_cw.  No dataset ships with the repo, so tests, smoke() and bench.py render their
// frames with this; the same inline code runs on the host (synth_host.cc) and in a HIP kernel (sdvl_synth.hip).
// Texture = 5 octaves of bilinear value noise on lattices of 3..48 px (at the nominal depth) whose lattice
// values come from an integer hash, plus +-2 grey levels of per-frame sensor noise.  Only + - * / floor on
// doubles -> identical bytes on CPU and GPU when both are compiled with -ffp-contract=off.
// Round 5: a second texture, SDVL_TEXTURE_CAMERA — what a camera sees indoors rather than a corner on every second pixel:
// piecewise-smooth shading, weak grain below the FAST threshold, and three scales of soft-edged shapes (rotated rectangles
// and discs); 2-5 k FAST-10 keypoints per 640x480 frame over the three detection levels, a few % of the tested pixels.
#ifndef SDVL_SYNTH_H_
#define SDVL_SYNTH_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define SDVL_HD __host__ __device__
#else
#define SDVL_HD
#endif

struct sdvl_synth_view {
  double fx, fy, u0, v0;
  double R[9];       // rotation of T_cw (world -> camera), row-major
  double t[3];       // translation of T_cw
  double plane[4];   // n.X = d  (world)
  uint32_t seed;     // texture seed
  uint32_t frame_id; // sensor-noise seed
  uint32_t texture;  // SDVL_TEXTURE_PLANE_NOISE (rounds 1-4) or SDVL_TEXTURE_CAMERA
  uint32_t reserved_;
  // Round 6: the lens.  dist = (k1, k2, p1, p2, k3) of the radial-tangential model (Camera.d1..d5 of the reference's cfg files); with
  // dist[0] != 0 the view is what a camera with that lens records — the image Camera::UndistortImage (camera.cc:100-105) turns back into
  // the pinhole view.  dist[0] == 0 (as the reference tests it, camera.cc:46): pinhole, the bytes of rounds 1-5.
  double dist[5];
};
#define SDVL_TEXTURE_PLANE_NOISE 0u
#define SDVL_TEXTURE_CAMERA 1u

SDVL_HD inline uint32_t sdvl_hash3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

SDVL_HD inline double sdvl_floor(double x) {
  double f = (double)(long long)x;
  return (f > x) ? f - 1.0 : f;
}

// value-noise octave: lattice pitch `cell` (world metres), values in [0,1)
SDVL_HD inline double sdvl_value_noise(double X, double Y, double cell, uint32_t seed, uint32_t oct) {
  const double gx = X / cell, gy = Y / cell;
  const double fx0 = sdvl_floor(gx), fy0 = sdvl_floor(gy);
  const int ix = (int)fx0, iy = (int)fy0;
  const double ax = gx - fx0, ay = gy - fy0;
  const double v00 = (sdvl_hash3((uint32_t)ix, (uint32_t)iy, seed + oct) >> 8) * (1.0 / 16777216.0);
  const double v10 = (sdvl_hash3((uint32_t)(ix + 1), (uint32_t)iy, seed + oct) >> 8) * (1.0 / 16777216.0);
  const double v01 = (sdvl_hash3((uint32_t)ix, (uint32_t)(iy + 1), seed + oct) >> 8) * (1.0 / 16777216.0);
  const double v11 = (sdvl_hash3((uint32_t)(ix + 1), (uint32_t)(iy + 1), seed + oct) >> 8) * (1.0 / 16777216.0);
  const double top = v00 + (v10 - v00) * ax;
  const double bot = v01 + (v11 - v01) * ax;
  return top + (bot - top) * ay;
}

// ---- SDVL_TEXTURE_CAMERA -------------------------------------------------------------------------------------------------------
// One lattice of shapes: every cell of pitch `pitch` (world metres) may hold one shape whose centre lies anywhere in the cell, so a
// point looks at the 2 x 2 cells around it.  A shape adds `contrast * mask` grey levels, mask = product of two clamped ramps of
// width `soft` (the lens blur, about 1.2 px) over the signed distances to its sides: rectangles turned by one of eight exact
// rational rotations (no trigonometry: the same bits on host and device), or discs.
SDVL_HD inline double sdvl_clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }

SDVL_HD inline double sdvl_shape_layer(double X, double Y, double pitch, double soft, uint32_t seed, uint32_t layer, uint32_t keep_of_256) {
  const double gx = X / pitch - 0.5, gy = Y / pitch - 0.5;
  const double fx0 = sdvl_floor(gx), fy0 = sdvl_floor(gy);
  const int ix = (int)fx0, iy = (int)fy0;
  double sum = 0.0;
  for (int j = 0; j < 2; j++) {
    for (int i = 0; i < 2; i++) {
      const uint32_t h0 = sdvl_hash3((uint32_t)(ix + i), (uint32_t)(iy + j), seed + 0x51u + layer * 7u);
      if ((h0 & 255u) >= keep_of_256) continue;                       // empty cell
      const uint32_t h1 = sdvl_hash3((uint32_t)(ix + i), (uint32_t)(iy + j), seed + 0x52u + layer * 7u);
      const double cx = ((double)(ix + i) + (double)((h0 >> 8) & 1023u) * (1.0 / 1024.0)) * pitch;
      const double cy = ((double)(iy + j) + (double)((h0 >> 18) & 1023u) * (1.0 / 1024.0)) * pitch;
      const double a = (0.10 + 0.26 * (double)(h1 & 255u) * (1.0 / 256.0)) * pitch;          // half sizes
      const double b = (0.10 + 0.26 * (double)((h1 >> 8) & 255u) * (1.0 / 256.0)) * pitch;
      double contrast = 22.0 + 50.0 * (double)((h1 >> 16) & 255u) * (1.0 / 256.0);
      if ((h1 >> 24) & 1u) contrast = -contrast;
      const uint32_t kind = (h1 >> 25) & 7u;                           // 0..5 rectangle, 6..7 disc
      const uint32_t rot = (h1 >> 28) & 7u;
      const double dx = X - cx, dy = Y - cy;
      double m;
      if (kind >= 6u) {
        const double r = a < b ? a : b;
        const double d = (r * r - (dx * dx + dy * dy)) / (2.0 * r);    // ~ signed distance to the rim near the rim
        m = sdvl_clamp01(d / soft + 0.5);
      } else {
        // (c, s) from exact Pythagorean pairs: 0, 16.3, 36.9, 53.1, 73.7 degrees and mirror images
        const double c = (rot & 3u) == 0u ? 1.0 : ((rot & 3u) == 1u ? 0.96 : ((rot & 3u) == 2u ? 0.8 : 0.6));
        double sn = (rot & 3u) == 0u ? 0.0 : ((rot & 3u) == 1u ? 0.28 : ((rot & 3u) == 2u ? 0.6 : 0.8));
        if (rot & 4u) sn = -sn;
        const double lx = c * dx + sn * dy, ly = c * dy - sn * dx;
        const double ex = a - (lx < 0.0 ? -lx : lx), ey = b - (ly < 0.0 ? -ly : ly);
        m = sdvl_clamp01(ex / soft + 0.5) * sdvl_clamp01(ey / soft + 0.5);
      }
      sum += contrast * m;
    }
  }
  return sum;
}

// grey level (before sensor noise) of the camera-like texture at world point (X, Y); `pix` = world size of a pixel there
SDVL_HD inline double sdvl_camera_texture(double X, double Y, double pix, uint32_t seed) {
  const double soft = 1.2 * pix;
  double g = 118.0 + 70.0 * (sdvl_value_noise(X, Y, 0.45, seed, 8) - 0.5) + 26.0 * (sdvl_value_noise(X, Y, 0.12, seed, 9) - 0.5) +
             7.0 * (sdvl_value_noise(X, Y, 0.0116, seed, 10) - 0.5);
  g += sdvl_shape_layer(X, Y, 0.080, soft, seed, 0, 205u);
  g += sdvl_shape_layer(X, Y, 0.150, soft, seed, 1, 225u);
  g += sdvl_shape_layer(X, Y, 0.290, soft, seed, 2, 230u);
  g += sdvl_shape_layer(X, Y, 0.600, soft, seed, 3, 200u);
  return g;
}

SDVL_HD inline uint8_t sdvl_synth_pixel(const sdvl_synth_view *s, int u, int v) {
  // ray in camera coords, rotate to world: R^T * r ; camera centre C = -R^T t
  double rx = (u - s->u0) / s->fx, ry = (v - s->v0) / s->fy;
  const double rz = 1.0;
  if (s->dist[0] != 0.0) {
    // (rx, ry) is where the LENS put the ray: invert x_d = x kr + 2 p1 x y + p2 (r2 + 2 x2), y_d = y kr + p1 (r2 + 2 y2) + 2 p2 x y by the
    // usual fixed-point iteration (eight rounds; + - * / only)
    const double xd = rx, yd = ry;
    for (int it = 0; it < 8; it++) {
      const double x2 = rx * rx, y2 = ry * ry, r2 = x2 + y2, xy2 = 2.0 * rx * ry;
      const double kr = 1.0 + ((s->dist[4] * r2 + s->dist[1]) * r2 + s->dist[0]) * r2;
      const double dx = s->dist[2] * xy2 + s->dist[3] * (r2 + 2.0 * x2), dy = s->dist[2] * (r2 + 2.0 * y2) + s->dist[3] * xy2;
      rx = (xd - dx) / kr;
      ry = (yd - dy) / kr;
    }
  }
  const double wx = s->R[0] * rx + s->R[3] * ry + s->R[6] * rz;
  const double wy = s->R[1] * rx + s->R[4] * ry + s->R[7] * rz;
  const double wz = s->R[2] * rx + s->R[5] * ry + s->R[8] * rz;
  const double cx = -(s->R[0] * s->t[0] + s->R[3] * s->t[1] + s->R[6] * s->t[2]);
  const double cy = -(s->R[1] * s->t[0] + s->R[4] * s->t[1] + s->R[7] * s->t[2]);
  const double cz = -(s->R[2] * s->t[0] + s->R[5] * s->t[1] + s->R[8] * s->t[2]);
  const double denom = s->plane[0] * wx + s->plane[1] * wy + s->plane[2] * wz;
  const double num = s->plane[3] - (s->plane[0] * cx + s->plane[1] * cy + s->plane[2] * cz);
  double val = 0.0;
  if (denom > 1e-9 || denom < -1e-9) {
    const double k = num / denom;
    if (k > 0.0) {
      const double X = cx + k * wx, Y = cy + k * wy;
      if (s->texture == SDVL_TEXTURE_CAMERA) {
        double g = sdvl_camera_texture(X, Y, k / s->fx, s->seed);
        const uint32_t nz = sdvl_hash3((uint32_t)u, (uint32_t)v, s->frame_id * 0x632BE5ABu + 0x1234567u) & 7u;
        g += ((double)nz - 3.5) * 0.6;
        if (g < 0.0) g = 0.0;
        if (g > 255.0) g = 255.0;
        return (uint8_t)(g + 0.5);
      }
      const double base = 0.0116;  // ~3 px at z = 2 m, fx ~ 517
      val = 0.30 * sdvl_value_noise(X, Y, base, s->seed, 0) + 0.26 * sdvl_value_noise(X, Y, base * 2.0, s->seed, 1) +
            0.20 * sdvl_value_noise(X, Y, base * 4.0, s->seed, 2) + 0.14 * sdvl_value_noise(X, Y, base * 8.0, s->seed, 3) +
            0.10 * sdvl_value_noise(X, Y, base * 16.0, s->seed, 4);
    }
  }
  // contrast stretch around the mean (sum of 5 uniform-ish terms concentrates near 0.5)
  double g = 128.0 + (val - 0.5) * 560.0;
  const uint32_t nz = sdvl_hash3((uint32_t)u, (uint32_t)v, s->frame_id * 0x632BE5ABu + 0x1234567u) & 7u;  // 0..7
  g += ((double)nz - 3.5) * 0.6;
  if (g < 0.0) g = 0.0;
  if (g > 255.0) g = 255.0;
  return (uint8_t)(g + 0.5);
}

#endif  // SDVL_SYNTH_H_
