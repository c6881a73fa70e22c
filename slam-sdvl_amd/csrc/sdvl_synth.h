// Seeded synthetic sequence generator (SURVEY §8d "Synthetic inputs"): a textured plane n.X = d seen by a
// pinhole camera with pose T_cw.  No dataset ships with the repo, so tests, smoke() and bench.py render their
// frames with this; the same inline code runs on the host (synth_host.cc) and in a HIP kernel (sdvl_synth.hip).
// Texture = 5 octaves of bilinear value noise on lattices of 3..48 px (at the nominal depth) whose lattice
// values come from an integer hash, plus +-2 grey levels of per-frame sensor noise.  Only + - * / floor on
// doubles -> identical bytes on CPU and GPU when both are compiled with -ffp-contract=off.
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
};

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

SDVL_HD inline uint8_t sdvl_synth_pixel(const sdvl_synth_view *s, int u, int v) {
  // ray in camera coords, rotate to world: R^T * r ; camera centre C = -R^T t
  const double rx = (u - s->u0) / s->fx, ry = (v - s->v0) / s->fy, rz = 1.0;
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
