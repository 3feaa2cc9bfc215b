// kmc_synth.h -- deterministic synthetic Velodyne-like point generator, shared by host and device.
//
// Measurement infrastructure for BASELINE.json configs 2-5 (no KITTI drive can be shipped): 64 rings,
// azimuth sweeping the full circle once per ring in scan order (frac 0 -> 1), elevation ring-linear
// -24.8 .. +2.0 deg, range U[2,80) m, intensity on a 0.01 grid.  Every value is a pure function of
// (seed, point index) through a splitmix64 counter hash, built ONLY from IEEE +,-,*,/ and fma in a fixed
// order (no libm/ocml calls, contraction off), so the host and the gfx950 device produce bit-identical
// floats -- the tests regenerate a frame on the host to feed the CPU oracle instead of copying it back.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KMC_HD __host__ __device__ __forceinline__
#else
#define KMC_HD static inline
#endif

#include <math.h>

namespace kmc_synth {

constexpr uint32_t kRings = 64;

KMC_HD uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// cos / sin of an angle given in TURNS (v in [-0.5, 0.5]); quadrant reduction + Taylor on [-pi/4, pi/4].
KMC_HD void sincos_turns(float v, float* s_out, float* c_out) {
#pragma clang fp contract(off)
  const float jf = floorf(v * 4.0f + 0.5f);          // nearest quadrant, exact arithmetic on small ints
  const float r = fmaf(-0.25f, jf, v);               // [-1/8, 1/8] turns
  const float th = r * 6.283185307179586f;           // radians
  const float u = th * th;
  float sp = 2.7557319e-06f;                          // 1/9!
  sp = fmaf(sp, u, -1.9841270e-04f);
  sp = fmaf(sp, u, 8.3333333e-03f);
  sp = fmaf(sp, u, -1.6666667e-01f);
  sp = fmaf(sp, u, 1.0f);
  const float sn = sp * th;
  float cp = -2.7557319e-07f;                         // -1/10!
  cp = fmaf(cp, u, 2.4801587e-05f);
  cp = fmaf(cp, u, -1.3888889e-03f);
  cp = fmaf(cp, u, 4.1666667e-02f);
  cp = fmaf(cp, u, -0.5f);
  cp = fmaf(cp, u, 1.0f);
  const int j = ((int)jf) & 3;
  float s, c;
  if (j == 0) { s = sn; c = cp; }
  else if (j == 1) { s = cp; c = -sn; }
  else if (j == 2) { s = -sn; c = -cp; }
  else { s = -cp; c = sn; }
  *s_out = s;
  *c_out = c;
}

struct Point { float x, y, z, i; };

// point `idx` of an n-point frame
KMC_HD Point make_point(uint64_t idx, uint64_t n, uint64_t seed) {
#pragma clang fp contract(off)
  const uint64_t steps = (n + kRings - 1) / kRings;  // azimuth steps per ring
  const uint64_t ring = idx / steps;
  const uint64_t k = idx - ring * steps;
  const uint64_t h = splitmix64(seed ^ (idx * 0xD1342543DE82EF95ull));
  const float u_range = (float)(uint32_t)(h >> 40) * 5.9604645e-08f;        // 24 bits -> [0,1)
  const float u_jit = (float)(uint32_t)((h >> 16) & 0xFFFFFFu) * 5.9604645e-08f;
  const float inten = (float)(uint32_t)((h & 0xFFFFu) % 100u) * 0.01f;
  const float frac = ((float)k + u_jit) / (float)steps;                      // fraction of scan, [0,1)
  const float az = 0.5f - frac;                                              // azimuth in turns, (-0.5, 0.5]
  const float el = (-24.8f + (float)ring * (26.8f / 63.0f)) * (1.0f / 360.0f);
  float sa, ca, se, ce;
  sincos_turns(az, &sa, &ca);
  sincos_turns(el, &se, &ce);
  const float range = 2.0f + 78.0f * u_range;
  const float rxy = range * ce;
  Point p;
  p.x = rxy * ca;
  p.y = rxy * sa;
  p.z = range * se;
  p.i = inten;
  return p;
}

}  // namespace kmc_synth
