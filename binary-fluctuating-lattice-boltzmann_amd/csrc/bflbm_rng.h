// bflbm_rng.h -- counter-based Gaussian stream used for the thermal noise.
//
// The reference draws amrex::RandomNormal(0,1,engine) (LBM_binary.H:117,125,126),
// an un-vendored generator whose stream depends on the box decomposition and
// cannot be reproduced offline.  This project defines its own
// stream instead: Philox4x32-10 keyed by `seed` with counter
//   (site_lo, site_hi, noise_index, block)      block = 0..8, 4 normals each
// where site = x + nx*(y + ny*z) is the GLOBAL lattice index, so the noise field is
// independent of the slab decomposition and of the GPU count.  The Box-Muller
// transform is written with binary32 + - * / sqrt only (no libm, no FMA) so the host
// and the gfx950 kernels give identical bits.
#ifndef BFLBM_RNG_H_
#define BFLBM_RNG_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define BFLBM_HD __host__ __device__ __forceinline__
#else
#define BFLBM_HD static inline
#endif

BFLBM_HD uint32_t bflbm_mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

BFLBM_HD void bflbm_philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
#if defined(BFLBM_PHILOX_MUL64)
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
    const uint32_t hi0 = bflbm_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = bflbm_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#endif
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

BFLBM_HD float bflbm_u2f(uint32_t i) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(i);
#else
  union { uint32_t i; float f; } v; v.i = i; return v.f;
#endif
}
BFLBM_HD uint32_t bflbm_f2u(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(f);
#else
  union { uint32_t i; float f; } v; v.f = f; return v.i;
#endif
}

BFLBM_HD float bflbm_sqrtf_rn(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // __fsqrt_rn is not correctly rounded on gfx950 (measured: 15 % of inputs off by one ulp).
  // sqrt in binary64 (correctly rounded on device, verified by tools/rng_probe.hip) then one
  // rounding to binary32 is exact: 53 >= 2*24+2 bits rules out double rounding.
  return (float)sqrt((double)x);
#else
  return __builtin_sqrtf(x);
#endif
}
BFLBM_HD float bflbm_divf_rn(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __fdiv_rn(a, b);
#else
  return a / b;
#endif
}

// ln(u), u in (0,1): exponent split + atanh series on [sqrt(1/2), sqrt(2)).
BFLBM_HD float bflbm_logf(float u) {
  uint32_t bits = bflbm_f2u(u);
  int e = (int)(bits >> 23) - 127;
  float m = bflbm_u2f((bits & 0x007FFFFFu) | 0x3F800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  const float t = m - 1.0f;
  const float s = bflbm_divf_rn(t, 2.0f + t);
  const float z = s * s;
  float p = 0.11111111f;
  p = p * z + 0.14285714f;
  p = p * z + 0.2f;
  p = p * z + 0.33333333f;
  p = p * z + 1.0f;
  const float lnm = (2.0f * s) * p;
  return (float)e * 0.69314718f + lnm;
}

// sin, cos of 2*pi*k/2^24
BFLBM_HD void bflbm_sincos2pi(uint32_t k, float& sn, float& cs) {
  const uint32_t q = k >> 22;
  uint32_t r = k & 0x3FFFFFu;
  const bool swap = r > 0x200000u;
  if (swap) r = 0x400000u - r;
  const float x = (float)r * 3.7450703e-07f;
  const float x2 = x * x;
  float ps = -1.9841270e-04f;
  ps = ps * x2 + 8.3333333e-03f;
  ps = ps * x2 - 1.6666667e-01f;
  ps = ps * x2 + 1.0f;
  float s = x * ps;
  float pc = 2.4801587e-05f;
  pc = pc * x2 - 1.3888889e-03f;
  pc = pc * x2 + 4.1666667e-02f;
  pc = pc * x2 - 0.5f;
  pc = pc * x2 + 1.0f;
  float c = pc;
  if (swap) { const float t = s; s = c; c = t; }
  sn = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
  cs = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}

BFLBM_HD void bflbm_box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u = ((float)(a >> 9) + 0.5f) * 1.1920929e-07f;
  const float r = bflbm_sqrtf_rn(-2.0f * bflbm_logf(u));
  float s, c;
  bflbm_sincos2pi(b >> 8, s, c);
  n0 = r * c;
  n1 = r * s;
}

// Four standard normals of (site, noise_index, block).
BFLBM_HD void bflbm_rng_block(uint32_t seed_lo, uint32_t seed_hi, uint64_t site, uint32_t noise_index,
                              uint32_t blk, float& a, float& b, float& c, float& d) {
  uint32_t c0 = (uint32_t)site, c1 = (uint32_t)(site >> 32), c2 = noise_index, c3 = blk;
  bflbm_philox4x32_10(c0, c1, c2, c3, seed_lo, seed_hi);
  bflbm_box_muller(c0, c1, a, b);
  bflbm_box_muller(c2, c3, c, d);
}

#endif  // BFLBM_RNG_H_
