// bflbm_rng.h -- counter-based Gaussian stream used for the thermal noise.
//
// The reference draws amrex::RandomNormal(0,1,engine) (LBM_binary.H:117,125,126),
// an un-vendored generator whose stream depends on the box decomposition and
// cannot be reproduced offline.  This project defines its own
// stream instead (round 2: 2650 -> ~1100 VALU instructions per site):
//   bits     one Philox4x32-10 block (KAT-checked) keyed by `seed` with counter
//            (site_lo, site_hi, noise_index, 0) seeds xoshiro128++, which supplies the site's
//            33 words (draw order: the 3 momentum modes, modes 4..18 of f, modes 4..18 of g);
//            site = x + nx*(y + ny*z) is the GLOBAL lattice index, so the noise field is
//            independent of the slab decomposition and of the GPU count;
//   normals  table-driven inverse CDF: bit 31 is the sign, the other 31 bits are the tail
//            probability t = P(|N| > x) as a binary fraction; its octave (leading zeros) and the
//            next two bits select one of 128 cubics, the next 24 bits are the argument
//            (tools/make_normal_table.py: max error 2.1e-6, variance 1 + 7e-8, <x^4> 3 - 1.3e-7;
//            tails to 6.3 sigma).  Three binary32 FMAs -- exactly rounded on the host (fmaf) and on
//            gfx950 (v_fma_f32) -- and integer operations only, so both give identical bits.
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
BFLBM_HD float bflbm_fmaf(float a, float b, float c) { return __builtin_fmaf(a, b, c); }   // one rounding on both sides
BFLBM_HD uint32_t bflbm_clz32(uint32_t v) {             // 32 for v == 0
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__clz((int)v);
#else
  return v ? (uint32_t)__builtin_clz(v) : 32u;
#endif
}
BFLBM_HD uint32_t bflbm_rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

// xoshiro128++ (Blackman & Vigna): the site's word stream
struct bflbm_rng_state { uint32_t s0, s1, s2, s3; };
BFLBM_HD uint32_t bflbm_rng_next(bflbm_rng_state& s) {
  const uint32_t result = bflbm_rotl32(s.s0 + s.s3, 7) + s.s0;
  const uint32_t t = s.s1 << 9;
  s.s2 ^= s.s0; s.s3 ^= s.s1; s.s1 ^= s.s2; s.s0 ^= s.s3;
  s.s2 ^= t;
  s.s3 = bflbm_rotl32(s.s3, 11);
  return result;
}
BFLBM_HD void bflbm_rng_seed(uint32_t seed_lo, uint32_t seed_hi, uint64_t site, uint32_t noise_index, bflbm_rng_state& s) {
  uint32_t c0 = (uint32_t)site, c1 = (uint32_t)(site >> 32), c2 = noise_index, c3 = 0u;
  bflbm_philox4x32_10(c0, c1, c2, c3, seed_lo, seed_hi);
  s.s0 = c0; s.s1 = c1; s.s2 = c2; s.s3 = c3 | 1u;     // never the all-zero state
}

#define BFLBM_NORMAL_TABLE_FLOATS 512
// standard normal from one 32-bit word; tab = the 128 cubics of bflbm_normal_table.h (LDS on the device)
template <typename TabPtr>
BFLBM_HD float bflbm_normal_from_bits(uint32_t u, TabPtr tab) {
  const uint32_t sign = u & 0x80000000u, v = u & 0x7FFFFFFFu;
  const uint32_t lz = bflbm_clz32(v);                   // 1..32: octave + 1
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t top = v << (lz & 31u);                 // v == 0: lz = 32 shifts by 0 and leaves 0 (what the hardware does anyway)
#else
  const uint32_t top = (lz >= 32u) ? 0u : (v << lz);    // leading one at bit 31
#endif
  const uint32_t r = top << 1;                          // the bits after it, left-aligned
  const uint32_t oct = (lz - 1u < 31u) ? lz - 1u : 31u; // min(lz - 1, 31)
  const uint32_t cell = oct * 4u + (r >> 30);
  const float W = (float)((r >> 6) & 0xFFFFFFu);        // 24 bits: exact
  const float c0 = tab[cell * 4u + 0u], c1 = tab[cell * 4u + 1u], c2 = tab[cell * 4u + 2u], c3 = tab[cell * 4u + 3u];
  const float x = bflbm_fmaf(bflbm_fmaf(bflbm_fmaf(c3, W, c2), W, c1), W, c0);
  return bflbm_u2f(bflbm_f2u(x) ^ sign);
}

#endif  // BFLBM_RNG_H_
