// bflbm_rng.h -- counter-based Gaussian stream used for the thermal noise.
//
// The reference draws amrex::RandomNormal(0,1,engine) (LBM_binary.H:117,125,126),
// an un-vendored generator whose stream depends on the box decomposition and
// cannot be reproduced offline.  This project defines its own
// stream instead (round 1: ~2650 VALU instructions per site, rounds 2-3: ~1170, round 4: ~520):
//   bits     one Philox4x32-10 block (KAT-checked) keyed by `seed` with counter
//            (site_lo, site_hi, noise_index, 0) seeds xoshiro128+, which supplies the site's
//            33 words (draw order: the 3 momentum modes, modes 4..18 of f, modes 4..18 of g);
//            site = x + nx*(y + ny*z) is the GLOBAL lattice index, so the noise field is
//            independent of the slab decomposition and of the GPU count;
//   normals  quantile table: s = sum of the four bytes of the word (0..1020, one v_dot4_u32_u8), x = T[s]
//            (one 8-byte LDS read).  s has the exactly known bell-shaped distribution of a sum of four
//            uniform bytes; T carries its cumulative cells to the normal quantiles between the same
//            probabilities (conditional means, unit variance: tools/make_normal_table.py -- 1021 levels,
//            0.0065 sigma apart in the centre, <x^4> = 3 - 7e-6, outermost level 6.38 sigma).  No floating-
//            point arithmetic is involved, so host and device agree by construction.
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

#ifndef BFLBM_PHILOX_ROUNDS
#define BFLBM_PHILOX_ROUNDS 10     // anything else is a timing experiment (tools/), never the product
#endif
BFLBM_HD void bflbm_philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < BFLBM_PHILOX_ROUNDS; ++r) {
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

BFLBM_HD uint32_t bflbm_rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

// xoshiro128+ (Blackman & Vigna): the site's word stream.  The plain sum is the cheapest of the family's output
// functions (8 integer instructions per word against 10 for ++); its known weakness, low linear complexity of the
// lowest bits, is immaterial for a word that is consumed as the sum of its bytes.
struct bflbm_rng_state { uint32_t s0, s1, s2, s3; };
BFLBM_HD uint32_t bflbm_rng_next(bflbm_rng_state& s) {
  const uint32_t result = s.s0 + s.s3;
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

#define BFLBM_NORMAL_TABLE_N 1024      // doubles: levels 0..1020 and three entries of padding (8 KB)
// byte offset of the word's level in the table: 8 * (b0 + b1 + b2 + b3)
BFLBM_HD uint32_t bflbm_level_offset(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_udot4(u, 0x08080808u, 0u, false);
#else
  return 8u * ((u & 255u) + ((u >> 8) & 255u) + ((u >> 16) & 255u) + (u >> 24));
#endif
}
// standard normal from one 32-bit word; tab = the table of bflbm_normal_table.h (LDS on the device)
template <typename TabPtr>
BFLBM_HD double bflbm_normal_from_bits(uint32_t u, TabPtr tab) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(&tab[0]) + bflbm_level_offset(u));
}

#endif  // BFLBM_RNG_H_
