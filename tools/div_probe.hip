// Probe: d_div3/d_div9 (mul + 2 fma) against IEEE division on gfx950, 2^32 samples + edge patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../binary-fluctuating-lattice-boltzmann_amd/csrc/bflbm_site.h"

__global__ void k(unsigned long long* bad, unsigned long long seed, int rounds) {
  unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1);
  unsigned long long nb = 0;
  for (int r = 0; r < rounds; ++r) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    // random mantissa, exponent in [-60, 60], random sign; every 4th sample has a structured mantissa
    unsigned long long mant = s & 0xFFFFFFFFFFFFFull;
    if ((r & 3) == 3) mant = ((s & 1) ? (0xFFFFFFFFFFFFFull << ((s >> 58) % 52)) : ((1ull << ((s >> 58) % 52)) - 1)) & 0xFFFFFFFFFFFFFull;
    long long e = 1023 + (long long)((s >> 52) % 121) - 60;
    unsigned long long bits = ((s >> 63) << 63) | ((unsigned long long)e << 52) | mant;
    double x = __longlong_as_double((long long)bits);
    if (d_div3(x) != x / 3.0) ++nb;
    if (d_div9(x) != x / 9.0) ++nb;
    if (d_div9(x) * 0.25 != x / 36.0) ++nb;
    if (d_div3(x) * 0.0625 != x / 48.0) ++nb;
    if (d_div9(x) * 0.125 != x / 72.0) ++nb;
    // shared-reciprocal division against a/b: denominators in [2^-40, 2^40], cs4 exactly
    unsigned long long s2 = s * 0x9E3779B97F4A7C15ull + r;
    long long eb = 1023 + (long long)((s2 >> 52) % 81) - 40;
    double b = __longlong_as_double((long long)(((unsigned long long)eb << 52) | (s2 & 0xFFFFFFFFFFFFFull)));
    if (d_div(x, b, d_recip(b)) != x / b) ++nb;
    const double cs4 = (1. / 3.) * (1. / 3.);
    if (d_div(x, cs4, d_recip(cs4)) != x / cs4) ++nb;
  }
  if (nb) atomicAdd(bad, nb);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8); hipMemset(d, 0, 8);
  const int blocks = 4096, threads = 256, rounds = 4096;   // 2^32 samples
  k<<<blocks, threads>>>(d, 12345, rounds);
  unsigned long long h = 1; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("div_probe: %llu mismatches in %llu samples x 7 forms\n", h, (unsigned long long)blocks * threads * rounds);
  return h != 0;
}
