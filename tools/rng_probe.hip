// Probe: do the primitives of csrc/bflbm_rng.h give the same bits on the host and on gfx950?
// normals of 2^24 random words (v_dot4_u32_u8 byte sum on the device, shifts and adds on the host; one table read), the
// xoshiro128+/Philox stream of 2^24 sites, and the correctly rounded binary64 sqrt and division the amplitudes rely on.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/probe/rng_probe tools/rng_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include "../binary-fluctuating-lattice-boltzmann_amd/csrc/bflbm_rng.h"
#include "../binary-fluctuating-lattice-boltzmann_amd/csrc/bflbm_normal_table.h"

__constant__ double tab_dev[BFLBM_NORMAL_TABLE_N] = BFLBM_NORMAL_TABLE_VALUES;
static const double tab_host[BFLBM_NORMAL_TABLE_N] = BFLBM_NORMAL_TABLE_VALUES;

struct Out { double nrm; double site[33]; double sq64, dv64; };

template <typename Tab>
__host__ __device__ inline void eval(uint32_t a, uint32_t b, Tab tab, Out& o) {
  o.nrm = bflbm_normal_from_bits(a, tab);
  bflbm_rng_state st;
  bflbm_rng_seed(12345u, 0u, ((uint64_t)b << 8) | (a & 255u), a >> 20, st);
  for (int k = 0; k < 33; ++k) o.site[k] = bflbm_normal_from_bits(bflbm_rng_next(st), tab);
  const double x = (double)a * 1e-3 + 1e-9, y = (double)b + 3.0;
  o.sq64 = sqrt(x);
  o.dv64 = x / y;
}
__global__ void k(const uint32_t* a, const uint32_t* b, Out* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) eval(a[i], b[i], tab_dev, o[i]);
}
int main() {
  const int n = 1 << 20, rounds = 16;
  long bad[4] = {0, 0, 0, 0};
  uint64_t s = 88172645463325252ull;
  uint32_t *da, *db; Out* dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, n * sizeof(Out));
  std::vector<uint32_t> a(n), b(n);
  std::vector<Out> d(n);
  for (int r = 0; r < rounds; ++r) {
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (uint32_t)s; b[i] = (uint32_t)(s >> 32); }
    if (r == 0) { a[0] = 0u; a[1] = 0x80000000u; a[2] = 1u; a[3] = 0x7fffffffu; a[4] = 0xffffffffu; a[5] = 0x40000000u; }   // edge words
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dout, n);
    hipMemcpy(d.data(), dout, n * sizeof(Out), hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
      Out h; eval(a[i], b[i], tab_host, h);
      bad[0] += h.nrm != d[i].nrm;
      for (int q = 0; q < 33; ++q) bad[1] += h.site[q] != d[i].site[q];
      bad[2] += h.sq64 != d[i].sq64; bad[3] += h.dv64 != d[i].dv64;
    }
  }
  printf("mismatches host vs gfx950 over %d words: normal %ld, site streams %ld, sqrt64 %ld, div64 %ld\n", n * rounds, bad[0], bad[1], bad[2], bad[3]);
  return (bad[0] | bad[1] | bad[2] | bad[3]) != 0;
}
