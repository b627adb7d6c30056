// Probe: which binary32 primitive of bflbm_rng.h differs between host and gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
#include "../binary-fluctuating-lattice-boltzmann_amd/csrc/bflbm_rng.h"

struct Out { float lg, sq, dv, sn, cs, n0, n1, u, sqb; double sq64, dv64; };

__host__ __device__ inline void eval(uint32_t a, uint32_t b, Out& o) {
  const float u = ((float)(a >> 9) + 0.5f) * 1.1920929e-07f;
  o.u = u;
  o.lg = bflbm_logf(u);
  o.sq = bflbm_sqrtf_rn(-2.0f * o.lg);
  o.sqb = __builtin_sqrtf(-2.0f * o.lg + (float)(b & 1023) * 1.0e-3f);
  o.dv = bflbm_divf_rn(u, 2.0f + u);
  bflbm_sincos2pi(b >> 8, o.sn, o.cs);
  bflbm_box_muller(a, b, o.n0, o.n1);
  double x = (double)a * 1e-3 + 1e-9, y = (double)b + 3.0;
  o.sq64 = sqrt(x);
  o.dv64 = x / y;
}
__global__ void k(const uint32_t* a, const uint32_t* b, Out* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) eval(a[i], b[i], o[i]);
}
int main() {
  const int n = 1 << 20;
  std::vector<uint32_t> a(n), b(n);
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; a[i] = (uint32_t)s; b[i] = (uint32_t)(s >> 32); }
  uint32_t *da, *db; Out* dout;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, n * sizeof(Out));
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, db, dout, n);
  std::vector<Out> d(n);
  hipMemcpy(d.data(), dout, n * sizeof(Out), hipMemcpyDeviceToHost);
  long bad[11] = {0};
  for (int i = 0; i < n; ++i) {
    Out h; eval(a[i], b[i], h);
    bad[0] += h.u != d[i].u; bad[1] += h.lg != d[i].lg; bad[2] += h.sq != d[i].sq; bad[3] += h.dv != d[i].dv;
    bad[4] += h.sn != d[i].sn; bad[5] += h.cs != d[i].cs; bad[6] += h.n0 != d[i].n0; bad[7] += h.n1 != d[i].n1;
    bad[8] += h.sq64 != d[i].sq64; bad[9] += h.dv64 != d[i].dv64; bad[10] += h.sqb != d[i].sqb;
  }
  printf("mismatches of %d: u %ld log %ld sqrt %ld div %ld sin %ld cos %ld n0 %ld n1 %ld sqrt64 %ld div64 %ld builtin_sqrtf %ld\n",
         n, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7], bad[8], bad[9], bad[10]);
  return 0;
}
