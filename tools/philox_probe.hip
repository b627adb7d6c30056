// philox_probe.hip -- issue cost of the two ways to form the 32x32->64-bit products of a Philox4x32 round on
// gfx950: v_mul_lo_u32 + v_mul_hi_u32 versus one v_mad_u64_u32.  Prints ns per Philox4x32-10 block per lane-wave.
// build: hipcc --offload-arch=gfx950 -O3 -o build/probe/philox_probe tools/philox_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int VARIANT>
__device__ __forceinline__ void philox(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    if (VARIANT == 0) {
      hi0 = __umulhi(0xD2511F53u, c0); lo0 = 0xD2511F53u * c0;
      hi1 = __umulhi(0xCD9E8D57u, c2); lo1 = 0xCD9E8D57u * c2;
    } else {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
    }
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k(uint32_t* out, int blocks) {
  uint32_t acc = 0;
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int b = 0; b < blocks; ++b) {
    uint32_t c0 = tid, c1 = b, c2 = 7, c3 = 1;
    philox<VARIANT>(c0, c1, c2, c3, 12345u, 0u);
    acc ^= c0 ^ c1 ^ c2 ^ c3;
  }
  out[tid] = acc;
}

int main() {
  const int nwg = 256 * 8, blocks = 512;
  uint32_t* d; hipMalloc(&d, nwg * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t h[2][4];
  for (int v = 0; v < 2; ++v) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 0, 0, d, blocks);
      else        hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 0, 0, d, blocks);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(h[v], d, 16, hipMemcpyDeviceToHost);
    // waves per SIMD: nwg*4 waves / (256 CUs * 4 SIMDs) = 8; cycles per block per wave at 2.4 GHz
    const double waves_per_simd = nwg * 4.0 / (256 * 4);
    printf("variant %d (%s): %.3f ms, %.0f cycles per Philox4x32-10 block per wave (2.4 GHz)\n", v,
           v == 0 ? "v_mul_lo_u32 + v_mul_hi_u32" : "64-bit product", best, best * 1e-3 * 2.4e9 / (blocks * waves_per_simd));
  }
  printf("same output: %d\n", (int)(h[0][0] == h[1][0] && h[0][1] == h[1][1]));
  return 0;
}
