// grid_barrier_probe.hip -- what does a grid-wide barrier cost on MI355X (256 CUs in 8 XCDs)?  Decides whether a persistent
// multi-step kernel can pay for itself on the reference's own lattice sizes (32^3 ... 64^3: VERDICT r3 item 6).
//   variant 0  one counter: every workgroup adds 1, the last one bumps the generation; the others poll the generation
//   variant 1  the same plus agent-scope release/acquire fences and a data exchange: each workgroup writes a line and reads
//              its neighbour's after the barrier (checks visibility across XCDs: the L2s are not coherent with each other)
// Every spin loop is bounded: a workgroup that waits longer than ~0.2 s sets an error flag and all workgroups leave.
// build: hipcc --offload-arch=gfx950 -O3 -o build/probe/grid_barrier_probe tools/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Bar { unsigned count; unsigned pad0[31]; unsigned gen; unsigned pad1[31]; unsigned err; };

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned nwg, unsigned& my_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned target = my_gen + 1;
    const unsigned arrived = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1;
    if (arrived == nwg) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 4000000u || __hip_atomic_load(&b->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(&b->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = false;
          break;
        }
      }
    }
  }
  my_gen += 1;
  __syncthreads();
  return ok;
}

template <int VARIANT>
__global__ void __launch_bounds__(256) k_probe(Bar* b, double* data, int iters, unsigned long long* bad) {
  unsigned gen = 0;
  const unsigned nwg = gridDim.x;
  __shared__ int alive;
  if (threadIdx.x == 0) alive = 1;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 1) {
      data[(size_t)blockIdx.x * 256 + threadIdx.x] = (double)(it * 1000 + (int)blockIdx.x);
      __atomic_thread_fence(__ATOMIC_RELEASE);      // agent scope on HIP
    }
    const bool ok = grid_barrier(b, nwg, gen);
    if (threadIdx.x == 0 && !ok) alive = 0;
    __syncthreads();
    if (!alive) return;                             // every wave of every workgroup reaches an exit
    if (VARIANT == 1) {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
      const unsigned nb = (blockIdx.x + 37) % nwg;  // another XCD (workgroups are dealt round-robin over the 8 XCDs)
      const double v = __builtin_nontemporal_load(&data[(size_t)nb * 256 + threadIdx.x]);
      if (v != (double)(it * 1000 + (int)nb)) atomicAdd(bad, 1ull);
      const bool ok2 = grid_barrier(b, nwg, gen);   // nobody overwrites before everybody has read
      if (threadIdx.x == 0 && !ok2) alive = 0;
      __syncthreads();
      if (!alive) return;
    }
  }
}

int main() {
  int ncu = 0; hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  Bar* b; double* data; unsigned long long* bad;
  hipMalloc(&b, sizeof(Bar)); hipMalloc(&data, 1024 * 256 * sizeof(double)); hipMalloc(&bad, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 2; ++variant) {
    for (int nwg : {64, 128, 256, 512}) {
      if (nwg > ncu * 2) continue;                  // all workgroups must be resident (256 threads each: at least 2 fit a CU)
      hipMemset(b, 0, sizeof(Bar)); hipMemset(bad, 0, 8);
      const int iters = 2000;
      for (int rep = 0; rep < 2; ++rep) {
        hipMemset(b, 0, sizeof(Bar));
        hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(k_probe<0>, dim3(nwg), dim3(256), 0, 0, b, data, iters, bad);
        else hipLaunchKernelGGL(k_probe<1>, dim3(nwg), dim3(256), 0, 0, b, data, iters, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      Bar hb; unsigned long long hbad = 0;
      hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost); hipMemcpy(&hbad, bad, 8, hipMemcpyDeviceToHost);
      const int nbar = iters * (variant == 1 ? 2 : 1);
      printf("variant %d  %3d workgroups on %d CUs: %.3f us per barrier%s  (timed out: %u, stale reads: %llu)\n", variant, nwg, ncu,
             ms * 1e3 / nbar, variant == 1 ? " incl. fences + exchange" : "", hb.err, hbad);
    }
  }
  // for comparison: the gap between two dependent empty kernels in one stream
  hipEventRecord(e0);
  for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_probe<0>, dim3(256), dim3(256), 0, 0, b, data, 0, bad);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  printf("2000 empty 256-workgroup kernels back to back in one stream: %.3f us per launch\n", ms * 1e3 / 2000);
  return 0;
}
