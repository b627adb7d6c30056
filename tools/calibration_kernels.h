// Calibration-only kernels (not part of the product library): pull-copy variants used to measure the practical
// bandwidth ceiling of the access pattern (profiles/r01_calibrate_*.json, NOTES.md section 3.1).  Build them into a
// diagnostic library with  make CXXFLAGS+=' -DBFLBM_CALIBRATION'  (csrc/bflbm.hip includes this file then).
// calibration only: the same pull-copy with two x-adjacent sites per thread (16-byte accesses where aligned)
__global__ void __launch_bounds__(256) k_pull2(const double* __restrict__ S, double* __restrict__ N, Geo G, int p0) {
  const long long s_ = ((long long)blockIdx.x*blockDim.x + threadIdx.x) * 2;
  if (s_ >= G.plane) return;
  const int p = p0 + (int)blockIdx.y;
  const int y = (int)(s_ / G.nx);
  const int x = (int)(s_ - (long long)y*G.nx);
  SiteIdx I; site_index(G, x, y, p, I);
  const long long o = I.row[1][1] + x;
#pragma unroll
  for (int i = 0; i < 2*Q; ++i) {
    const int k = i % Q;
    const long long r = I.row[1 - Vel::cz[k]][1 - Vel::cy[k]];
    const double* __restrict__ src = S + (long long)i*G.vol + r;
    double2 v;
    if (Vel::cx[k] == 0) v = *reinterpret_cast<const double2*>(src + x);
    else if (Vel::cx[k] > 0) { v.x = src[I.xm]; v.y = src[x]; }
    else { v.x = src[x + 1]; v.y = src[(x + 2 >= G.nx) ? x + 2 - G.nx : x + 2]; }
    *reinterpret_cast<double2*>(N + (long long)i*G.vol + o) = v;
  }
}

// calibration only: the pull-copy on a row-interleaved layout [p][y][c][x] (component rows of one lattice
// row adjacent) -- same bytes, same coalescing, different DRAM stream structure (single slab only)
__global__ void __launch_bounds__(256) k_pull_rows(const double* __restrict__ S, double* __restrict__ N, Geo G, int p0) {
  BFLBM_SITE_FROM_BLOCK();
  const int xm = x == 0 ? G.nx - 1 : x - 1, xp = x == G.nx - 1 ? 0 : x + 1;
  const int ym = y == 0 ? G.ny - 1 : y - 1, yp = y == G.ny - 1 ? 0 : y + 1;
  const int pm = p == 0 ? G.nzs - 1 : p - 1, pp = p == G.nzs - 1 ? 0 : p + 1;
  const int xs[3] = { xm, x, xp }, ys[3] = { ym, y, yp }, ps[3] = { pm, p, pp };
#pragma unroll
  for (int i = 0; i < 2*Q; ++i) {
    const int k = i % Q;
    const long long src = (((long long)ps[1 - Vel::cz[k]]*G.ny + ys[1 - Vel::cy[k]])*(2*Q) + i)*G.nx + xs[1 - Vel::cx[k]];
    N[(((long long)p*G.ny + y)*(2*Q) + i)*G.nx + x] = S[src];
  }
}

