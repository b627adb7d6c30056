// bflbm_droplet.h -- droplet observables reduced on the device: raw mass moments up
// to second order (centre of mass, covariance -> principal axes) and the least-squares tanh-profile fit
// of Droplet_Fluctuation.ipynb / Surface_Tension.ipynb cell 3.  The reference's C++ twins are
// getCenterOfMass / fittingDropletCovariance / fittingDropletParams (LBM_hydrovs.H:62-335, off by default,
// main_run_job.cpp:111); the host-side numpy restatement these are tested against is analysis.py.
// Included by bflbm.hip (needs bflbm_ctx).
#ifndef BFLBM_DROPLET_H_
#define BFLBM_DROPLET_H_

namespace {

constexpr int kNMom = 20;    // {1, x, y, z, xx, xy, xz, yy, yz, zz} x {plain, trapezoid-weighted}
constexpr int kNFit = 15;    // J^T J (10), J^T e (4), e^2

template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* __restrict__ partial) {
  __shared__ double sh[NV][256];
  for (int k = 0; k < NV; ++k) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) for (int k = 0; k < NV; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x;
    for (int k = 0; k < NV; ++k) partial[b * NV + k] = sh[k][0];
  }
}

// raw moments of rho in GLOBAL cell indices; the weighted set counts the end planes of the lattice half in
// every direction (Integration::trapezoid3DWeightTensor in the reference, the notebook's `wt`)
__global__ void __launch_bounds__(256) k_moments(const double* __restrict__ rho, double* __restrict__ partial, Geo G, int p0) {
  const long long s_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = p0 + (int)blockIdx.y;
  double v[kNMom];
  for (int k = 0; k < kNMom; ++k) v[k] = 0.;
  const int y = (int)(s_ / G.pitch);
  const int x = (int)(s_ - (long long)y * G.pitch);
  if (s_ < G.plane && x < G.nx) {
    const int z = G.z0 + (p - G.H);
    const double r = rho[(long long)p * G.plane + s_];
    double w = r;
    if (x == 0 || x == G.nx - 1) w *= 0.5;
    if (y == 0 || y == G.ny - 1) w *= 0.5;
    if (z == 0 || z == G.nz - 1) w *= 0.5;
    const double m[10] = { 1., (double)x, (double)y, (double)z, (double)x * x, (double)x * y, (double)x * z,
                           (double)y * y, (double)y * z, (double)z * z };
    for (int k = 0; k < 10; ++k) { v[k] = r * m[k]; v[10 + k] = w * m[k]; }
  }
  block_sum<kNMom>(v, partial);
}

// normal equations of rho(r) = hi - (hi - lo)/2 (1 + tanh((r - R)/W)), r = |cell centre - r0| in the unit box
struct FitParams { double hi, lo, R, W, r0[3], inv_n[3]; };
__global__ void __launch_bounds__(256) k_tanhfit(const double* __restrict__ rho, double* __restrict__ partial, Geo G, int p0, FitParams F) {
  const long long s_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = p0 + (int)blockIdx.y;
  double v[kNFit];
  for (int k = 0; k < kNFit; ++k) v[k] = 0.;
  const int y = (int)(s_ / G.pitch);
  const int x = (int)(s_ - (long long)y * G.pitch);
  if (s_ < G.plane && x < G.nx) {
    const int z = G.z0 + (p - G.H);
    const double dx = (x + 0.5) * F.inv_n[0] - F.r0[0], dy = (y + 0.5) * F.inv_n[1] - F.r0[1], dz = (z + 0.5) * F.inv_n[2] - F.r0[2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    const double u = (r - F.R) / F.W;
    const double t = tanh(u);
    const double half = 0.5 * (1. + t);
    const double model = F.hi - (F.hi - F.lo) * half;
    const double e = rho[(long long)p * G.plane + s_] - model;
    const double sech2 = 1. - t * t;
    const double amp = 0.5 * (F.hi - F.lo) * sech2 / F.W;
    const double J[4] = { 1. - half, half, amp, amp * u };          // d model / d (hi, lo, R, W)
    int q = 0;
    for (int a = 0; a < 4; ++a) for (int b = a; b < 4; ++b) v[q++] = J[a] * J[b];
    for (int a = 0; a < 4; ++a) v[10 + a] = J[a] * e;
    v[14] = e * e;
  }
  block_sum<kNFit>(v, partial);
}

// sum the block partials of one launch on the host (fixed order -> deterministic)
template <int NV, class Launch>
int reduce_blocks(bflbm_ctx* c, double (&out)[NV], Launch launch) {
  if (c->step_open) return fail("reduction requested inside an open step");
  HIP_TRY(hipSetDevice(c->dom.device));
  if (ensure_density(c)) return 1;
  const dim3 g = plane_grid(c, c->nzl);
  const size_t nblocks = (size_t)g.x * g.y;
  static_assert(NV <= 20, "scratch sizing");
  double* scratch = c->S[1 - c->cur];               // far larger than nblocks*NV
  launch(g, scratch);
  HIP_TRY(hipGetLastError());
  std::vector<double> h(nblocks * NV);
  HIP_TRY(hipMemcpyAsync(h.data(), scratch, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int k = 0; k < NV; ++k) out[k] = 0.;
  for (size_t b = 0; b < nblocks; ++b) for (int k = 0; k < NV; ++k) out[k] += h[b * NV + k];
  return 0;
}

int moments_of(const std::vector<bflbm_ctx*>& ctx, double out[kNMom]) {
  for (int k = 0; k < kNMom; ++k) out[k] = 0.;
  for (bflbm_ctx* c : ctx) {
    double m[kNMom];
    if (reduce_blocks<kNMom>(c, m, [&](dim3 g, double* scratch) {
          hipLaunchKernelGGL(k_moments, g, dim3(256), 0, c->stream, c->rho, scratch, c->G, own_lo(c)); })) return 1;
    for (int k = 0; k < kNMom; ++k) out[k] += m[k];
  }
  return 0;
}

int normal_equations(const std::vector<bflbm_ctx*>& ctx, const FitParams& F, double out[kNFit]) {
  for (int k = 0; k < kNFit; ++k) out[k] = 0.;
  for (bflbm_ctx* c : ctx) {
    double m[kNFit];
    if (reduce_blocks<kNFit>(c, m, [&](dim3 g, double* scratch) {
          hipLaunchKernelGGL(k_tanhfit, g, dim3(256), 0, c->stream, c->rho, scratch, c->G, own_lo(c), F); })) return 1;
    for (int k = 0; k < kNFit; ++k) out[k] += m[k];
  }
  return 0;
}

// solve the 4x4 system (A + lam diag(A)) d = g by Gaussian elimination with partial pivoting
bool solve4(const double A[4][4], const double g[4], double lam, double d[4]) {
  double M[4][5];
  for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) M[i][j] = A[i][j] + (i == j ? lam * A[i][i] : 0.); M[i][4] = g[i]; }
  for (int c = 0; c < 4; ++c) {
    int piv = c;
    for (int r = c + 1; r < 4; ++r) if (std::fabs(M[r][c]) > std::fabs(M[piv][c])) piv = r;
    if (std::fabs(M[piv][c]) < 1e-300) return false;
    if (piv != c) for (int j = 0; j < 5; ++j) std::swap(M[piv][j], M[c][j]);
    for (int r = c + 1; r < 4; ++r) { const double f = M[r][c] / M[c][c]; for (int j = c; j < 5; ++j) M[r][j] -= f * M[c][j]; }
  }
  for (int i = 3; i >= 0; --i) { double s = M[i][4]; for (int j = i + 1; j < 4; ++j) s -= M[i][j] * d[j]; d[i] = s / M[i][i]; }
  return true;
}

// Levenberg-Marquardt on the device-reduced normal equations; p = (hi, lo, R, W) in/out
int fit_droplet(const std::vector<bflbm_ctx*>& ctx, const double r0[3], double p[4], int max_iter, double tol, double* cost_out, int* iters_out) {
  const Geo& G = ctx[0]->G;
  FitParams F;
  for (int d = 0; d < 3; ++d) F.r0[d] = r0[d];
  F.inv_n[0] = 1. / G.nx; F.inv_n[1] = 1. / G.ny; F.inv_n[2] = 1. / G.nz;
  auto eval = [&](const double q[4], double s[kNFit]) { F.hi = q[0]; F.lo = q[1]; F.R = q[2]; F.W = q[3]; return normal_equations(ctx, F, s); };
  double s[kNFit];
  if (eval(p, s)) return 1;
  double cost = s[14], lam = 1e-3;
  int it = 0;
  for (; it < max_iter; ++it) {
    double A[4][4], g[4];
    int q = 0;
    for (int a = 0; a < 4; ++a) for (int b = a; b < 4; ++b) { A[a][b] = A[b][a] = s[q++]; }
    for (int a = 0; a < 4; ++a) g[a] = s[10 + a];
    bool improved = false;
    double step_rel = 0.;
    for (int tries = 0; tries < 12 && !improved; ++tries) {
      double d[4], cand[4], s2[kNFit];
      if (!solve4(A, g, lam, d)) { lam *= 10.; continue; }
      for (int a = 0; a < 4; ++a) cand[a] = p[a] + d[a];
      if (!(cand[3] > 0.)) { lam *= 10.; continue; }            // the width stays positive
      if (eval(cand, s2)) return 1;
      if (s2[14] < cost && std::isfinite(s2[14])) {
        step_rel = 0.;
        for (int a = 0; a < 4; ++a) step_rel = std::max(step_rel, std::fabs(d[a]) / (std::fabs(cand[a]) + 1e-30));
        const double drop = (cost - s2[14]) / (cost + 1e-300);
        for (int a = 0; a < 4; ++a) p[a] = cand[a];
        for (int k = 0; k < kNFit; ++k) s[k] = s2[k];
        cost = s2[14];
        lam = std::max(lam * 0.1, 1e-12);
        improved = true;
        if (step_rel < tol && drop < tol) { it += 1; goto done; }
      } else {
        lam *= 10.;
      }
    }
    if (!improved) break;                                        // no descent direction left: converged
  }
done:
  if (cost_out) *cost_out = cost;
  if (iters_out) *iters_out = it;
  return 0;
}

}  // namespace

extern "C" {

int bflbm_droplet_moments(bflbm_ctx* c, double moments[20]) {
  if (!c || !moments) return fail("null argument");
  return moments_of({c}, moments);
}
int bflbm_ring_droplet_moments(bflbm_ring* r, double moments[20]) {
  if (!r || !moments) return fail("null argument");
  return moments_of(r->ctx, moments);
}
int bflbm_fit_droplet(bflbm_ctx* c, const double r0[3], double params[4], int max_iter, double tol, double* cost, int* iterations) {
  if (!c || !r0 || !params) return fail("null argument");
  if (!c->G.zwrap) return fail("bflbm_fit_droplet: a slab of a decomposed lattice; use bflbm_ring_fit_droplet");
  return fit_droplet({c}, r0, params, max_iter > 0 ? max_iter : 100, tol > 0 ? tol : 1e-10, cost, iterations);
}
int bflbm_ring_fit_droplet(bflbm_ring* r, const double r0[3], double params[4], int max_iter, double tol, double* cost, int* iterations) {
  if (!r || !r0 || !params) return fail("null argument");
  return fit_droplet(r->ctx, r0, params, max_iter > 0 ? max_iter : 100, tol > 0 ? tol : 1e-10, cost, iterations);
}

}  // extern "C"

#endif  // BFLBM_DROPLET_H_
