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

// second stage on the device: one workgroup per plane adds that plane's block partials (fixed order: every thread a strided
// subsequence, then the tree of block_sum), so that a reduction hands O(nz) doubles to the host instead of one per block --
// the gradient-flow fit runs ~400 such reductions per fit, at 512^3 that was 8 MB per reduction
template <int NV>
__global__ void __launch_bounds__(256) k_sum_partials(const double* __restrict__ partial, double* __restrict__ out, int nbx) {
  __shared__ double sh[NV][256];
  double v[NV];
  for (int k = 0; k < NV; ++k) v[k] = 0.;
  const long long base = (long long)blockIdx.x * nbx;
  for (int b = threadIdx.x; b < nbx; b += 256) for (int k = 0; k < NV; ++k) v[k] += partial[(base + b) * NV + k];
  for (int k = 0; k < NV; ++k) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) for (int k = 0; k < NV; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) for (int k = 0; k < NV; ++k) out[(long long)blockIdx.x * NV + k] = sh[k][0];
}

// block partials of one launch -> per-plane sums on the device -> ADDED to `out` plane by plane on the host.  The caller zeroes
// `out` once and passes the slabs in order, so the sum runs over the planes of the whole lattice in one fixed sequence whatever
// the decomposition: a ring gives the same doubles as the single context.
template <int NV, class Launch>
int reduce_blocks(bflbm_ctx* c, double (&out)[NV], Launch launch) {
  if (c->step_open) return fail("reduction requested inside an open step");
  HIP_TRY(hipSetDevice(c->dom.device));
  if (ensure_density(c)) return 1;
  const dim3 g = plane_grid(c, c->nzl);
  const size_t nblocks = (size_t)g.x * g.y;
  static_assert(NV <= 20, "scratch sizing");
  double* scratch = c->S[1 - c->cur];               // far larger than (nblocks + planes)*NV
  double* planes = scratch + nblocks * NV;
  launch(g, scratch);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL((k_sum_partials<NV>), dim3(g.y), dim3(256), 0, c->stream, scratch, planes, (int)g.x);
  HIP_TRY(hipGetLastError());
  std::vector<double> h((size_t)g.y * NV);
  HIP_TRY(hipMemcpyAsync(h.data(), planes, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (size_t b = 0; b < (size_t)g.y; ++b) for (int k = 0; k < NV; ++k) out[k] += h[b * NV + k];
  return 0;
}

int moments_of(const std::vector<bflbm_ctx*>& ctx, double out[kNMom]) {
  double m[kNMom];
  for (int k = 0; k < kNMom; ++k) m[k] = 0.;
  for (bflbm_ctx* c : ctx)
    if (reduce_blocks<kNMom>(c, m, [&](dim3 g, double* scratch) {
          hipLaunchKernelGGL(k_moments, g, dim3(256), 0, c->stream, c->rho, scratch, c->G, own_lo(c)); })) return 1;
  for (int k = 0; k < kNMom; ++k) out[k] = m[k];
  return 0;
}

int normal_equations(const std::vector<bflbm_ctx*>& ctx, const FitParams& F, double out[kNFit]) {
  double m[kNFit];
  for (int k = 0; k < kNFit; ++k) m[k] = 0.;
  for (bflbm_ctx* c : ctx)
    if (reduce_blocks<kNFit>(c, m, [&](dim3 g, double* scratch) {
          hipLaunchKernelGGL(k_tanhfit, g, dim3(256), 0, c->stream, c->rho, scratch, c->G, own_lo(c), F); })) return 1;
  for (int k = 0; k < kNFit; ++k) out[k] = m[k];
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

// ---------------------------------------------------------------------------------------------------------------------
// The reference's own radius fit (fittingDroplet / fittingDropletParams, LBM_hydrovs.H:117-213; paramsVariations and
// its coefficient functions, externlib.H:25-406; off by default, main_run_job.cpp:111, :364-367): a semi-implicit
// gradient flow of  F(W, R) = int (rho - 1/2 (1 + tanh((R - |r - r0|) / sqrt(2W))))^2 dV  in the unit box.  Per step
//     (dW, dR) = 1/det D . A . B . (Mf_W - K_W / 2,  Mf_R - K_R / 2),   A = [[1 - J_RR, J_WR], [J_RW, 1 - J_WW]],
//     B = diag(-eta_W dt, eta_R dt),  det D = (1 - J_WW)(1 - J_RR) - J_WR J_RW
// where Mf_* are lattice integrals of rho against the profile's parameter derivatives (the device reductions below),
// K_* the same integrals of the profile itself and J_* the linearisation, both closed forms in c = R / sqrt(2W):
//     I_n(c) = int_{-c}^{inf} (x + c)^n sech^4(x) dx ,  n = 2, 3, 4            (integral_func2_series, d = delta = 1)
// evaluated like the reference does: |x| < 1 from the Taylor series of sech^4 (20 terms, getCoefS), the tails from
// sech^4(y) = 16/6 sum_k (-1)^k (k+1)(k+2)(k+3) e^{-(2k+4) y} integrated against the polynomial term by term.
// No output of this fit is recorded anywhere in the reference: parity unpinned; tests/test_flowfit.py checks the closed
// forms against numerical quadrature and the fit against the notebooks' least-squares radii.

constexpr int kFlowTerms = 20;                 // NumOfTerms, externlib.H:23
struct FlowSeries { double S[kFlowTerms]; };   // Taylor coefficients of sech^4: sech^4(x) = sum_k S_k x^(2k)

inline double binom(int n, int k) {            // comb(), externlib.H:35-44
  if (k > n) return 0.;
  if (k == 0 || k == n) return 1.;
  double r = 1.;
  for (int i = 1; i <= k; ++i) r *= (double)(n - i + 1) / i;
  return r;
}
// getCoefS (externlib.H:57-91): Euler numbers A_k by sum_{j<=k} A_j C(2k,2j) = 0, sech(x) = sum A_k x^(2k) / (2k)!,
// and the fourth power of that series by a four-fold Cauchy product
inline FlowSeries flow_series() {
  double A[kFlowTerms], a[kFlowTerms];
  A[0] = 1.;
  for (int k = 1; k < kFlowTerms; ++k) { double t = 0.; for (int j = 0; j < k; ++j) t += A[j] * binom(2 * k, 2 * j); A[k] = -t; }
  FlowSeries F;
  for (int k = 0; k < kFlowTerms; ++k) {
    double fact = 1.; for (int i = 1; i <= 2 * k; ++i) fact *= i;
    a[k] = A[k] / fact;
    double sk = 0.;
    for (int k1 = 0; k1 <= k; ++k1) for (int k2 = 0; k2 <= k - k1; ++k2) for (int k3 = 0; k3 <= k - k1 - k2; ++k3)
      sk += a[k1] * a[k2] * a[k3] * a[k - k1 - k2 - k3];
    F.S[k] = sk;
  }
  return F;
}
// I_n(c), n in {2, 3, 4} (integral_func2_series with d = 1, delta = 1/d, externlib.H:108-157; extended precision like there)
inline double flow_In(int n, double c, const FlowSeries& F) {
  typedef long double ld;
  const ld delta = 1.0L, q = (ld)c;            // q = c / d
  ld sum = 0.0L;
  for (int k = 0; k < kFlowTerms; ++k) {
    const ld a = (ld)(2 * k + 4), i1 = 1.0L / a, i2 = i1 * i1, i3 = i2 * i1, i4 = i3 * i1, i5 = i4 * i1;
    const ld lo = q - delta, hi = q + delta, ed = expl(-a * delta), ec = expl(-a * q);
    ld left, right;                            // int_{-c}^{-delta} and int_{delta}^{inf} of (x + c)^n e^{-a |x|}
    if (n == 4) {
      left = (i1 * lo * lo * lo * lo - 4 * i2 * lo * lo * lo + 12 * i3 * lo * lo - 24 * i4 * lo + 24 * i5) * ed - 24 * i5 * ec;
      right = (i1 * hi * hi * hi * hi + 4 * i2 * hi * hi * hi + 12 * i3 * hi * hi + 24 * i4 * hi + 24 * i5) * ed;
    } else if (n == 3) {
      left = (i1 * lo * lo * lo - 3 * i2 * lo * lo + 6 * i3 * lo - 6 * i4) * ed + 6 * i4 * ec;
      right = (i1 * hi * hi * hi + 3 * i2 * hi * hi + 6 * i3 * hi + 6 * i4) * ed;
    } else {
      left = (i1 * lo * lo - 2 * i2 * lo + 2 * i3) * ed - 2 * i3 * ec;
      right = (i1 * hi * hi + 2 * i2 * hi + 2 * i3) * ed;
    }
    const ld w = (16.0L / 6.0L) * (k + 1) * (k + 2) * (k + 3) * (left + right);
    sum += (k % 2 == 0) ? w : -w;
    ld mid = 0.0L;                             // int_{-delta}^{delta} (x + c)^n x^(2k) dx by the binomial theorem
    for (int l = 0; l <= n; ++l) {
      const int e = 2 * k + l + 1;
      mid += (ld)binom(n, l) * powl(q, n - l) * (powl(delta, e) - powl(-delta, e)) / e;
    }
    sum += (ld)F.S[k] * mid;
  }
  return (double)sum;
}
// integral_func3_series (externlib.H:159-173) and integral_func1_series (:175-193)
inline double flow_f3(int n, double c) {
  double v = 0.;
  for (int k = 1; k <= 50; ++k) {
    const double k2 = (double)k * k, sg = (k % 2) ? 1. : -1.;      // (-1)^(k+1)
    if (n == 3) v += 6. * sg * (c / k2 + 0.25 / (k2 * k) * exp(-2. * k * c));
    else        v += -sg * exp(-2. * k * c) / k2 + sg * 2. / k2;
  }
  return v + 2. * pow(c, n);
}
inline double flow_f1(int n, double a) {
  if (n != 3) return -a - log(2.) - log(cosh(a));
  double s1 = 0., s2 = 0.;
  for (int k = 1; k <= 100; ++k) { const double k2 = (double)k * k, sg = (k % 2) ? 1. : -1.; s1 += sg / k2 * exp(-2. * k * a); s2 += sg / k2; }
  return 1.5 * s1 - 3. * s2 - 3. * a * a;
}
// J_RR, J_WR, J_RW, J_WW (JRn_Rn, JWn_Rn, JRn_Wn, JWn_Wn, externlib.H:199-253) and K_W, K_R (KWn, KRn, :344-371)
struct FlowCoef { double Jrr, Jwr, Jrw, Jww, Kw, Kr; };
inline FlowCoef flow_coefficients(double W, double R, double eta_W, double eta_R, double dt, double C0, const FlowSeries& F) {
  const double s2w = sqrt(2. * W), c = R / s2w, pi = M_PI;
  const double I2 = flow_In(2, c, F), I3 = flow_In(3, c, F), I4 = flow_In(4, c, F);
  FlowCoef o;
  o.Jrr = -C0 * eta_R * dt * s2w * pi * I2;
  o.Jrw = C0 * 0.25 * eta_R * dt * pi / (W * W) * (R * 2. * W * s2w * I2 - 4. * W * W * I3);
  o.Jwr = C0 * 0.25 * eta_W * dt * (2. * sqrt(2.) * pi * R / sqrt(W) * I2 - 4. * pi * I3);
  o.Jww = -C0 * 0.125 * eta_W * dt * pi / (W * W * W) * (s2w * s2w * s2w * R * R * I2 + pow(s2w, 5.) * I4 - 2. * R * pow(s2w, 4.) * I3);
  const double s2w3 = s2w * s2w * s2w;
  o.Kw = sqrt(2.) * pi / pow(sqrt(W), 3.) * (R * s2w3 * flow_f3(2, c) - 4. * W * W * flow_f3(3, c) + R * s2w3 * flow_f1(2, c) - 4. * W * W * flow_f1(3, c));
  o.Kr = 4. * pi * 2. * W * (flow_f3(2, c) + flow_f1(2, c));
  return o;
}

// Mf_W, Mf_R (MfWn, MfRn, externlib.H:255-342): sum over cell centres of rho dist sech^2(dist / sqrt(2W)) and rho sech^2(...)
struct FlowParams { double R, inv_s2w, r0[3], inv_n[3]; };
__global__ void __launch_bounds__(256) k_flowfit(const double* __restrict__ rho, double* __restrict__ partial, Geo G, int p0, FlowParams F) {
  const long long s_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = p0 + (int)blockIdx.y;
  double v[2] = {0., 0.};
  const int y = (int)(s_ / G.pitch);
  const int x = (int)(s_ - (long long)y * G.pitch);
  if (s_ < G.plane && x < G.nx) {
    const int z = G.z0 + (p - G.H);
    const double dx = (x + 0.5) * F.inv_n[0] - F.r0[0], dy = (y + 0.5) * F.inv_n[1] - F.r0[1], dz = (z + 0.5) * F.inv_n[2] - F.r0[2];
    const double dist = F.R - sqrt(dx * dx + dy * dy + dz * dz);
    const double u = dist * F.inv_s2w;
    const double sech = fabs(u) < 710.4 ? 1. / cosh(u) : 0.;       // inv_acosh, externlib.H:25-32
    const double r = rho[(long long)p * G.plane + s_];
    v[0] = r * (dist * sech * sech);
    v[1] = r * (sech * sech);
  }
  block_sum<2>(v, partial);
}
// per-block minimum and maximum of rho (C0 = max - min, LBM_hydrovs.H:128-129)
__global__ void __launch_bounds__(256) k_minmax(const double* __restrict__ rho, double* __restrict__ partial, Geo G, int p0) {
  __shared__ double lo[256], hi[256];
  const long long s_ = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int p = p0 + (int)blockIdx.y;
  const int y = (int)(s_ / G.pitch);
  const int x = (int)(s_ - (long long)y * G.pitch);
  const bool in = s_ < G.plane && x < G.nx;
  const double r = in ? rho[(long long)p * G.plane + s_] : 0.;
  lo[threadIdx.x] = in ? r : 1e300; hi[threadIdx.x] = in ? r : -1e300;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { lo[threadIdx.x] = fmin(lo[threadIdx.x], lo[threadIdx.x + w]); hi[threadIdx.x] = fmax(hi[threadIdx.x], hi[threadIdx.x + w]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { const long long b = (long long)blockIdx.y * gridDim.x + blockIdx.x; partial[2 * b] = lo[0]; partial[2 * b + 1] = hi[0]; }
}
// second stage of k_minmax: one workgroup per plane
__global__ void __launch_bounds__(256) k_minmax_partials(const double* __restrict__ partial, double* __restrict__ out, int nbx) {
  __shared__ double lo[256], hi[256];
  double l = 1e300, h = -1e300;
  const long long base = (long long)blockIdx.x * nbx;
  for (int b = threadIdx.x; b < nbx; b += 256) { l = fmin(l, partial[2 * (base + b)]); h = fmax(h, partial[2 * (base + b) + 1]); }
  lo[threadIdx.x] = l; hi[threadIdx.x] = h;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { lo[threadIdx.x] = fmin(lo[threadIdx.x], lo[threadIdx.x + w]); hi[threadIdx.x] = fmax(hi[threadIdx.x], hi[threadIdx.x + w]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = lo[0]; out[2 * blockIdx.x + 1] = hi[0]; }
}
int rho_range(const std::vector<bflbm_ctx*>& ctx, double& lo, double& hi) {
  lo = 1e300; hi = -1e300;
  for (bflbm_ctx* c : ctx) {
    if (c->step_open) return fail("reduction requested inside an open step");
    HIP_TRY(hipSetDevice(c->dom.device));
    if (ensure_density(c)) return 1;
    const dim3 g = plane_grid(c, c->nzl);
    const size_t nb = (size_t)g.x * g.y;
    double* scratch = c->S[1 - c->cur];
    hipLaunchKernelGGL(k_minmax, g, dim3(256), 0, c->stream, c->rho, scratch, c->G, own_lo(c));
    HIP_TRY(hipGetLastError());
    double* planes = scratch + 2 * nb;
    hipLaunchKernelGGL(k_minmax_partials, dim3(g.y), dim3(256), 0, c->stream, scratch, planes, (int)g.x);
    HIP_TRY(hipGetLastError());
    std::vector<double> h(2 * (size_t)g.y);
    HIP_TRY(hipMemcpyAsync(h.data(), planes, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t b = 0; b < (size_t)g.y; ++b) { lo = std::min(lo, h[2 * b]); hi = std::max(hi, h[2 * b + 1]); }
  }
  return 0;
}

struct FlowOpts { double W0 = 0.02, R0 = 0.3, eta_W = 0.2, eta_R = 0.2, dt = 0.02, undul = 0.005; int nstep = 400, window = 30, max_retry = 10; };

// fittingDroplet (LBM_hydrovs.H:117-148): nstep parameter pairs of the flow started at (W0, R0)
int flow_run(const std::vector<bflbm_ctx*>& ctx, const FlowSeries& S, const double r0[3], double C0, double W0, double R0,
             double eta_W, double eta_R, double dt, int nstep, std::vector<std::array<double, 2>>& traj) {
  const Geo& G = ctx[0]->G;
  const double cell = 1. / ((double)G.nx * G.ny * G.nz);
  traj.assign(nstep, {W0, R0});
  double W = W0, R = R0;
  for (int k = 1; k < nstep; ++k) {
    const FlowCoef J = flow_coefficients(W, R, eta_W, eta_R, dt, C0, S);
    FlowParams P; P.R = R; P.inv_s2w = 1. / sqrt(2. * W);
    for (int d = 0; d < 3; ++d) P.r0[d] = r0[d];
    P.inv_n[0] = 1. / G.nx; P.inv_n[1] = 1. / G.ny; P.inv_n[2] = 1. / G.nz;
    double m[2] = {0., 0.};
    for (bflbm_ctx* c : ctx)
      if (reduce_blocks<2>(c, m, [&](dim3 g, double* scratch) {
            hipLaunchKernelGGL(k_flowfit, g, dim3(256), 0, c->stream, c->rho, scratch, c->G, own_lo(c), P); })) return 1;
    const double s2w = sqrt(2. * W);
    const double MfW = m[0] * cell / (s2w * s2w * s2w), MfR = m[1] * cell / s2w;
    const double C[2] = { MfW - 0.5 * J.Kw, MfR - 0.5 * J.Kr };
    const double A[2][2] = { {1. - J.Jrr, J.Jwr}, {J.Jrw, 1. - J.Jww} }, B[2] = { -eta_W * dt, eta_R * dt };
    const double det = (1. - J.Jww) * (1. - J.Jrr) - J.Jwr * J.Jrw;
    const double dW = (A[0][0] * B[0] * C[0] + A[0][1] * B[1] * C[1]) / det;
    const double dR = (A[1][0] * B[0] * C[0] + A[1][1] * B[1] * C[1]) / det;
    W += dW; R += dR;
    if (W <= 0.) { W -= dW; dt /= 5.; }                            // too large a step: undo it for W and shorten the steps (:134-137)
    if (std::fabs(W) < 1e-6) W = W0;                               // MIN_LEN_SCALE (:16, :138-140)
    traj[k] = {W, R};
    if (!std::isfinite(W) || !std::isfinite(R)) return fail("droplet flow fit: parameters became non-finite at step %d", k);
  }
  return 0;
}
// fittingDropletParams (LBM_hydrovs.H:160-213): mean of the last `window` pairs; retried from that mean with a five times
// shorter step while (max - min) / mean of either parameter over the window exceeds `undul`
int flow_fit(const std::vector<bflbm_ctx*>& ctx, const FlowOpts& o, double out[3], int* retries) {
  if (o.nstep < 2 || o.window < 1 || o.window > o.nstep || !(o.W0 > 0.)) return fail("droplet flow fit: bad options");
  static const FlowSeries S = flow_series();
  double mom[kNMom];
  if (moments_of(ctx, mom)) return 1;
  const Geo& G = ctx[0]->G;
  const double r0[3] = { (mom[1] / mom[0] + 0.5) / G.nx, (mom[2] / mom[0] + 0.5) / G.ny, (mom[3] / mom[0] + 0.5) / G.nz };   // getCenterOfMass (:62-113): cell centres, unit box
  double lo, hi;
  if (rho_range(ctx, lo, hi)) return 1;
  const double C0 = hi - lo;
  std::vector<std::array<double, 2>> traj;
  auto stats = [&](double mean[2], double und[2]) {
    for (int d = 0; d < 2; ++d) {
      double s = 0., mx = traj[o.nstep - o.window][d], mn = mx;
      for (int k = o.nstep - o.window; k < o.nstep; ++k) { s += traj[k][d]; mx = std::max(mx, traj[k][d]); mn = std::min(mn, traj[k][d]); }
      mean[d] = s / o.window; und[d] = (mx - mn) / mean[d];
    }
  };
  double mean[2], und[2];
  if (flow_run(ctx, S, r0, C0, o.W0, o.R0, o.eta_W, o.eta_R, o.dt, o.nstep, traj)) return 1;
  stats(mean, und);
  int it = 0;
  double dt = o.dt / 5.;
  while (it < o.max_retry && !(und[0] <= o.undul && und[1] <= o.undul)) {
    if (flow_run(ctx, S, r0, C0, mean[0], mean[1], o.eta_W, o.eta_R, dt, o.nstep, traj)) return 1;
    stats(mean, und);
    ++it; dt /= 5.;
  }
  if (retries) *retries = it;
  out[0] = mean[0]; out[1] = mean[1]; out[2] = std::max(und[0], und[1]);
  if (!(und[0] <= o.undul && und[1] <= o.undul))
    return fail("statistical undulation (%.2e, %.2e) out of bounds (fittingDropletParams)", und[0], und[1]);
  return 0;
}
FlowOpts flow_opts(const bflbm_flowfit_opts* u) {
  FlowOpts o;
  if (u) { o.W0 = u->W0; o.R0 = u->R0; o.eta_W = u->eta_W; o.eta_R = u->eta_R; o.dt = u->dt; o.undul = u->undul_ratio;
           o.nstep = u->nstep; o.window = u->step_window; o.max_retry = u->max_retry; }
  return o;
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
void bflbm_flowfit_default_opts(bflbm_flowfit_opts* o) {
  if (!o) return;
  const FlowOpts d;
  o->W0 = d.W0; o->R0 = d.R0; o->eta_W = d.eta_W; o->eta_R = d.eta_R; o->dt = d.dt; o->undul_ratio = d.undul;
  o->nstep = d.nstep; o->step_window = d.window; o->max_retry = d.max_retry;
}
int bflbm_fit_droplet_flow(bflbm_ctx* c, const bflbm_flowfit_opts* opts, double result[3], int* retries) {
  if (!c || !result) return fail("null argument");
  if (!c->G.zwrap) return fail("bflbm_fit_droplet_flow: a slab of a decomposed lattice; use bflbm_ring_fit_droplet_flow");
  return flow_fit({c}, flow_opts(opts), result, retries);
}
int bflbm_ring_fit_droplet_flow(bflbm_ring* r, const bflbm_flowfit_opts* opts, double result[3], int* retries) {
  if (!r || !result) return fail("null argument");
  return flow_fit(r->ctx, flow_opts(opts), result, retries);
}
// host-only: the closed forms of one flow step (no GPU needed; tests compare them with numerical quadrature)
int bflbm_flowfit_coefficients(double W, double R, double eta_W, double eta_R, double dt, double C0, double out[9]) {
  if (!out || !(W > 0.)) return fail("bad argument");
  static const FlowSeries S = flow_series();
  const FlowCoef J = flow_coefficients(W, R, eta_W, eta_R, dt, C0, S);
  const double c = R / sqrt(2. * W);
  out[0] = J.Jrr; out[1] = J.Jwr; out[2] = J.Jrw; out[3] = J.Jww; out[4] = J.Kw; out[5] = J.Kr;
  out[6] = flow_In(2, c, S); out[7] = flow_In(3, c, S); out[8] = flow_In(4, c, S);
  return 0;
}

}  // extern "C"

#endif  // BFLBM_DROPLET_H_
