// bflbm_site.h -- per-site arithmetic of the D3Q19 binary fluctuating-LBM step for gfx950.
//
// Every expression keeps the floating-point operation order of the reference
// (file:line cited per function) and this translation unit is compiled with
// -ffp-contract=off, so results are bit-identical to the reference's CPU build.
// Uniform sub-expressions (those that depend only on the model parameters) are
// evaluated once on the host, in the reference's order, into DevParams.
#ifndef BFLBM_SITE_H_
#define BFLBM_SITE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>
#include "bflbm_rng.h"
#include "bflbm_normal_table.h"

#define Q 19

// Velocity set, reference order (LBM_d3q19.H:12-32).
#define BFLBM_CX {0, 1,-1, 0, 0, 0, 0, 1,-1, 1,-1, 0, 0, 0, 0, 1,-1, 1,-1}
#define BFLBM_CY {0, 0, 0, 1,-1, 0, 0, 1,-1,-1, 1, 1,-1, 1,-1, 0, 0, 0, 0}
#define BFLBM_CZ {0, 0, 0, 0, 0, 1,-1, 0, 0, 0, 0, 1,-1,-1, 1, 1,-1,-1, 1}

struct DevParams {
  // model constants (LBM_binary.H:17-30, LBM_d3q19.H:6-10)
  double cs2, cs4, tau_f, tau_g, alpha0, kBT;
  // uniform sub-expressions, evaluated on the host in the reference's order
  double wcs1, wcs2;            // w[1]/cs2, w[7]/cs2                (LBM_binary.H:143)
  double neg_cs2_alpha0;        // -cs2*alpha0                       (:254-255)
  double kf, kg;                // 0.5/(tau_f+0.5), 0.5/(tau_g+0.5)  (:266, :270)
  double inv_tau_f_bar, inv_tau_g_bar;   // 1./tau_f_bar, 1./tau_g_bar (:504-508)
  double coefC;                 // 1./cs2                            (:366, :414)
  double coefC_cs2;             // coefC*cs2                         (:383)
  double two_cs4, six_cs4;      // 2.*cs4, 6.*cs4                    (:387-389)
  double modifactor;            // 1./(1.+1./(2.*tau_f))             (:424)
  double mod2;                  // modifactor*2.                     (:431, :433)
  double amp_j;                 // 2.*(tfb-0.5*tfb2)*kBT             (:117)
  double amp_f[Q], amp_g[Q];    // 2.*(tfb-0.5*tfb2)*kBT/cs2*b[a]    (:125-126)
  double samp[6];               // sqrt(amp_f[a]) for the six distinct mode norms (a = 4, 5, 6, 7, 13, 16)
  uint32_t seed_lo, seed_hi;
  int noise_on;                 // kBT != 0
};

// RN(x/3) and RN(x/9) with one multiply and two FMAs instead of the ~10-instruction IEEE
// division sequence.  Exactness (for normal x, no under/overflow): write x = X*2^e with X a
// 53-bit integer.  X/3 = k + r/3, r in {0,1,2}; the floats near X/3 are spaced 1/4 or 1/2 and the
// rounding midpoints are odd multiples of 1/8 or 1/4, so X/3 is either a float or at least 1/6 ulp
// away from every midpoint.  For 9: X/9 = k + r/9, spacing 1/16 or 1/8, and |r/9 - (2j+1)/32| =
// |32 r - 9 (2j+1)|/288 >= 1/288 (odd numerator), i.e. at least 1/18 ulp from every midpoint.
// q0 = RN(x*y), y = RN(1/d), is within 1.5 ulp of x/d; rem = x - d*q0 is an integer multiple
// (|.| <= 13) of ulp(q0), hence exact in the FMA; q0 + rem*y = x/d + (x/d - q0)*eps with
// |eps| <= 2^-53, i.e. within 2^-52 ulp of x/d, far inside the 1/18 ulp margin, so the final
// FMA rounding returns RN(x/d).  Division by 36, 12, 24, 48, 72 (LBM_d3q19.H:175-193) is one of
// these followed by an exact power-of-two scaling.  tools/div_probe.hip checks 2^32 samples.
BFLBM_HD double d_div3(double x) {
  const double y = 0x1.5555555555555p-2;
  const double q = x * y;
  const double r = __builtin_fma(-3.0, q, x);
  return __builtin_fma(r, y, q);
}
BFLBM_HD double d_div9(double x) {
  const double y = 0x1.c71c71c71c71cp-4;
  const double q = x * y;
  const double r = __builtin_fma(-9.0, q, x);
  return __builtin_fma(r, y, q);
}

// IEEE division a/b with the reciprocal of b shared between numerators.  hipcc expands a/b into
//   v_div_scale x2, v_rcp_f64, 4 FMAs refining the reciprocal, q = a*y, r = fma(-b,q,a),
//   v_div_fmas (= fma(r,y,q) plus the range scaling), v_div_fixup (specials)
// d_recip/d_div perform exactly those operations on unscaled operands, so for finite operands
// whose quotient and reciprocal are far from the over/underflow thresholds (always the case for
// densities, velocities and cs4) the result is bit-identical to a/b; tools/div_probe.hip checks it.
// The per-site denominators are only rho, phi, rho+phi and the uniform cs4: 53 divisions become
// 4 reciprocals + 53 x (mul + 2 fma).
__device__ __forceinline__ double d_recip(double b) {
  double y = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
  return y;
}
__device__ __forceinline__ double d_div(double a, double b, double yb) {
  const double q = a * yb;
  const double r = __builtin_fma(-b, q, a);
  return __builtin_fma(r, yb, q);
}
struct SiteRecip { double rho, phi, tot, cs4; };   // refined reciprocals of rho, phi, rho+phi, cs4
__device__ __forceinline__ void d_site_recips(const DevParams& P, double rho, double phi, SiteRecip& R) {
  R.rho = d_recip(rho); R.phi = d_recip(phi); R.tot = d_recip(rho + phi); R.cs4 = d_recip(P.cs4);
}

// ---- the 19x19 moment transform (LBM_d3q19.H:100-156 and its inverse :167-247), in this project's own terms.
// The velocity set is one rest population, three AXES with a (+,-) pair each (1,2 | 3,4 | 5,6) and three coordinate
// PLANES with four diagonal populations each (xy 7..10, yz 11..14, xz 15..18).  A pair enters the moments through
// its sum and difference; a plane's quad q0..q3 through four signed sums, accumulated left to right:
//   u = q0 - q1 + q2 - q3,  v = q0 - q1 - q2 + q3,  w = q0 + q1 - q2 - q3,  t = q0 + q1 + q2 + q3.
// Which Cartesian component u and v carry follows from the plane's velocities (table in d_moments).  Every
// accumulator receives its terms in increasing population index and the combinations below keep the association
// of the reference's expressions, so the doubles equal the reference's; tests check it against the oracle.
struct Quad { double u, v, w, t; };
BFLBM_HD Quad d_quad(double q0, double q1, double q2, double q3) {
  Quad r;
  r.u = q0; r.v = q0; r.w = q0; r.t = q0;
  r.u -= q1; r.v -= q1; r.w += q1; r.t += q1;
  r.u += q2; r.v -= q2; r.w -= q2; r.t += q2;
  r.u -= q3; r.v += q3; r.w -= q3; r.t += q3;
  return r;
}

// populations -> moments
BFLBM_HD void d_moments(const double (&fs)[Q], double (&m)[Q]) {
  double dif[3], sum[3];                       // per axis: f(+) - f(-), f(+) + f(-)
#pragma unroll
  for (int a = 0; a < 3; ++a) { dif[a] = fs[1 + 2*a]; sum[a] = fs[1 + 2*a]; dif[a] -= fs[2 + 2*a]; sum[a] += fs[2 + 2*a]; }
  const Quad xy = d_quad(fs[7], fs[8], fs[9], fs[10]);      // u: x-momentum, v: y-momentum
  const Quad yz = d_quad(fs[11], fs[12], fs[13], fs[14]);   // u: y-momentum, v: z-momentum
  const Quad xz = d_quad(fs[15], fs[16], fs[17], fs[18]);   // u: x-momentum, v: z-momentum
  const double rest = fs[0];
  const double shell1 = sum[0] + sum[1] + sum[2];           // populations of speed 1
  const double shell2 = xy.t + yz.t + xz.t;                 // populations of speed sqrt 2
  m[0]  = rest + shell1 + shell2;
  m[1]  = dif[0] + xy.u + xz.u;
  m[2]  = dif[1] + yz.u + xy.v;
  m[3]  = dif[2] + xz.v + yz.v;
  m[4]  = shell2 - rest;
  m[5]  = 3.*sum[0] - shell1 + shell2 - 3.*yz.t;
  m[6]  = sum[1] - sum[2] + xy.t - xz.t;
  m[7]  = xy.w;
  m[8]  = yz.w;
  m[9]  = xz.w;
  m[10] = m[1] - 3.*dif[0];
  m[11] = m[2] - 3.*dif[1];
  m[12] = m[3] - 3.*dif[2];
  m[13] = xy.u - xz.u;
  m[14] = yz.u - xy.v;
  m[15] = xz.v - yz.v;
  m[16] = m[0] - 3.*shell1;
  m[17] = shell1 - 3.*sum[0] + shell2 - 3.*yz.t;
  m[18] = sum[2] - sum[1] + xy.t - xz.t;
}

// moments -> populations.  PopTerms: what each group of populations is assembled from -- the rest population,
// per axis the even part E and odd part O of its pair (f(+-) = E +- O), per plane the base B and the three signed
// terms of its quad: q0 = B + p + q + r, q1 = B - p - q + r, q2 = B + p - q - r, q3 = B - p + q - r.
struct PopTerms { double rest, E[3], O[3], B[3], p[3], q[3], r[3]; };   // planes: 0 xy, 1 yz, 2 xz
BFLBM_HD void d_population_terms(const double (&mom)[Q], PopTerms& T) {
  double m[Q];
  // mom/{36,12,12,12,24,48,16,4,4,4,24,24,24,8,8,8,72,48,16} (the norms of the basis), bit-identical to the divisions
  m[0]  = d_div9(mom[0])  * 0.25;
  m[1]  = d_div3(mom[1])  * 0.25;
  m[2]  = d_div3(mom[2])  * 0.25;
  m[3]  = d_div3(mom[3])  * 0.25;
  m[4]  = d_div3(mom[4])  * 0.125;
  m[5]  = d_div3(mom[5])  * 0.0625;
  m[6]  = mom[6]  * 0.0625;
  m[7]  = mom[7]  * 0.25;
  m[8]  = mom[8]  * 0.25;
  m[9]  = mom[9]  * 0.25;
  m[10] = d_div3(mom[10]) * 0.125;
  m[11] = d_div3(mom[11]) * 0.125;
  m[12] = d_div3(mom[12]) * 0.125;
  m[13] = mom[13] * 0.125;
  m[14] = mom[14] * 0.125;
  m[15] = mom[15] * 0.125;
  m[16] = d_div9(mom[16]) * 0.125;
  m[17] = d_div3(mom[17]) * 0.0625;
  m[18] = mom[18] * 0.0625;
  T.rest = 12.*(m[0] - m[4] + m[16]);
  const double e1 = 2.*(m[0] - 2.*m[16]);      // common even part of the speed-1 shell
  const double e2 = m[0] + m[4] + m[16];       // common even part of the speed-sqrt-2 shell
#pragma unroll
  for (int a = 0; a < 3; ++a) T.O[a] = 2.*(m[1 + a] - 2.*m[10 + a]);
  T.E[0] = e1 + 4.*(m[5] - m[17]);
  T.E[1] = e1 - 2.*(m[5] - m[6]) + 2.*(m[17] - m[18]);
  T.E[2] = e1 - 2.*(m[5] + m[6]) + 2.*(m[17] + m[18]);
  T.B[0] = e2 + (m[5] + m[6]) + (m[17] + m[18]);
  T.B[1] = e2 - 2.*(m[5] + m[17]);
  T.B[2] = e2 + (m[5] - m[6]) + (m[17] - m[18]);
  // momentum-like parts carried by the diagonals: (component sum) +- (ghost-vector part m13..15)
  const double jx = m[1] + m[10], jy = m[2] + m[11], jz = m[3] + m[12];
  T.p[0] = jx + m[13]; T.q[0] = jy - m[14]; T.r[0] = m[7];     // xy: x-part, y-part, shear
  T.p[1] = jy + m[14]; T.q[1] = jz - m[15]; T.r[1] = m[8];     // yz: y-part, z-part, shear
  T.p[2] = jz + m[15]; T.q[2] = jx - m[13]; T.r[2] = m[9];     // xz: z-part, x-part, shear
}
BFLBM_HD void d_populations(const double (&mom)[Q], double (&f)[Q]) {
  PopTerms T;
  d_population_terms(mom, T);
  f[0] = T.rest;
#pragma unroll
  for (int a = 0; a < 3; ++a) { f[1 + 2*a] = T.E[a] + T.O[a]; f[2 + 2*a] = T.E[a] - T.O[a]; }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const double q0 = T.B[g] + T.p[g] + T.q[g] + T.r[g];
    const double q1 = T.B[g] - T.p[g] - T.q[g] + T.r[g];
    const double q2 = T.B[g] + T.p[g] - T.q[g] - T.r[g];
    const double q3 = T.B[g] - T.p[g] + T.q[g] - T.r[g];
    // the xz plane lists (1,0,-1) before (-1,0,1) (LBM_d3q19.H:29-32): its third and fourth populations swap
    f[7 + 4*g] = q0; f[8 + 4*g] = q1; f[9 + 4*g] = (g == 2) ? q3 : q2; f[10 + 4*g] = (g == 2) ? q2 : q3;
  }
}

// rho = sum_i f_i in index order (hydrovars_bar_density, LBM_binary.H:320-328)
__device__ __forceinline__ double d_density(const double (&fs)[Q]) {
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < Q; ++i) r += fs[i];
  return r;
}

// j = sum_i f_i c_i in index order (hydrovars, LBM_binary.H:218-228): terms with c=0 add +-0.
__device__ __forceinline__ void d_momentum(const double (&fs)[Q], double (&j)[3]) {
  double jx = 0.0, jy = 0.0, jz = 0.0;
  jx += fs[1];  jx -= fs[2];
  jy += fs[3];  jy -= fs[4];
  jz += fs[5];  jz -= fs[6];
  jx += fs[7];  jy += fs[7];
  jx -= fs[8];  jy -= fs[8];
  jx += fs[9];  jy -= fs[9];
  jx -= fs[10]; jy += fs[10];
  jy += fs[11]; jz += fs[11];
  jy -= fs[12]; jz -= fs[12];
  jy += fs[13]; jz -= fs[13];
  jy -= fs[14]; jz += fs[14];
  jx += fs[15]; jz += fs[15];
  jx -= fs[16]; jz -= fs[16];
  jx += fs[17]; jz -= fs[17];
  jx -= fs[18]; jz += fs[18];
  j[0] = jx; j[1] = jy; j[2] = jz;
}

// gradient (LBM_binary.H:134-150, use_SC_pseudo=false) from the 18 neighbour values
// nb[i] = field(x + c_i); nb[0] is unused (c_0 = 0 contributes +-0).
__device__ __forceinline__ void d_gradient(const DevParams& P, const double (&nb)[Q], double (&g)[3]) {
  double t[Q];
#pragma unroll
  for (int i = 1; i < 7; ++i) t[i] = P.wcs1 * nb[i];
#pragma unroll
  for (int i = 7; i < Q; ++i) t[i] = P.wcs2 * nb[i];
  double gx = 0.0, gy = 0.0, gz = 0.0;
  gx += t[1];  gx -= t[2];
  gy += t[3];  gy -= t[4];
  gz += t[5];  gz -= t[6];
  gx += t[7];  gy += t[7];
  gx -= t[8];  gy -= t[8];
  gx += t[9];  gy -= t[9];
  gx -= t[10]; gy += t[10];
  gy += t[11]; gz += t[11];
  gy -= t[12]; gz -= t[12];
  gy += t[13]; gz -= t[13];
  gy -= t[14]; gz += t[14];
  gx += t[15]; gz += t[15];
  gx -= t[16]; gz -= t[16];
  gx += t[17]; gz -= t[17];
  gx -= t[18]; gz += t[18];
  g[0] = gx; g[1] = gy; g[2] = gz;
}

// thermal_noise (LBM_binary.H:73-132) for one site.  The amplitude of mode a >= 4 is
// sqrt(c*b[a]*|rho|), c = 2(tb - tb^2/2) kBT/cs2, with the mode norm b[a] taking only six distinct values
// (2/3, 4/3, 4/9, 1/9, 2/9, 2; LBM_d3q19.H:56-76): sqrt(c*b) is uniform (host, DevParams.samp) and the site
// contributes sqrt|rho|, sqrt|phi| -- 3 square roots per site instead of the 31 of the reference's loop.  (The
// generated noise is pinned statistically only; the oracle restates the same factorisation, so GPU and
// oracle agree bit for bit, and tests/test_oracle_pins.py bounds the factorised amplitude against the reference's
// literal expression.)  Normals: the site's word stream, bflbm_rng.h; words 0..2 the momentum modes shared
// by both fluids, 3..17 fluid f, 18..32 fluid g, so each fluid's noise is generated right before its relaxation.
struct NoiseAmp { double sj, sr, sp; };          // sqrt(amp_j |rho phi/rhot|), sqrt|rho|, sqrt|phi|
__device__ __forceinline__ int d_noise_group(int a) {   // modes 4..18 -> index of b[a] among the six values
  return (a == 4 || (a >= 10 && a <= 12)) ? 0 : (a == 5 || a == 17) ? 1 : (a == 6 || a == 18) ? 2
       : (a >= 7 && a <= 9) ? 3 : (a >= 13 && a <= 15) ? 4 : 5;
}
// rho, phi, rhot: the state the amplitudes are taken from -- the site's own densities with
// rhot = rho + phi (LBM_binary.H:109-111) or the reference state under USE_REF_STATE (:92-107).
__device__ __forceinline__ void d_noise_amp(const DevParams& P, double rho, double phi, double rhot, NoiseAmp& A) {
  A.sj = sqrt(P.amp_j * fabs(rho*phi/rhot));
  A.sr = sqrt(fabs(rho)); A.sp = sqrt(fabs(phi));
}
// noise of mode a >= 4 of one fluid (s = A.sr or A.sp), drawing the next word of the site's stream
template <typename Tab>
__device__ __forceinline__ double d_noise_mode(const DevParams& P, double s, int a, Tab tab, bflbm_rng_state& st) {
  return (P.samp[d_noise_group(a)] * s) * bflbm_normal_from_bits(bflbm_rng_next(st), tab);
}
// seeds the site's stream and draws the momentum-mode noise (modes 1..3 of f; g gets the negative)
template <typename Tab>
__device__ __forceinline__ void d_noise_head(const DevParams& P, const NoiseAmp& A, uint64_t site, uint32_t idx, Tab tab,
                                             bflbm_rng_state& st, double (&fn3)[3]) {
  bflbm_rng_seed(P.seed_lo, P.seed_hi, site, idx, st);
#pragma unroll
  for (int k = 0; k < 3; ++k) fn3[k] = A.sj * bflbm_normal_from_bits(bflbm_rng_next(st), tab);
}
template <typename Tab>
__device__ __forceinline__ void d_noise_f(const DevParams& P, const NoiseAmp& A, Tab tab, bflbm_rng_state& st, const double (&fn3)[3], double (&fn)[Q]) {
  fn[0] = 0.; fn[1] = fn3[0]; fn[2] = fn3[1]; fn[3] = fn3[2];
#pragma unroll
  for (int a = 4; a < Q; ++a) fn[a] = d_noise_mode(P, A.sr, a, tab, st);
}
template <typename Tab>
__device__ __forceinline__ void d_noise_g(const DevParams& P, const NoiseAmp& A, Tab tab, bflbm_rng_state& st, const double (&fn3)[3], double (&gn)[Q]) {
  gn[0] = 0.; gn[1] = -fn3[0]; gn[2] = -fn3[1]; gn[3] = -fn3[2];
#pragma unroll
  for (int a = 4; a < Q; ++a) gn[a] = d_noise_mode(P, A.sp, a, tab, st);
}
// all 38 noise moments (observables)
template <typename Tab>
__device__ __forceinline__ void d_noise(const DevParams& P, double rho, double phi, double rhot, uint64_t site,
                                        uint32_t noise_index, Tab tab, double (&fn)[Q], double (&gn)[Q]) {
  NoiseAmp A; d_noise_amp(P, rho, phi, rhot, A);
  double fn3[3]; bflbm_rng_state st;
  d_noise_head(P, A, site, noise_index, tab, st, fn3);
  d_noise_f(P, A, tab, st, fn3, fn);
  d_noise_g(P, A, tab, st, fn3, gn);
}

// the normal table in LDS: every thread of the workgroup calls this before any early exit
__constant__ double bflbm_normal_table_dev[BFLBM_NORMAL_TABLE_N] = BFLBM_NORMAL_TABLE_VALUES;
__device__ __forceinline__ void d_load_normal_table(double* lds, bool wanted) {
  if (wanted) for (int i = threadIdx.x; i < BFLBM_NORMAL_TABLE_N; i += blockDim.x) lds[i] = bflbm_normal_table_dev[i];
  __syncthreads();
}

// The quantities hydrovars() derives per site (LBM_binary.H:196-295) that collide() consumes.
struct SiteHydro {
  double uf[3], ug[3];        // h[2..4], h[6..8]  real velocities
  double af[3], ag[3];        // h[9..11], h[12..14]
  double ufbar[3], ugbar[3];  // lattice velocities
  double nfvel[3], ngvel[3];
};

__device__ __forceinline__ void d_hydrovars_j(const DevParams& P, const double (&jf)[3], const double (&jg)[3],
                                              double rho, double phi,
                                              const double (&grad_rho)[3], const double (&grad_phi)[3],
                                              const double (&nf3)[3], const double (&ng3)[3], SiteHydro& H,
                                              const SiteRecip& R) {
  const bool okr = fabs(rho) > (double)FLT_EPSILON;
  const bool okp = fabs(phi) > (double)FLT_EPSILON;
  const double tot = rho + phi;
  const double wphi = d_div(P.kf*phi, tot, R.tot);
  const double wrho = d_div(P.kg*rho, tot, R.tot);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    H.ufbar[k] = okr ? d_div(jf[k], rho, R.rho) : 0.;
    H.ugbar[k] = okp ? d_div(jg[k], phi, R.phi) : 0.;
    H.af[k] = okr ? d_div(P.neg_cs2_alpha0*rho*grad_phi[k], rho, R.rho) : 0.;
    H.ag[k] = okp ? d_div(P.neg_cs2_alpha0*phi*grad_rho[k], phi, R.phi) : 0.;
    H.nfvel[k] = okr ? d_div(nf3[k], rho, R.rho) : 0.;
    H.ngvel[k] = okp ? d_div(ng3[k], phi, R.phi) : 0.;
    H.uf[k] = H.ufbar[k] + 0.5*H.af[k] - wphi*(H.ufbar[k]-H.ugbar[k] + 0.5*(H.af[k]-H.ag[k])) + 0.5*H.nfvel[k];
    H.ug[k] = H.ugbar[k] + 0.5*H.ag[k] - wrho*(H.ugbar[k]-H.ufbar[k] + 0.5*(H.ag[k]-H.af[k])) + 0.5*H.ngvel[k];
  }
}

__device__ __forceinline__ void d_hydrovars(const DevParams& P, const double (&fs)[Q], const double (&gs)[Q],
                                            double rho, double phi,
                                            const double (&grad_rho)[3], const double (&grad_phi)[3],
                                            const double (&nf)[Q], const double (&ng)[Q], SiteHydro& H,
                                            const SiteRecip& R) {
  double jf[3], jg[3];
  d_momentum(fs, jf);
  d_momentum(gs, jg);
  const double nf3[3] = { nf[1], nf[2], nf[3] }, ng3[3] = { ng[1], ng[2], ng[3] };
  d_hydrovars_j(P, jf, jg, rho, phi, grad_rho, grad_phi, nf3, ng3, H, R);
}

// equilibrium_moments (LBM_binary.H:356-402): only modes 0..9 are non-zero.
__device__ __forceinline__ void d_equilibrium(const DevParams& P, double rho, const double (&u)[3], double (&mEq)[10], double ycs4) {
  const double A00 = d_div((rho*u[0]*u[0])/2., P.cs4, ycs4);
  const double A01 = d_div((rho*u[0]*u[1])/2., P.cs4, ycs4);
  const double A02 = d_div((rho*u[0]*u[2])/2., P.cs4, ycs4);
  const double A11 = d_div((rho*u[1]*u[1])/2., P.cs4, ycs4);
  const double A12 = d_div((rho*u[1]*u[2])/2., P.cs4, ycs4);
  const double A22 = d_div((rho*u[2]*u[2])/2., P.cs4, ycs4);
  const double tr = A00 + A11 + A22;
  mEq[0] = rho*1.;
  mEq[1] = P.coefC_cs2*(rho*u[0]);
  mEq[2] = P.coefC_cs2*(rho*u[1]);
  mEq[3] = P.coefC_cs2*(rho*u[2]);
  mEq[4] = P.two_cs4*tr;
  mEq[5] = P.six_cs4*A00 - P.two_cs4*tr;
  mEq[6] = P.two_cs4*(A11 - A22);
  mEq[7] = P.cs4*(A01 + A01);
  mEq[8] = P.cs4*(A12 + A12);
  mEq[9] = P.cs4*(A02 + A02);
}

// phi_moments (LBM_binary.H:404-449)
__device__ __forceinline__ void d_force_moments(const DevParams& P, double rho, const double (&u)[3], const double (&a)[3], double (&mPhi)[10], double ycs4) {
  const double coefAC = rho*P.coefC;
  double ru[3] = { rho*u[0], rho*u[1], rho*u[2] };
  const double A00 = d_div(a[0]*ru[0], P.cs4, ycs4), A01 = d_div(a[0]*ru[1], P.cs4, ycs4), A02 = d_div(a[0]*ru[2], P.cs4, ycs4);
  const double A10 = d_div(a[1]*ru[0], P.cs4, ycs4), A11 = d_div(a[1]*ru[1], P.cs4, ycs4), A12 = d_div(a[1]*ru[2], P.cs4, ycs4);
  const double A20 = d_div(a[2]*ru[0], P.cs4, ycs4), A21 = d_div(a[2]*ru[1], P.cs4, ycs4), A22 = d_div(a[2]*ru[2], P.cs4, ycs4);
  const double tr = A00 + A11 + A22;
  mPhi[0] = P.modifactor*(rho*0.);
  mPhi[1] = P.modifactor*coefAC*P.cs2*a[0];
  mPhi[2] = P.modifactor*coefAC*P.cs2*a[1];
  mPhi[3] = P.modifactor*coefAC*P.cs2*a[2];
  mPhi[4] = P.mod2*P.cs4*tr;
  mPhi[5] = P.modifactor*(P.six_cs4*A00 - P.two_cs4*tr);
  mPhi[6] = P.mod2*P.cs4*(A11 - A22);
  mPhi[7] = P.modifactor*P.cs4*(A01 + A10);
  mPhi[8] = P.modifactor*P.cs4*(A12 + A21);
  mPhi[9] = P.modifactor*P.cs4*(A02 + A20);
}

// The relaxation loop of collide (LBM_binary.H:504-511) for one fluid, in moment space:
// m += (mEq - m)/tau_bar + mPhi + noise, with the fluid's density rho_k, the barycentric
// velocity v_b (equilibrium), its own real velocity u and acceleration a (force moments).
// The noise of mode k is asked from `noise(k)` in mode order right where it is added (a generated stream never
// exists as an array of 19 doubles); NOISE=false drops the terms (they are exactly +-0 when kBT == 0).
template <bool NOISE, typename NoiseFn>
__device__ __forceinline__ void d_relax_with(const DevParams& P, double (&m)[Q], double rho_k, const double (&v_b)[3],
                                             const double (&u)[3], const double (&a)[3], double inv_tau_bar,
                                             NoiseFn noise, double ycs4) {
  double mEq[10], mPhi[10];
  d_equilibrium(P, rho_k, v_b, mEq, ycs4);
  d_force_moments(P, rho_k, u, a, mPhi, ycs4);
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    double R = inv_tau_bar*(mEq[k] - m[k]) + mPhi[k];
    if (NOISE) R = R + noise(k);
    m[k] = m[k] + R;
  }
#pragma unroll
  for (int k = 10; k < Q; ++k) {
    double R = inv_tau_bar*(0. - m[k]) + 0.;
    if (NOISE) R = R + noise(k);
    m[k] = m[k] + R;
  }
}
template <bool NOISE>
__device__ __forceinline__ void d_relax(const DevParams& P, double (&m)[Q], double rho_k, const double (&v_b)[3],
                                        const double (&u)[3], const double (&a)[3], double inv_tau_bar,
                                        const double (&noise)[Q], double ycs4) {
  d_relax_with<NOISE>(P, m, rho_k, v_b, u, a, inv_tau_bar, [&](int k) { return noise[k]; }, ycs4);
}
// relaxation of one fluid with the generated stream: n3 = its momentum-mode noise (fn3 or -fn3), s = A.sr / A.sp.
// The fluid's 15 normals are drawn first (their 15 table look-ups are in flight together) and scaled where they are added.
template <typename Tab>
__device__ __forceinline__ void d_relax_generated(const DevParams& P, double (&m)[Q], double rho_k, const double (&v_b)[3],
                                                  const double (&u)[3], const double (&a)[3], double inv_tau_bar,
                                                  const double (&n3)[3], double s, Tab tab, bflbm_rng_state& st, double ycs4) {
  double nrm[Q - 4];
#pragma unroll
  for (int k = 4; k < Q; ++k) nrm[k - 4] = bflbm_normal_from_bits(bflbm_rng_next(st), tab);
  double amp[6];                                 // the six distinct amplitudes of this fluid (same products as d_noise_mode)
#pragma unroll
  for (int g = 0; g < 6; ++g) amp[g] = P.samp[g] * s;
  d_relax_with<true>(P, m, rho_k, v_b, u, a, inv_tau_bar,
                     [&](int k) { return k == 0 ? 0. : (k < 4 ? n3[k - 1] : amp[d_noise_group(k)] * nrm[k - 4]); }, ycs4);
}

__device__ __forceinline__ void d_barycentric(double rho, double phi, const SiteHydro& H, double (&v_b)[3], const SiteRecip& R) {
#pragma unroll
  for (int k = 0; k < 3; ++k) v_b[k] = d_div(rho*H.uf[k] + phi*H.ug[k], rho + phi, R.tot);   // LBM_binary.H:471
}

// collide (LBM_binary.H:451-516): fs,gs are replaced by the post-collision populations.
// NOISE=false drops the noise terms (they are exactly +-0 when kBT == 0).
template <bool NOISE>
__device__ __forceinline__ void d_collide(const DevParams& P, double (&fs)[Q], double (&gs)[Q],
                                          double rho, double phi, const SiteHydro& H,
                                          const double (&fn)[Q], const double (&gn)[Q], const SiteRecip& R) {
  double v_b[3];
  d_barycentric(rho, phi, H, v_b, R);
  {
    double m[Q];
    d_moments(fs, m);
    d_relax<NOISE>(P, m, rho, v_b, H.uf, H.af, P.inv_tau_f_bar, fn, R.cs4);
    d_populations(m, fs);
  }
  {
    double m[Q];
    d_moments(gs, m);
    d_relax<NOISE>(P, m, phi, v_b, H.ug, H.ag, P.inv_tau_g_bar, gn, R.cs4);
    d_populations(m, gs);
  }
}

#endif  // BFLBM_SITE_H_
