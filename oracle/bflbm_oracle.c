/*
 * bflbm_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the reference's D3Q19 binary fluctuating-LBM path
 * (reference = MDProject/Binary-Fluctuating-Lattice-Boltzmann, files LBM_d3q19.H,
 * LBM_binary.H, LBM_hydrovs.H).  Every function cites the reference file:line it
 * follows.  Floating-point operation ORDER is kept identical to the reference
 * expressions so that, compiled with -ffp-contract=off, this file produces the
 * same doubles as the reference CPU build (g++, no FMA).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker / reported baseline.  The product path
 * (binary-fluctuating-lattice-boltzmann_amd/csrc) never links or calls it.
 *
 * PINNING: the reference cannot be compiled in this image (it needs AMReX, an
 * external library that is absent; writing stand-in headers is not allowed), and
 * it ships no golden vectors.  The oracle is pinned by outputs of the reference
 * that its authors recorded in the notebooks under /root/reference: Flat_Interface.ipynb
 * cell 4 (interface height 47.86628666 at 8x256x64, frame 2000), Surface_Tension.ipynb cells
 * 13-19 (36 densities with 16 digits, force integrals, fitted radii and both Laplace-law surface
 * tensions of nine 32^3 droplets at frame 20000) and Droplet_Fluctuation.ipynb cell 5 (centre of
 * mass after 20000 steps) -- tests/test_oracle_pins.py, tests/test_gpu_notebook_surface_tension.py,
 * tests/test_gpu_fullsize.py -- and by algebraic identities of the D3Q19 basis.  All of those runs
 * used tau = 1/2 (full relaxation); for tau != 1/2 see DESIGN.md section 4 (parity unpinned by any
 * recorded reference number; invariant tests instead).  The three numbers of SURVEY.md section 8c
 * (8^3 stripe, 10 steps) are kept as a regression check only: they came from a survey-time build of the
 * reference headers against stand-in AMReX types and pin nothing by themselves.
 * The Gaussian random stream (amrex::RandomNormal) is an un-vendored dependency:
 * noise parity with the reference is statistical only ("parity unpinned" at the
 * RNG boundary); this file defines the project's own counter-based stream.
 *
 * Geometry: one periodic box nx*ny*nz, no ghost cells.  Array layout is
 * component-slowest, x-fastest (the AMReX Array4 order): a[((c*nz+z)*ny+y)*nx+x].
 * The reference executes collide+push on valid+1 ghost layer with halo-filled
 * inputs; on a periodic domain this is the same as collide on valid cells and
 * push with periodic wrap, which is what orc_timestep does.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define Q 19
#define NHBAR 15   /* hydrovsbar comps allocated by the driver (main_run_job.cpp:210); 0..8 written */
#define NHYDRO 22  /* hydrovs comps (main_run_job.cpp:147) */

/* Threads over z planes for the optional "all cores" timing of bench.py's cpu_baseline leg.  Sites are
 * independent in every loop that carries the pragma, so results do not depend on the thread count;
 * the default is 1 = the reference as shipped (serial build, GNUmakefile:16-19). */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }

typedef struct orc_params {
  double tau_f, tau_g;   /* LBM_binary.H:18-19 */
  double alpha0, alpha1; /* LBM_binary.H:20-21 */
  double kappa;          /* LBM_binary.H:30 */
  double kBT;            /* LBM_d3q19.H:10 */
  double cs2, cs4;       /* LBM_d3q19.H:6-7 */
  double rho_lo, rho_hi; /* LBM_binary.H:25-26 */
  uint64_t seed;         /* LBM_binary.H:17 */
} orc_params;

/* LBM_d3q19.H:12-32 */
static const int C[Q][3] = {
  {0,0,0},
  {1,0,0},{-1,0,0},{0,1,0},{0,-1,0},{0,0,1},{0,0,-1},
  {1,1,0},{-1,-1,0},{1,-1,0},{-1,1,0},
  {0,1,1},{0,-1,-1},{0,1,-1},{0,-1,1},
  {1,0,1},{-1,0,-1},{1,0,-1},{-1,0,1}
};
/* LBM_d3q19.H:34-54 */
static const double W[Q] = {
  1./3.,
  1./18.,1./18.,1./18.,1./18.,1./18.,1./18.,
  1./36.,1./36.,1./36.,1./36.,1./36.,1./36.,
  1./36.,1./36.,1./36.,1./36.,1./36.,1./36.
};
/* LBM_d3q19.H:56-76 */
static const double B[Q] = {
  1.0, 1./3., 1./3., 1./3., 2./3., 4./3., 4./9., 1./9., 1./9., 1./9.,
  2./3., 2./3., 2./3., 2./9., 2./9., 2./9., 2.0, 4./3., 4./9.
};

void orc_default_params(orc_params* p) {
  p->tau_f = 1./2.; p->tau_g = 1./2.;
  p->alpha0 = 4.; p->alpha1 = 0.;
  p->kappa = 4;
  p->kBT = 0.;
  p->cs2 = 1./3.; p->cs4 = (1./3.)*(1./3.);
  p->rho_lo = 0.; p->rho_hi = 1.0;
  p->seed = 12345ULL;
}

void orc_lattice(int* c_out, double* w_out, double* b_out) {
  for (int i = 0; i < Q; ++i) {
    for (int d = 0; d < 3; ++d) c_out[3*i+d] = C[i][d];
    w_out[i] = W[i]; b_out[i] = B[i];
  }
}

static inline size_t IDX(int nx, int ny, int nz, int c, int x, int y, int z) {
  return (((size_t)c*nz + z)*ny + y)*(size_t)nx + x;
}
static inline int wrap(int a, int n) { return a < 0 ? a + n : (a >= n ? a - n : a); }

/* ------------------------------------------------------------------ */
/* populations -> moments.  LBM_d3q19.H:100-156                        */
void orc_moments(const double* fs, double* m) {
  double f;
  double mc0, mc1, mc2;
  double mx1, my1, mz1, mx2, my2, mz2, mx3, my3, mz3;
  double mxy, mxz, myz, mxx1, myy1, mzz1, mxx2, myy2, mzz2;
  f = fs[0];  mc0 = f;
  f = fs[1];  mx1 = f;  mxx1 = f;
  f = fs[2];  mx1 -= f; mxx1 += f;
  f = fs[3];  my1 = f;  myy1 = f;
  f = fs[4];  my1 -= f; myy1 += f;
  f = fs[5];  mz1 = f;  mzz1 = f;
  f = fs[6];  mz1 -= f; mzz1 += f;
  f = fs[7];  mx2 = f;  my3 = f;  mxy = f;  mxx2 = f;
  f = fs[8];  mx2 -= f; my3 -= f; mxy += f; mxx2 += f;
  f = fs[9];  mx2 += f; my3 -= f; mxy -= f; mxx2 += f;
  f = fs[10]; mx2 -= f; my3 += f; mxy -= f; mxx2 += f;
  f = fs[11]; my2 = f;  mz3 = f;  myz = f;  myy2 = f;
  f = fs[12]; my2 -= f; mz3 -= f; myz += f; myy2 += f;
  f = fs[13]; my2 += f; mz3 -= f; myz -= f; myy2 += f;
  f = fs[14]; my2 -= f; mz3 += f; myz -= f; myy2 += f;
  f = fs[15]; mz2 = f;  mx3 = f;  mxz = f;  mzz2 = f;
  f = fs[16]; mz2 -= f; mx3 -= f; mxz += f; mzz2 += f;
  f = fs[17]; mz2 -= f; mx3 += f; mxz -= f; mzz2 += f;
  f = fs[18]; mz2 += f; mx3 -= f; mxz -= f; mzz2 += f;

  mc1 = mxx1 + myy1 + mzz1;
  mc2 = mxx2 + myy2 + mzz2;

  m[0]  = mc0 + mc1 + mc2;
  m[1]  = mx1 + mx2 + mx3;
  m[2]  = my1 + my2 + my3;
  m[3]  = mz1 + mz2 + mz3;
  m[4]  = mc2 - mc0;
  m[5]  = 3.*mxx1 - mc1 + mc2 - 3.*myy2;
  m[6]  = myy1 - mzz1 + mxx2 - mzz2;
  m[7]  = mxy;
  m[8]  = myz;
  m[9]  = mxz;
  m[10] = m[1] - 3.*mx1;
  m[11] = m[2] - 3.*my1;
  m[12] = m[3] - 3.*mz1;
  m[13] = mx2 - mx3;
  m[14] = my2 - my3;
  m[15] = mz2 - mz3;
  m[16] = m[0] - 3.*mc1;
  m[17] = mc1 - 3.*mxx1 + mc2 - 3.*myy2;
  m[18] = mzz1 - myy1 + mxx2 - mzz2;
}

/* moments -> populations.  LBM_d3q19.H:167-247 */
void orc_populations(const double* mom, double* f) {
  double m[Q];
  double mc0, mc1, mc2;
  double mx1, my1, mz1, mx2, my2, mz2, mx3, my3, mz3;
  double mxx1, myy1, mzz1, mxy, mxz, myz, mxy2, mxz2, myz2;
  m[0]  = mom[0]  / 36.;
  m[1]  = mom[1]  / 12.;
  m[2]  = mom[2]  / 12.;
  m[3]  = mom[3]  / 12.;
  m[4]  = mom[4]  / 24.;
  m[5]  = mom[5]  / 48.;
  m[6]  = mom[6]  / 16.;
  m[7]  = mom[7]  / 4.;
  m[8]  = mom[8]  / 4.;
  m[9]  = mom[9]  / 4.;
  m[10] = mom[10] / 24.;
  m[11] = mom[11] / 24.;
  m[12] = mom[12] / 24.;
  m[13] = mom[13] / 8.;
  m[14] = mom[14] / 8.;
  m[15] = mom[15] / 8.;
  m[16] = mom[16] / 72.;
  m[17] = mom[17] / 48.;
  m[18] = mom[18] / 16.;

  mc0 = 12.*(m[0] - m[4] + m[16]);
  mc1 =  2.*(m[0] - 2.*m[16]);
  mc2 = m[0] + m[4] + m[16];

  mx1 = 2.*(m[1] - 2.*m[10]);
  my1 = 2.*(m[2] - 2.*m[11]);
  mz1 = 2.*(m[3] - 2.*m[12]);

  mx2 = m[1] + m[10] + m[13];
  my2 = m[2] + m[11] + m[14];
  mz2 = m[3] + m[12] + m[15];

  mx3 = m[1] + m[10] - m[13];
  my3 = m[2] + m[11] - m[14];
  mz3 = m[3] + m[12] - m[15];

  mxx1 = mc1 + 4.*(m[5] - m[17]);
  myy1 = mc1 - 2.*(m[5] - m[6]) + 2.*(m[17] - m[18]);
  mzz1 = mc1 - 2.*(m[5] + m[6]) + 2.*(m[17] + m[18]);

  mxy2 = mc2 + (m[5] + m[6]) + (m[17] + m[18]);
  mxz2 = mc2 + (m[5] - m[6]) + (m[17] - m[18]);
  myz2 = mc2 - 2.*(m[5] + m[17]);

  mxy = m[7];
  myz = m[8];
  mxz = m[9];

  f[0]  = mc0;
  f[1]  = mxx1 + mx1;
  f[2]  = mxx1 - mx1;
  f[3]  = myy1 + my1;
  f[4]  = myy1 - my1;
  f[5]  = mzz1 + mz1;
  f[6]  = mzz1 - mz1;
  f[7]  = mxy2 + mx2 + my3 + mxy;
  f[8]  = mxy2 - mx2 - my3 + mxy;
  f[9]  = mxy2 + mx2 - my3 - mxy;
  f[10] = mxy2 - mx2 + my3 - mxy;
  f[11] = myz2 + my2 + mz3 + myz;
  f[12] = myz2 - my2 - mz3 + myz;
  f[13] = myz2 + my2 - mz3 - myz;
  f[14] = myz2 - my2 + mz3 - myz;
  f[15] = mxz2 + mz2 + mx3 + mxz;
  f[16] = mxz2 - mz2 - mx3 + mxz;
  f[17] = mxz2 - mz2 + mx3 - mxz;
  f[18] = mxz2 + mz2 - mx3 - mxz;
}

/* ------------------------------------------------------------------ */
/* Project RNG (replaces amrex::RandomNormal, LBM_binary.H:117,125,126, whose stream is not available
 * offline -- parity unpinned at that boundary).  Per (site, noise index): one Philox4x32-10
 * block (Salmon et al., SC'11) keyed by the seed with counter (site lo, site hi, noise index, 0) seeds
 * xoshiro128+ (Blackman & Vigna), whose first 33 words become the site's 33 normals through a quantile
 * table: s = sum of the word's four bytes (0..1020) has the exactly known distribution of a sum of four
 * uniform bytes, and T[s] (normal_table.h, generated by tools/make_normal_table.py) is the mean of N(0,1)
 * between the normal quantiles of that cell's cumulative probabilities, scaled to unit variance.  Integer
 * operations and one table read, so the HIP kernels reproduce it bit for bit.  The product-side definition
 * lives in csrc/bflbm_rng.h; tests/test_oracle_pins.py and tests/test_gpu_noise.py check that the two agree. */
#include "normal_table.h"
static const double NORMAL_TABLE[1024] = ORC_NORMAL_TABLE_VALUES;

static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                 uint32_t k0, uint32_t k1, uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
static inline uint32_t xoshiro128p(uint32_t s[4]) {
  const uint32_t result = s[0] + s[3];
  const uint32_t t = s[1] << 9;
  s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
  s[2] ^= t;
  s[3] = rotl32(s[3], 11);
  return result;
}

/* one standard normal from one 32-bit word */
static inline double normal_from_bits(uint32_t u) {
  return NORMAL_TABLE[(u & 255u) + ((u >> 8) & 255u) + ((u >> 16) & 255u) + (u >> 24)];
}

/* The 33 normals of one site for one noise index (out36[33..35] = 0; the array length is the C-ABI's). */
void orc_site_normals(uint64_t seed, uint64_t site, uint32_t noise_index, double* out36) {
  uint32_t s[4];
  philox4x32_10((uint32_t)site, (uint32_t)(site >> 32), noise_index, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), s);
  s[3] |= 1u;                                          /* never the all-zero state */
  for (int k = 0; k < 36; ++k) out36[k] = (k < 33) ? normal_from_bits(xoshiro128p(s)) : 0.;
}

void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
  philox4x32_10(c0, c1, c2, c3, k0, k1, out);
}

/* ------------------------------------------------------------------ */
/* thermal_noise, LBM_binary.H:73-132.  Draw count per site as in the reference: 3 + 2*15 normals.
 * gz0 = global z of local plane 0 and gnz = global nz (slab tests; the site id that keys the
 * random stream is the GLOBAL lattice index).
 * ref == NULL: the shipped branch (:109-111), rho,phi from hydrovsbar, rhot = rho+phi.
 * ref != NULL: the USE_REF_STATE branch (:92-107): rho,phi,rhot from the equilibrium fields
 * ref[0..2] (global lattice, one component each) at the site shifted by the truncated relative
 * centre of mass, wrapped once into the domain. */
static void thermal_noise_impl(const orc_params* p, int nx, int ny, int nz, int gz0, int gnz,
                               const double* hbar, const double* const* ref, const double* pos_com_relative,
                               uint32_t noise_index, double* fn, double* gn) {
  const double tau_f_bar = 1./(p->tau_f+0.5);
  const double tau_g_bar = tau_f_bar;                /* :80 (sic) */
  const double tau_f_bar2 = tau_f_bar*tau_f_bar;
  const double tau_g_bar2 = tau_g_bar*tau_g_bar;
  const double kBT = p->kBT, cs2 = p->cs2;
  #pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
    int gz = (z + gz0) % gnz; if (gz < 0) gz += gnz;
    double rho, phi, rhot;
    if (ref) {
      /* static_cast<int>: truncation, :94-96.  The reference then wraps ONCE (:98-103), which is in range
       * only for |shift| < n; the shift is therefore reduced modulo n first (no change for |shift| < n,
       * where the reference is defined) so that no centre of mass can index outside the fields. */
      int x_shift = x - (int)fmod(trunc(pos_com_relative[0]), (double)nx);
      int y_shift = y - (int)fmod(trunc(pos_com_relative[1]), (double)ny);
      int z_shift = gz - (int)fmod(trunc(pos_com_relative[2]), (double)gnz);
      if (x_shift < 0) x_shift += nx;                 /* :98-103 */
      if (x_shift > nx-1) x_shift -= nx;
      if (y_shift < 0) y_shift += ny;
      if (y_shift > ny-1) y_shift -= ny;
      if (z_shift < 0) z_shift += gnz;
      if (z_shift > gnz-1) z_shift -= gnz;
      const size_t o = (size_t)x_shift + (size_t)nx*((size_t)y_shift + (size_t)ny*(size_t)z_shift);
      rho = ref[0][o]; phi = ref[1][o]; rhot = ref[2][o];
    } else {
      rho = hbar[IDX(nx,ny,nz,0,x,y,z)];
      phi = hbar[IDX(nx,ny,nz,1,x,y,z)];
      rhot = rho + phi;
    }
    double nrm[36];
    uint64_t site = (uint64_t)x + (uint64_t)nx*((uint64_t)y + (uint64_t)ny*(uint64_t)gz);
    orc_site_normals(p->seed, site, noise_index, nrm);
    /* Assignment of the site's normals to modes (project-defined, like the stream itself):
     * nrm[0..2] -> momentum modes 1..3 (gn = -fn), nrm[3 + (a-4)] -> f mode a, nrm[18 + (a-4)] -> g mode a;
     * a = 4..18.  The reference interleaves f,g draws (:124-127), which is immaterial for an i.i.d. stream.
     * Amplitudes (:117, :125-126): sqrt(c kBT |rho phi/rhot|) and sqrt(c kBT/cs2 b[a] |rho|), the latter
     * evaluated as sqrt(c kBT/cs2 b[a]) * sqrt(|rho|) -- within 2 ulp of it (tests/test_oracle_pins.py) with 3 instead of 31 square
     * roots per site; project-defined like the stream (the generated noise is pinned statistically only). */
    fn[IDX(nx,ny,nz,0,x,y,z)] = 0.;
    gn[IDX(nx,ny,nz,0,x,y,z)] = 0.;
    for (int a = 1; a <= 3; a++) {
      double v = sqrt(2.*(tau_f_bar - 0.5*tau_f_bar2)*kBT*fabs(rho*phi/rhot))*nrm[a-1];
      fn[IDX(nx,ny,nz,a,x,y,z)] = v;
      gn[IDX(nx,ny,nz,a,x,y,z)] = -v;
    }
    const double sr = sqrt(fabs(rho)), sp = sqrt(fabs(phi));
    for (int a = 4; a < Q; a++) {
      fn[IDX(nx,ny,nz,a,x,y,z)] = sqrt(2.*(tau_f_bar - 0.5*tau_f_bar2)*kBT/cs2*B[a])*sr*nrm[3 + (a-4)];
      gn[IDX(nx,ny,nz,a,x,y,z)] = sqrt(2.*(tau_g_bar - 0.5*tau_g_bar2)*kBT/cs2*B[a])*sp*nrm[18 + (a-4)];
    }
  }
}

void orc_thermal_noise_slab(const orc_params* p, int nx, int ny, int nz, int gz0, int gnz,
                            const double* hbar, uint32_t noise_index,
                            double* fn, double* gn) {
  thermal_noise_impl(p, nx, ny, nz, gz0, gnz, hbar, NULL, NULL, noise_index, fn, gn);
}

/* USE_REF_STATE branch on a slab: rho_eq, phi_eq, rhot_eq cover the GLOBAL lattice nx*ny*gnz. */
void orc_thermal_noise_ref_slab(const orc_params* p, int nx, int ny, int nz, int gz0, int gnz,
                                const double* rho_eq, const double* phi_eq, const double* rhot_eq,
                                const double* pos_com_relative, uint32_t noise_index,
                                double* fn, double* gn) {
  const double* ref[3] = { rho_eq, phi_eq, rhot_eq };
  thermal_noise_impl(p, nx, ny, nz, gz0, gnz, NULL, ref, pos_com_relative, noise_index, fn, gn);
}

void orc_thermal_noise(const orc_params* p, int nx, int ny, int nz,
                       const double* hbar, uint32_t noise_index, double* fn, double* gn) {
  orc_thermal_noise_slab(p, nx, ny, nz, 0, nz, hbar, noise_index, fn, gn);
}

/* gradient, LBM_binary.H:134-150 (use_SC_pseudo == false, :23) */
static void gradient(const orc_params* p, int nx, int ny, int nz, int x, int y, int z,
                     const double* field, int icomp, double g[3]) {
  g[0] = 0.0; g[1] = 0.0; g[2] = 0.0;
  for (int i = 0; i < Q; i++) {
    int xp = wrap(x + C[i][0], nx);
    int yp = wrap(y + C[i][1], ny);
    int zp = wrap(z + C[i][2], nz);
    double v = field[IDX(nx,ny,nz,icomp,xp,yp,zp)];
    for (int dir = 0; dir < 3; dir++) {
      g[dir] += W[i]/p->cs2*v*C[i][dir];
    }
  }
}

/* hydrovars_bar_density + LBM_hydrovars_density, LBM_binary.H:315-354 */
void orc_hydrovars_density(const orc_params* p, int nx, int ny, int nz,
                           const double* f, const double* g, double* hb) {
  (void)p;
  #pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
    double rho = 0.0, phi = 0.0;
    double fs[Q], gs[Q], mf[Q], mg[Q];
    for (int i = 0; i < Q; ++i) {
      fs[i] = f[IDX(nx,ny,nz,i,x,y,z)];
      gs[i] = g[IDX(nx,ny,nz,i,x,y,z)];
      rho += fs[i];
      phi += gs[i];
    }
    hb[IDX(nx,ny,nz,0,x,y,z)] = rho;
    hb[IDX(nx,ny,nz,1,x,y,z)] = phi;
    orc_moments(fs, mf);
    orc_moments(gs, mg);
    for (int i = 1; i <= 3; ++i) {
      hb[IDX(nx,ny,nz,i+1,x,y,z)]   = (fabs(mf[0]) > FLT_EPSILON) ? mf[i]/mf[0] : 0.;
      hb[IDX(nx,ny,nz,i+2+3,x,y,z)] = (fabs(mg[0]) > FLT_EPSILON) ? mg[i]/mg[0] : 0.;
    }
    hb[IDX(nx,ny,nz,5,x,y,z)] = mf[0] + mg[0];
  }
}

/* hydrovars + LBM_hydrovars, LBM_binary.H:196-313.  grad_laplacian_2nd (:170-194,
 * :232-235) is evaluated by the reference but its result is unused (:256-257
 * commented out), so it is omitted here. */
void orc_hydrovars(const orc_params* p, int nx, int ny, int nz,
                   const double* f, const double* g, const double* hbar,
                   const double* nf, const double* ng, double* h) {
  const double cs2 = p->cs2, alpha0 = p->alpha0, tau_f = p->tau_f, tau_g = p->tau_g;
  #pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
    double ufbar[3], ugbar[3], afbar[3], agbar[3];
    const double rho = hbar[IDX(nx,ny,nz,0,x,y,z)];
    const double phi = hbar[IDX(nx,ny,nz,1,x,y,z)];
    double jf[3] = {0.0,0.0,0.0}, jg[3] = {0.0,0.0,0.0};
    for (int i = 0; i < Q; ++i) {
      double fi = f[IDX(nx,ny,nz,i,x,y,z)];
      double gi = g[IDX(nx,ny,nz,i,x,y,z)];
      jf[0] += fi*C[i][0]; jf[1] += fi*C[i][1]; jf[2] += fi*C[i][2];
      jg[0] += gi*C[i][0]; jg[1] += gi*C[i][1]; jg[2] += gi*C[i][2];
    }
    double grad_rho[3], grad_phi[3];
    gradient(p, nx,ny,nz, x,y,z, hbar, 0, grad_rho);
    gradient(p, nx,ny,nz, x,y,z, hbar, 1, grad_phi);
    for (int k = 0; k < 3; k++) {
      ufbar[k] = (fabs(rho) > FLT_EPSILON) ? jf[k]/rho : 0.;
      ugbar[k] = (fabs(phi) > FLT_EPSILON) ? jg[k]/phi : 0.;
      afbar[k] = (fabs(rho) > FLT_EPSILON) ? -cs2*alpha0*rho*grad_phi[k]/rho : 0.;
      agbar[k] = (fabs(phi) > FLT_EPSILON) ? -cs2*alpha0*phi*grad_rho[k]/phi : 0.;
    }
    double nfvel[3], ngvel[3];
    for (int k = 0; k < 3; k++) {
      nfvel[k] = nf[IDX(nx,ny,nz,1+k,x,y,z)];
      ngvel[k] = ng[IDX(nx,ny,nz,1+k,x,y,z)];
      nfvel[k] = (fabs(rho) > FLT_EPSILON) ? nfvel[k]/rho : 0.;
      ngvel[k] = (fabs(phi) > FLT_EPSILON) ? ngvel[k]/phi : 0.;
    }
    h[IDX(nx,ny,nz,0,x,y,z)] = rho;
    h[IDX(nx,ny,nz,1,x,y,z)] = phi;
    for (int k = 0; k < 3; k++) {
      h[IDX(nx,ny,nz,2+k,x,y,z)] = ufbar[k] + 0.5*afbar[k] - 0.5/(tau_f+0.5)*phi/(rho+phi)*(ufbar[k]-ugbar[k] + 0.5*(afbar[k]-agbar[k])) + 0.5*nfvel[k];
      h[IDX(nx,ny,nz,6+k,x,y,z)] = ugbar[k] + 0.5*agbar[k] - 0.5/(tau_g+0.5)*rho/(rho+phi)*(ugbar[k]-ufbar[k] + 0.5*(agbar[k]-afbar[k])) + 0.5*ngvel[k];
    }
    double rho_tot = rho + phi;
    h[IDX(nx,ny,nz,5,x,y,z)] = rho_tot;
    for (int k = 0; k < 3; k++) {
      h[IDX(nx,ny,nz,9+k,x,y,z)]  = afbar[k];
      h[IDX(nx,ny,nz,12+k,x,y,z)] = agbar[k];
      h[IDX(nx,ny,nz,15+k,x,y,z)] = (rho*ufbar[k] + phi*ugbar[k] + 0.5*(rho*afbar[k] + phi*agbar[k]))/rho_tot;
    }
    h[IDX(nx,ny,nz,18,x,y,z)] = nfvel[0];
    h[IDX(nx,ny,nz,19,x,y,z)] = ngvel[0];
    h[IDX(nx,ny,nz,20,x,y,z)] = ufbar[0];
    h[IDX(nx,ny,nz,21,x,y,z)] = ugbar[0];
  }
}

/* equilibrium_moments, LBM_binary.H:356-402 */
static void equilibrium_moments(const orc_params* p, double rho, const double u[3], double mEq[Q]) {
  const double cs2 = p->cs2, cs4 = p->cs4;
  const double coefA = rho;
  const double coefB = 1.;
  const double coefAB = coefA*coefB;
  const double coefC = 1./cs2;
  double AD[3][3];
  AD[0][0] = (coefA*u[0]*u[0])/2./cs4;
  AD[0][1] = (coefA*u[0]*u[1])/2./cs4;
  AD[0][2] = (coefA*u[0]*u[2])/2./cs4;
  AD[1][0] = AD[0][1];
  AD[1][1] = (coefA*u[1]*u[1])/2./cs4;
  AD[1][2] = (coefA*u[1]*u[2])/2./cs4;
  AD[2][0] = AD[0][2]; AD[2][1] = AD[1][2];
  AD[2][2] = (coefA*u[2]*u[2])/2./cs4;
  const double tr = AD[0][0] + AD[1][1] + AD[2][2];
  mEq[0] = coefAB;
  mEq[1] = coefC*cs2*(coefA*u[0]);
  mEq[2] = coefC*cs2*(coefA*u[1]);
  mEq[3] = coefC*cs2*(coefA*u[2]);
  mEq[4] = 2.*cs4*tr;
  mEq[5] = 6.*cs4*AD[0][0] - 2.*cs4*tr;
  mEq[6] = 2.*cs4*(AD[1][1] - AD[2][2]);
  mEq[7] = cs4*(AD[0][1] + AD[1][0]);
  mEq[8] = cs4*(AD[1][2] + AD[2][1]);
  mEq[9] = cs4*(AD[0][2] + AD[2][0]);
  for (int a = 10; a < Q; ++a) mEq[a] = 0.;
}

/* phi_moments, LBM_binary.H:404-449 */
static void phi_moments(const orc_params* p, double rho, const double u[3], const double a[3], double mEq[Q]) {
  const double cs2 = p->cs2, cs4 = p->cs4;
  const double coefA = rho;
  const double coefB = 0.;
  const double coefAB = coefA*coefB;
  const double coefC = 1./cs2;
  const double coefAC = coefA*coefC;
  double AD[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) AD[i][j] = a[i]*(coefA*u[j])/cs4;
  const double tr = AD[0][0] + AD[1][1] + AD[2][2];
  const double modifactor = 1./(1.+1./(2.*p->tau_f));   /* tau_f for both fluids (:424) */
  mEq[0] = modifactor*coefAB;
  mEq[1] = modifactor*coefAC*cs2*a[0];
  mEq[2] = modifactor*coefAC*cs2*a[1];
  mEq[3] = modifactor*coefAC*cs2*a[2];
  mEq[4] = modifactor*2.*cs4*tr;
  mEq[5] = modifactor*(6.*cs4*AD[0][0] - 2.*cs4*tr);
  mEq[6] = modifactor*2.*cs4*(AD[1][1] - AD[2][2]);
  mEq[7] = modifactor*cs4*(AD[0][1] + AD[1][0]);
  mEq[8] = modifactor*cs4*(AD[1][2] + AD[2][1]);
  mEq[9] = modifactor*cs4*(AD[0][2] + AD[2][0]);
  for (int k = 10; k < Q; ++k) mEq[k] = 0.;
}

/* collide, LBM_binary.H:451-516 -- in place on f,g at every site */
void orc_collide(const orc_params* p, int nx, int ny, int nz,
                 double* f, double* g, const double* h,
                 const double* fn, const double* gn) {
  const double tau_f_bar = p->tau_f*(1.+0.5/p->tau_f);
  const double tau_g_bar = p->tau_g*(1.+0.5/p->tau_g);
  #pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
    const double fields[2] = { h[IDX(nx,ny,nz,0,x,y,z)], h[IDX(nx,ny,nz,1,x,y,z)] };
    double uf[3], ug[3], af[3], ag[3], v_b[3];
    for (int k = 0; k < 3; ++k) {
      uf[k] = h[IDX(nx,ny,nz,2+k,x,y,z)];
      ug[k] = h[IDX(nx,ny,nz,6+k,x,y,z)];
      af[k] = h[IDX(nx,ny,nz,9+k,x,y,z)];
      ag[k] = h[IDX(nx,ny,nz,12+k,x,y,z)];
    }
    double fs[Q], gs[Q], mf[Q], mg[Q], mfEq[Q], mgEq[Q], mPhif[Q], mPhig[Q];
    for (int i = 0; i < Q; ++i) { fs[i] = f[IDX(nx,ny,nz,i,x,y,z)]; gs[i] = g[IDX(nx,ny,nz,i,x,y,z)]; }
    orc_moments(fs, mf);
    orc_moments(gs, mg);
    for (int k = 0; k < 3; ++k) v_b[k] = (fields[0]*uf[k] + fields[1]*ug[k])/(fields[0] + fields[1]);
    equilibrium_moments(p, fields[0], v_b, mfEq);
    equilibrium_moments(p, fields[1], v_b, mgEq);
    phi_moments(p, fields[0], uf, af, mPhif);
    phi_moments(p, fields[1], ug, ag, mPhig);
    for (int a = 0; a < Q; ++a) {
      double Raf = 1./tau_f_bar * (mfEq[a] - mf[a]) + mPhif[a] + fn[IDX(nx,ny,nz,a,x,y,z)];
      double Rag = 1./tau_g_bar * (mgEq[a] - mg[a]) + mPhig[a] + gn[IDX(nx,ny,nz,a,x,y,z)];
      mf[a] = mf[a] + Raf;
      mg[a] = mg[a] + Rag;
    }
    orc_populations(mf, fs);
    orc_populations(mg, gs);
    for (int i = 0; i < Q; ++i) { f[IDX(nx,ny,nz,i,x,y,z)] = fs[i]; g[IDX(nx,ny,nz,i,x,y,z)] = gs[i]; }
  }
}

/* stream_push, LBM_binary.H:519-531, with periodic wrap instead of ghost cells */
void orc_stream_push(int nx, int ny, int nz, const double* fold, const double* gold,
                     double* fnew, double* gnew) {
  for (int i = 0; i < Q; ++i)
    #pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
      int xp = wrap(x + C[i][0], nx), yp = wrap(y + C[i][1], ny), zp = wrap(z + C[i][2], nz);
      fnew[IDX(nx,ny,nz,i,xp,yp,zp)] = fold[IDX(nx,ny,nz,i,x,y,z)];
      gnew[IDX(nx,ny,nz,i,xp,yp,zp)] = gold[IDX(nx,ny,nz,i,x,y,z)];
    }
}

/* The tail shared by LBM_timestep (:583-592) and every LBM_init_* (:621-627 etc.):
 * densities -> noise -> real hydrodynamic variables. */
void orc_refresh(const orc_params* p, int nx, int ny, int nz, uint32_t noise_index,
                 const double* f, const double* g, double* hbar, double* fn, double* gn, double* h) {
  orc_hydrovars_density(p, nx,ny,nz, f, g, hbar);
  orc_thermal_noise(p, nx,ny,nz, hbar, noise_index, fn, gn);
  orc_hydrovars(p, nx,ny,nz, f, g, hbar, fn, gn, h);
}

/* LBM_timestep, LBM_binary.H:545-594.  f,g hold state t on entry and t+1 on
 * return (valid cells); ftmp,gtmp are scratch (the reference's fnew,gnew).
 * step_done = number of steps completed BEFORE this call; the noise drawn at
 * the end of this call gets index step_done+1.  inject != 0: keep the caller's
 * fn/gn (tests feed identical noise to both implementations). */
void orc_timestep(const orc_params* p, int nx, int ny, int nz, uint32_t step_done,
                  double* f, double* g, double* ftmp, double* gtmp,
                  double* hbar, double* fn, double* gn, double* h) {
  size_t n = (size_t)Q*nx*ny*nz*sizeof(double);
  orc_collide(p, nx,ny,nz, f, g, h, fn, gn);
  orc_stream_push(nx,ny,nz, f, g, ftmp, gtmp);
  memcpy(f, ftmp, n); memcpy(g, gtmp, n);        /* MultiFab::Swap, :579-580 */
  orc_refresh(p, nx,ny,nz, step_done+1, f, g, hbar, fn, gn, h);
}

/* The same tail with the USE_REF_STATE noise branch; pos_com_relative as the caller of
 * thermal_noise passes it (:588-590 relative to com_ref[0]; :623-625 absolute in LBM_init_mixture;
 * zero in LBM_init_stripe/_droplet, :690, :739). */
void orc_refresh_ref(const orc_params* p, int nx, int ny, int nz, uint32_t noise_index,
                     const double* f, const double* g, double* hbar, double* fn, double* gn, double* h,
                     const double* rho_eq, const double* phi_eq, const double* rhot_eq,
                     const double* pos_com_relative) {
  orc_hydrovars_density(p, nx,ny,nz, f, g, hbar);
  orc_thermal_noise_ref_slab(p, nx,ny,nz, 0, nz, rho_eq, phi_eq, rhot_eq, pos_com_relative, noise_index, fn, gn);
  orc_hydrovars(p, nx,ny,nz, f, g, hbar, fn, gn, h);
}

void orc_update_com(int nx, int ny, int nz, const double* hbar, double com[3]);

/* LBM_timestep with USE_REF_STATE defined, LBM_binary.H:545-594 incl. :585-590. */
void orc_timestep_ref(const orc_params* p, int nx, int ny, int nz, uint32_t step_done,
                      double* f, double* g, double* ftmp, double* gtmp,
                      double* hbar, double* fn, double* gn, double* h,
                      const double* rho_eq, const double* phi_eq, const double* rhot_eq,
                      const double* com_ref) {
  size_t n = (size_t)Q*nx*ny*nz*sizeof(double);
  orc_collide(p, nx,ny,nz, f, g, h, fn, gn);
  orc_stream_push(nx,ny,nz, f, g, ftmp, gtmp);
  memcpy(f, ftmp, n); memcpy(g, gtmp, n);
  orc_hydrovars_density(p, nx,ny,nz, f, g, hbar);
  double rel[3];
  orc_update_com(nx,ny,nz, hbar, rel);               /* :587 */
  for (int k = 0; k < 3; ++k) rel[k] -= com_ref[k];  /* :588 */
  orc_thermal_noise_ref_slab(p, nx,ny,nz, 0, nz, rho_eq, phi_eq, rhot_eq, rel, step_done+1, fn, gn);
  orc_hydrovars(p, nx,ny,nz, f, g, hbar, fn, gn, h);
}

/* Split form for injected-noise tests: collide+stream+densities only. */
void orc_collide_stream(const orc_params* p, int nx, int ny, int nz,
                        double* f, double* g, double* ftmp, double* gtmp,
                        const double* h, const double* fn, const double* gn) {
  size_t n = (size_t)Q*nx*ny*nz*sizeof(double);
  orc_collide(p, nx,ny,nz, f, g, h, fn, gn);
  orc_stream_push(nx,ny,nz, f, g, ftmp, gtmp);
  memcpy(f, ftmp, n); memcpy(g, gtmp, n);
}

/* ------------------------------------------------------------------ */
/* Initial populations (the ParallelFor bodies of the LBM_init_* functions). */

/* LBM_init_mixture, LBM_binary.H:606-618 */
void orc_init_mixture(const orc_params* p, int nx, int ny, int nz, double* f, double* g) {
  (void)p;
  const double C1 = 0.5, C2 = 0.5;
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
    const double rho = 2.*C1;
    const double phi = 2.*C2;
    for (int i = 0; i < Q; i++) {
      f[IDX(nx,ny,nz,i,x,y,z)] = W[i]*rho;
      g[IDX(nx,ny,nz,i,x,y,z)] = W[i]*phi;
    }
  }
}

/* LBM_init_stripe, LBM_binary.H:664-686 */
void orc_init_stripe(const orc_params* p, int nx, int ny, int nz, double frac, double* f, double* g) {
  const double rho_t = p->rho_hi + p->rho_lo;
  const double pos_lo = (-0.5*frac)*nz;
  const double pos_hi = (0.5*frac)*nz;
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
    const double pos = z - nz/2;                 /* integer division, :680 */
    const double rho = (p->rho_hi - p->rho_lo)*0.5*(tanh((pos-pos_lo)/sqrt(p->kappa)) + tanh((pos_hi-pos)/sqrt(p->kappa))) + p->rho_lo;
    for (int i = 0; i < Q; i++) {
      f[IDX(nx,ny,nz,i,x,y,z)] = W[i]*rho;
      g[IDX(nx,ny,nz,i,x,y,z)] = W[i]*(rho_t-rho);
    }
  }
}

/* LBM_init_droplet, LBM_binary.H:699-737 */
void orc_init_droplet(const orc_params* p, int nx, int ny, int nz, double r_frac, double* f, double* g) {
  const double R = r_frac*nx;                    /* box[0], :714 */
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
    const double rx = x - nx/2.;
    const double ry = y - ny/2.;
    const double rz = z - nx/2;                  /* box[0], integer division, :725 (sic) */
    const double r2 = rx*rx + ry*ry + rz*rz;
    const double r = sqrt(r2);
    const double rho_tot = p->rho_hi + p->rho_lo;
    const double rho = (p->rho_hi - p->rho_lo)*(1.+tanh((R-r)/sqrt(p->kappa)))/2. + p->rho_lo;
    for (int i = 0; i < Q; i++) {
      f[IDX(nx,ny,nz,i,x,y,z)] = W[i]*rho;
      g[IDX(nx,ny,nz,i,x,y,z)] = W[i]*(rho_tot - rho);
    }
  }
}

/* update_com, LBM_hydrovs.H:26-60: sums of rho*i, rho*j, rho*k over valid cells
 * (cell indices, not centres) divided by the total mass of comp 0. */
void orc_update_com(int nx, int ny, int nz, const double* hbar, double com[3]) {
  double mass = 0., sx = 0., sy = 0., sz = 0.;
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
    double r = hbar[IDX(nx,ny,nz,0,x,y,z)];
    mass += r; sx += r*x; sy += r*y; sz += r*z;
  }
  com[0] = sx/mass; com[1] = sy/mass; com[2] = sz/mass;
}

/* ------------------------------------------------------------------ */
/* cpu_baseline leg of bench.py: time nsteps of the restated LBM_timestep on a
 * stripe-initialised box; returns seconds of the timed steps. */
#include <time.h>
double orc_bench(const orc_params* p, int nx, int ny, int nz, int nsteps, double* checksum) {
  size_t ns = (size_t)nx*ny*nz;
  double *f = malloc(Q*ns*8), *g = malloc(Q*ns*8), *ft = malloc(Q*ns*8), *gt = malloc(Q*ns*8);
  double *fn = malloc(Q*ns*8), *gn = malloc(Q*ns*8), *hb = calloc(NHBAR*ns,8), *h = malloc(NHYDRO*ns*8);
  orc_init_stripe(p, nx,ny,nz, 0.5, f, g);
  orc_refresh(p, nx,ny,nz, 0, f, g, hb, fn, gn, h);
  orc_timestep(p, nx,ny,nz, 0, f, g, ft, gt, hb, fn, gn, h);   /* warm-up */
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int s = 0; s < nsteps; ++s) orc_timestep(p, nx,ny,nz, 1+s, f, g, ft, gt, hb, fn, gn, h);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  double sum = 0.; for (size_t i = 0; i < ns; ++i) sum += hb[i];
  if (checksum) *checksum = sum;
  free(f); free(g); free(ft); free(gt); free(fn); free(gn); free(hb); free(h);
  return (t1.tv_sec - t0.tv_sec) + 1e-9*(t1.tv_nsec - t0.tv_nsec);
}
