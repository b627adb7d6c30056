// bflbm_kernels.h -- HIP kernels (gfx950) of the collide-and-stream path.
//
// Resident state: S[c][p][y][x], c = 0..18 fluid f, 19..37 fluid g, x fastest,
// holding the POST-COLLISION populations; the reference's post-stream state
// (fold/gold after LBM_timestep) is recovered by the pull  f_i(x) = S_i(x - c_i),
// which the survey verified to be bit-identical to the reference's
// collide-on-(valid+1 ghost)-then-push (LBM_binary.H:519-542).
// p is the storage plane: p = z - z0 + H, H halo planes per side (0 when the
// slab is the whole periodic box and z wraps inside the kernel).
#ifndef BFLBM_KERNELS_H_
#define BFLBM_KERNELS_H_

#include "bflbm_site.h"

// Layout of the x direction.  BFLBM_XSHIFT 1 ("half-streamed"): population i of logical site x is stored at x + c_ix
// (periodic), i.e. it is streamed in x when it is STORED and in y,z when it is pulled: every pull reads x-aligned row
// segments (the line of the neighbouring tile that an x-shifted segment touches was 9 % of the reads) and
// every store of a population with c_ix != 0 is shifted by one element instead.  Pure data movement: the same doubles.
#ifndef BFLBM_XSHIFT
#define BFLBM_XSHIFT 0
#endif
#define BFLBM_PX(cx) (BFLBM_XSHIFT ? 0 : -(cx))      /* x displacement of the pull of a population with c_x = cx */
#define BFLBM_SX(cx) (BFLBM_XSHIFT ? (cx) : 0)       /* x displacement of its store */

struct Geo {
  int nx, ny, nzs;     // storage extent (nzs includes the halo planes)
  int zwrap;           // 1: plane neighbours wrap modulo nzs (single slab)
  int H;               // halo planes per side
  int z0;              // global z of storage plane H
  int nz;              // global nz
  int pitch;           // row stride of the resident arrays in elements: nx rounded up to 16 (128-byte rows)
  long long plane;     // pitch*ny: plane stride of the resident arrays (S, rho, phi)
  long long dplane;    // nx*ny: plane stride of dense arrays (observable outputs, injected noise, host tables)
  long long vol;       // nzs*plane + pad = component stride
};

// compile-time velocity tables for fully unrolled loops
struct Vel {
  static constexpr int cx[Q] = BFLBM_CX;
  static constexpr int cy[Q] = BFLBM_CY;
  static constexpr int cz[Q] = BFLBM_CZ;
};

struct SiteIdx {
  int x, y, p;
  int xm, xp;          // x-1, x+1 wrapped
  long long row[3][3]; // row[dz+1][dy+1] = (p+dz)*plane + (y+dy wrapped)*nx
};

__device__ __forceinline__ void site_index(const Geo& G, int x, int y, int p, SiteIdx& I) {
  I.x = x; I.y = y; I.p = p;
  I.xm = (x == 0) ? G.nx - 1 : x - 1;
  I.xp = (x == G.nx - 1) ? 0 : x + 1;
  const int ym = (y == 0) ? G.ny - 1 : y - 1;
  const int yp = (y == G.ny - 1) ? 0 : y + 1;
  int pm = p - 1, pp = p + 1;
  if (G.zwrap) { if (pm < 0) pm = G.nzs - 1; if (pp >= G.nzs) pp = 0; }
  const int ys[3] = { ym, y, yp };
  const int ps[3] = { pm, p, pp };
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) I.row[a][b] = (long long)ps[a]*G.plane + (long long)ys[b]*G.pitch;
}

// offset of the site displaced by (dx,dy,dz) in {-1,0,1}^3 (compile-time constants after unrolling)
__device__ __forceinline__ long long nb_off(const SiteIdx& I, int dx, int dy, int dz) {
  const int xx = dx > 0 ? I.xp : (dx < 0 ? I.xm : I.x);
  return I.row[dz+1][dy+1] + xx;
}

#ifndef BFLBM_COLLIDE_NT_STORES
#define BFLBM_COLLIDE_NT_STORES 0   // non-temporal population stores in k_collide: +3.7 % at 32^3, +1 ... -2 % elsewhere (profiles/r04_nt_hints.txt): off
#endif
__device__ __forceinline__ void st_pop(double* __restrict__ base, unsigned boff, double v) {   // population stores of k_collide
  double* q = reinterpret_cast<double*>(reinterpret_cast<char*>(base) + boff);
  if (BFLBM_COLLIDE_NT_STORES) __builtin_nontemporal_store(v, q); else *q = v;
}
// f_i(x) = S_i(x - c_i)
__device__ __forceinline__ void pull_site(const double* __restrict__ S, const Geo& G, const SiteIdx& I,
                                          double (&fs)[Q], double (&gs)[Q]) {
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const long long o = nb_off(I, BFLBM_PX(Vel::cx[i]), -Vel::cy[i], -Vel::cz[i]);
    fs[i] = S[(long long)i*G.vol + o];
    gs[i] = S[(long long)(i+Q)*G.vol + o];
  }
}

// nb[i] = field(x + c_i)
__device__ __forceinline__ void gather_field(const double* __restrict__ fld, const SiteIdx& I, double (&nb)[Q]) {
#pragma unroll
  for (int i = 0; i < Q; ++i) nb[i] = fld[nb_off(I, Vel::cx[i], Vel::cy[i], Vel::cz[i])];
}

__device__ __forceinline__ uint64_t global_site(const Geo& G, int x, int y, int p) {
  int gz = G.z0 + (p - G.H);
  if (gz < 0) gz += G.nz;
  if (gz >= G.nz) gz -= G.nz;
  return (uint64_t)x + (uint64_t)G.nx*((uint64_t)y + (uint64_t)G.ny*(uint64_t)gz);
}

// USE_REF_STATE (LBM_binary.H:92-107): noise amplitudes from equilibrium fields covering the GLOBAL
// lattice, looked up at the site shifted by the truncated relative centre of mass.
struct RefState {
  const double* rho; const double* phi; const double* rhot;
  int on;
  int sx, sy, sz;      // static_cast<int>(pos_com_relative[d])
};
__device__ __forceinline__ void noise_state(const RefState& Rf, const Geo& G, int x, int y, int p,
                                            double r, double ph, double& ar, double& ap, double& at) {
  if (!Rf.on) { ar = r; ap = ph; at = r + ph; return; }
  int gz = G.z0 + (p - G.H);
  if (gz < 0) gz += G.nz;
  if (gz >= G.nz) gz -= G.nz;
  int xs = x - Rf.sx, ys = y - Rf.sy, zs = gz - Rf.sz;
  if (xs < 0) xs += G.nx;                        // one wrap each way, like the reference (:98-103)
  if (xs > G.nx - 1) xs -= G.nx;
  if (ys < 0) ys += G.ny;
  if (ys > G.ny - 1) ys -= G.ny;
  if (zs < 0) zs += G.nz;
  if (zs > G.nz - 1) zs -= G.nz;
  const long long o = (long long)xs + (long long)G.nx*((long long)ys + (long long)G.ny*(long long)zs);
  ar = Rf.rho[o]; ap = Rf.phi[o]; at = Rf.rhot[o];
}

// The same addressing for the hot two-pass kernels as in the fused kernel: wave-uniform base (component
// volume + plane, 64-bit scalar) + 32-bit per-lane BYTE offset inside the plane, so that loads and stores
// take the scalar-base form.  The nine (dy,dx) offsets are made opaque in the block that uses them, or the
// loop-invariant zero-extensions are hoisted and instruction selection falls back to 64-bit lane adds.
struct SiteOff {
  long long pl[3];     // (p+dz) * plane, dz = -1, 0, 1 (elements; uniform over the workgroup)
  unsigned o[3][3];    // ((y+dy wrapped) * nx + (x+dx wrapped)) * 8
};
__device__ __forceinline__ void site_offsets(const Geo& G, int x, int y, int p, SiteOff& I) {
  const int xm = (x == 0) ? G.nx - 1 : x - 1, xp = (x == G.nx - 1) ? 0 : x + 1;
  const int ym = (y == 0) ? G.ny - 1 : y - 1, yp = (y == G.ny - 1) ? 0 : y + 1;
  int pm = p - 1, pp = p + 1;
  if (G.zwrap) { if (pm < 0) pm = G.nzs - 1; if (pp >= G.nzs) pp = 0; }
  const int xs[3] = { xm, x, xp }, ys[3] = { ym, y, yp }, ps[3] = { pm, p, pp };
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    I.pl[a] = (long long)ps[a]*G.plane;
#pragma unroll
    for (int b = 0; b < 3; ++b) { I.o[a][b] = ((unsigned)(ys[a]*G.pitch) + (unsigned)xs[b]) * 8u; asm volatile("" : "+v"(I.o[a][b])); }
  }
}
__device__ __forceinline__ double ld_sb(const double* __restrict__ base, unsigned boff) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + boff);
}
__device__ __forceinline__ void st_sb(double* __restrict__ base, unsigned boff, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + boff) = v;
}
// f_i(x) = S_i(x - c_i)
__device__ __forceinline__ void pull_site(const double* __restrict__ S, const Geo& G, const SiteOff& I,
                                          double (&fs)[Q], double (&gs)[Q]) {
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const double* __restrict__ b = S + (long long)i*G.vol + I.pl[1 - Vel::cz[i]];
    const unsigned o = I.o[1 - Vel::cy[i]][1 + BFLBM_PX(Vel::cx[i])];
    fs[i] = ld_sb(b, o);
    gs[i] = ld_sb(b + (long long)Q*G.vol, o);
  }
}
// nb[i] = field(x + c_i)
__device__ __forceinline__ void gather_field(const double* __restrict__ fld, const SiteOff& I, double (&nb)[Q]) {
#pragma unroll
  for (int i = 0; i < Q; ++i) nb[i] = ld_sb(fld + I.pl[1 + Vel::cz[i]], I.o[1 + Vel::cy[i]][1 + Vel::cx[i]]);
}

// Workgroups are dispatched round-robin over the 8 XCDs (workgroup b -> XCD b % 8), each with its own L2.  With
// the identity mapping consecutive rows of a plane land on different XCDs and every XCD fetches the rho,phi
// rows of its y-neighbours itself (measured: k_collide read 1.36x its algorithmic bytes).  This mapping gives
// each XCD a contiguous band of rows of every plane (grid.x % 8 == 0; identity otherwise).
#define BFLBM_SITE_FROM_BLOCK_XCD()                                     \
  int bx_ = (int)blockIdx.x, by_ = (int)blockIdx.y;                     \
  if ((gridDim.x & 7u) == 0u) {                                         \
    const unsigned b_ = blockIdx.y*gridDim.x + blockIdx.x, band_ = gridDim.x >> 3;   \
    const unsigned xcd_ = b_ & 7u, j_ = b_ >> 3;                        \
    by_ = (int)(j_ / band_); bx_ = (int)(xcd_*band_ + j_ % band_);      \
  }                                                                     \
  const long long s_ = (long long)bx_*blockDim.x + threadIdx.x;         \
  if (s_ >= G.plane) return;                                            \
  const int p = p0 + by_;                                               \
  const int y = (int)(s_ / G.pitch);                                    \
  const int x = (int)(s_ - (long long)y*G.pitch);                       \
  if (x >= G.nx) return;                                                /* row padding */

#define BFLBM_SITE_FROM_BLOCK()                                         \
  const long long s_ = (long long)blockIdx.x*blockDim.x + threadIdx.x;  \
  if (s_ >= G.plane) return;                                            \
  const int p = p0 + (int)blockIdx.y;                                   \
  const int y = (int)(s_ / G.pitch);                                    \
  const int x = (int)(s_ - (long long)y*G.pitch);                       \
  if (x >= G.nx) return;                                                /* row padding */

// ---- pass A of the two-pass schedule: rho,phi of the streamed state (LBM_binary.H:320-330)
__global__ void __launch_bounds__(256) k_density(const double* __restrict__ S, double* __restrict__ rho,
                                                 double* __restrict__ phi, Geo G, int p0) {
  BFLBM_SITE_FROM_BLOCK();
  SiteOff I; site_offsets(G, x, y, p, I);
  double fs[Q], gs[Q];
  pull_site(S, G, I, fs, gs);
  st_sb(rho + I.pl[1], I.o[1][1], d_density(fs));
  st_sb(phi + I.pl[1], I.o[1][1], d_density(gs));
}

// rho,phi of an UPLOADED state: N holds the reference's streamed populations f_i(x) at the site's own slot (bflbm_upload_fg),
// so the sums need no neighbour (LBM_binary.H:320-330 on the arrays LBM_init copies in, :643-646)
__global__ void __launch_bounds__(256) k_density_streamed(const double* __restrict__ N, double* __restrict__ rho,
                                                          double* __restrict__ phi, Geo G, int p0) {
  BFLBM_SITE_FROM_BLOCK();
  const long long o = (long long)p*G.plane + s_;
  double fs[Q], gs[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) { fs[i] = N[(long long)i*G.vol + o]; gs[i] = N[(long long)(i+Q)*G.vol + o]; }
  rho[o] = d_density(fs);
  phi[o] = d_density(gs);
}

// ---- pass B: pull, project (hydrovars), draw noise, collide, store post-collision state
#ifndef BFLBM_COLLIDE_WAVES
#define BFLBM_COLLIDE_WAVES 2
#endif
template <bool NOISE, bool INJECT>
__global__ void __launch_bounds__(256, BFLBM_COLLIDE_WAVES) k_collide(const double* __restrict__ S, double* __restrict__ D,
                                                 const double* __restrict__ rho, const double* __restrict__ phi,
                                                 const double* __restrict__ injf, const double* __restrict__ injg,
                                                 Geo G, DevParams P, int p0, uint32_t noise_index, RefState Rf) {
  __shared__ double ntab[(NOISE && !INJECT) ? BFLBM_NORMAL_TABLE_N : 4];
  if (NOISE && !INJECT) d_load_normal_table(ntab, true);
  BFLBM_SITE_FROM_BLOCK_XCD();
  SiteOff I; site_offsets(G, x, y, p, I);
  double fs[Q], gs[Q];
  pull_site(S, G, I, fs, gs);
  const double r = ld_sb(rho + I.pl[1], I.o[1][1]), ph = ld_sb(phi + I.pl[1], I.o[1][1]);
  double nb[Q], grad_rho[3], grad_phi[3];
  gather_field(rho, I, nb); d_gradient(P, nb, grad_rho);
  gather_field(phi, I, nb); d_gradient(P, nb, grad_phi);
  double* __restrict__ Dp = D + I.pl[1];
  unsigned o = I.o[1][1];
  // noise: the momentum modes first (the projection needs them), each fluid's other modes right before
  // its relaxation; every fluid is stored as soon as it is collided -- keeps the live set small
  const long long nvol = (long long)(G.nzs - 2*G.H)*G.dplane;         // injected arrays, dense: [a][p-H][y][x]
  const long long no = (long long)(p - G.H)*G.dplane + (long long)y*G.nx + x;
  double fn3[3] = {0., 0., 0.}, gn3[3] = {0., 0., 0.};
  NoiseAmp NA; bflbm_rng_state rst;
  if (INJECT) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { fn3[k] = injf[(1 + k)*nvol + no]; gn3[k] = injg[(1 + k)*nvol + no]; }
  } else if (NOISE) {
    double ar, ap, at;
    noise_state(Rf, G, x, y, p, r, ph, ar, ap, at);
    d_noise_amp(P, ar, ap, at, NA);
    d_noise_head(P, NA, global_site(G, x, y, p), noise_index, ntab, rst, fn3);
#pragma unroll
    for (int k = 0; k < 3; ++k) gn3[k] = -fn3[k];
  }
  SiteHydro Hy;
  SiteRecip R;
  d_site_recips(P, r, ph, R);
  {
    double jf[3], jg[3];
    d_momentum(fs, jf);
    d_momentum(gs, jg);
    d_hydrovars_j(P, jf, jg, r, ph, grad_rho, grad_phi, fn3, gn3, Hy, R);
  }
  double v_b[3];
  d_barycentric(r, ph, Hy, v_b, R);
  {
    double m[Q];
    d_moments(fs, m);
    if (INJECT) {
      double fn[Q];
#pragma unroll
      for (int a = 0; a < Q; ++a) fn[a] = injf[a*nvol + no];
      d_relax<true>(P, m, r, v_b, Hy.uf, Hy.af, P.inv_tau_f_bar, fn, R.cs4);
    } else if (NOISE) {
      d_relax_generated(P, m, r, v_b, Hy.uf, Hy.af, P.inv_tau_f_bar, fn3, NA.sr, ntab, rst, R.cs4);
    } else {
      const double zn[Q] = {0.};
      d_relax<false>(P, m, r, v_b, Hy.uf, Hy.af, P.inv_tau_f_bar, zn, R.cs4);
    }
    d_populations(m, fs);
#pragma unroll
    for (int i = 0; i < Q; ++i) st_pop(Dp + (long long)i*G.vol, BFLBM_XSHIFT ? I.o[1][1 + BFLBM_SX(Vel::cx[i])] : o, fs[i]);
  }
  {
    double m[Q];
    d_moments(gs, m);
    if (INJECT) {
      double gn[Q];
#pragma unroll
      for (int a = 0; a < Q; ++a) gn[a] = injg[a*nvol + no];
      d_relax<true>(P, m, ph, v_b, Hy.ug, Hy.ag, P.inv_tau_g_bar, gn, R.cs4);
    } else if (NOISE) {
      d_relax_generated(P, m, ph, v_b, Hy.ug, Hy.ag, P.inv_tau_g_bar, gn3, NA.sp, ntab, rst, R.cs4);
    } else {
      const double zn[Q] = {0.};
      d_relax<false>(P, m, ph, v_b, Hy.ug, Hy.ag, P.inv_tau_g_bar, zn, R.cs4);
    }
    d_populations(m, gs);
#pragma unroll
    for (int i = 0; i < Q; ++i) st_pop(Dp + (long long)(i+Q)*G.vol, BFLBM_XSHIFT ? I.o[1][1 + BFLBM_SX(Vel::cx[i])] : o, gs[i]);
  }
}

// ---- natural (post-stream) populations of the slab's own planes: N_i(x) = S_i(x - c_i)
__global__ void __launch_bounds__(256) k_pull(const double* __restrict__ S, double* __restrict__ N, Geo G, int p0) {
  BFLBM_SITE_FROM_BLOCK();
  SiteIdx I; site_index(G, x, y, p, I);
  double fs[Q], gs[Q];
  pull_site(S, G, I, fs, gs);
  const long long o = I.row[1][1] + x;
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    N[(long long)i*G.vol + o] = fs[i];
    N[(long long)(i+Q)*G.vol + o] = gs[i];
  }
}

// ---- inverse: S_i(x) = N_i(x + c_i)  (upload of fold/gold, LBM_binary.H:643)
__global__ void __launch_bounds__(256) k_unstream(const double* __restrict__ N, double* __restrict__ S, Geo G, int p0) {
  BFLBM_SITE_FROM_BLOCK();
  SiteIdx I; site_index(G, x, y, p, I);
  const long long o = I.row[1][1] + x;
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const long long src = nb_off(I, BFLBM_XSHIFT ? 0 : Vel::cx[i], Vel::cy[i], Vel::cz[i]);   // the slot of site x - c_x: N_i there + c_i
    S[(long long)i*G.vol + o] = N[(long long)i*G.vol + src];
    S[(long long)(i+Q)*G.vol + o] = N[(long long)(i+Q)*G.vol + src];
  }
}

// ---- initial state: f_i = w_i rho, g_i = w_i phi (LBM_binary.H:615-618, :682-685, :733-736)
// stored un-streamed: S_i(x) = w_i rho(x + c_i).  mode 0: uniform rho_c/phi_c; mode 1: rho from a
// per-plane table (ext plane index p+1+dz); mode 2: rho from a full field with nzs+2 planes.
__global__ void __launch_bounds__(256) k_init(double* __restrict__ S, const double* __restrict__ rho_ext,
                                              Geo G, int mode, double rho_c, double phi_c, double rho_t, int p0) {
  BFLBM_SITE_FROM_BLOCK();
  const int xm = (x == 0) ? G.nx - 1 : x - 1, xp = (x == G.nx - 1) ? 0 : x + 1;
  const int ym = (y == 0) ? G.ny - 1 : y - 1, yp = (y == G.ny - 1) ? 0 : y + 1;
  const long long o = (long long)p*G.plane + (long long)y*G.pitch + x;
  const double w0 = 1./3., w1 = 1./18., w2 = 1./36.;
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const double w = (i == 0) ? w0 : (i < 7 ? w1 : w2);
    double r, ph;
    if (mode == 0) { r = rho_c; ph = phi_c; }
    else {
      const int pe = p + 1 + Vel::cz[i];
      if (mode == 1) r = rho_ext[pe];
      else {
        const int xx = BFLBM_XSHIFT ? x : (Vel::cx[i] > 0 ? xp : (Vel::cx[i] < 0 ? xm : x));
        const int yy = Vel::cy[i] > 0 ? yp : (Vel::cy[i] < 0 ? ym : y);
        r = rho_ext[(long long)pe*G.dplane + (long long)yy*G.nx + xx];
      }
      ph = rho_t - r;
    }
    S[(long long)i*G.vol + o] = w*r;
    S[(long long)(i+Q)*G.vol + o] = w*ph;
  }
}

// ---- materialise hydrovsbar / noise / hydrovs for the streamed state (own planes, dense output)
// what: 0 hydrovsbar[9] (LBM_binary.H:315-340), 1 noise f,g [38] (:73-132), 2 hydrovs[ncomp] (:196-295)
template <int WHAT>
__global__ void __launch_bounds__(256) k_observe(const double* __restrict__ S, const double* __restrict__ rho,
                                                 const double* __restrict__ phi, const double* __restrict__ injf,
                                                 const double* __restrict__ injg, double* __restrict__ out,
                                                 Geo G, DevParams P, int p0, uint32_t noise_index, int ncomp, int inject,
                                                 RefState Rf) {
  __shared__ double ntab[WHAT != 0 ? BFLBM_NORMAL_TABLE_N : 4];
  if (WHAT != 0) d_load_normal_table(ntab, P.noise_on && !inject);
  BFLBM_SITE_FROM_BLOCK();
  SiteIdx I; site_index(G, x, y, p, I);
  double fs[Q], gs[Q];
  pull_site(S, G, I, fs, gs);
  const long long ovol = (long long)(G.nzs - 2*G.H)*G.dplane;         // dense output
  const long long oo = (long long)(p - G.H)*G.dplane + (long long)y*G.nx + x;
  const double r = d_density(fs), ph = d_density(gs);
  if (WHAT == 0) {
    double mf[Q], mg[Q];
    d_moments(fs, mf); d_moments(gs, mg);
    out[0*ovol + oo] = r;
    out[1*ovol + oo] = ph;
    const bool okf = fabs(mf[0]) > (double)FLT_EPSILON, okg = fabs(mg[0]) > (double)FLT_EPSILON;
#pragma unroll
    for (int k = 1; k <= 3; ++k) {
      out[(long long)(k+1)*ovol + oo] = okf ? mf[k]/mf[0] : 0.;
      out[(long long)(k+5)*ovol + oo] = okg ? mg[k]/mg[0] : 0.;
    }
    out[5*ovol + oo] = mf[0] + mg[0];
    return;
  }
  double fn[Q], gn[Q];
  if (inject) {
#pragma unroll
    for (int a = 0; a < Q; ++a) { fn[a] = injf[a*ovol + oo]; gn[a] = injg[a*ovol + oo]; }
  } else if (P.noise_on) {
    double ar, ap, at;
    noise_state(Rf, G, x, y, p, r, ph, ar, ap, at);
    d_noise(P, ar, ap, at, global_site(G, x, y, p), noise_index, ntab, fn, gn);
  } else {
#pragma unroll
    for (int a = 0; a < Q; ++a) { fn[a] = 0.; gn[a] = 0.; }
  }
  if (WHAT == 1) {
#pragma unroll
    for (int a = 0; a < Q; ++a) { out[(long long)a*ovol + oo] = fn[a]; out[(long long)(a+Q)*ovol + oo] = gn[a]; }
    return;
  }
  double nb[Q], grad_rho[3], grad_phi[3];
  gather_field(rho, I, nb); d_gradient(P, nb, grad_rho);
  gather_field(phi, I, nb); d_gradient(P, nb, grad_phi);
  SiteHydro Hy;
  SiteRecip R;
  d_site_recips(P, r, ph, R);
  d_hydrovars(P, fs, gs, r, ph, grad_rho, grad_phi, fn, gn, Hy, R);
  double h[BFLBM_NHYDRO_];
  h[0] = r; h[1] = ph; h[5] = r + ph;
  const double rho_tot = r + ph;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    h[2+k] = Hy.uf[k]; h[6+k] = Hy.ug[k]; h[9+k] = Hy.af[k]; h[12+k] = Hy.ag[k];
    h[15+k] = d_div(r*Hy.ufbar[k] + ph*Hy.ugbar[k] + 0.5*(r*Hy.af[k] + ph*Hy.ag[k]), rho_tot, R.tot);
  }
  h[18] = Hy.nfvel[0]; h[19] = Hy.ngvel[0]; h[20] = Hy.ufbar[0]; h[21] = Hy.ugbar[0];
#pragma unroll
  for (int k = 0; k < BFLBM_NHYDRO_; ++k) if (k < ncomp) out[(long long)k*ovol + oo] = h[k];
}

// ---- halo pack / unpack: 38 (component, plane) entries, each one contiguous nx*ny plane
struct HaloTable { int comp[38]; int plane[38]; };

__global__ void __launch_bounds__(256) k_halo_pack(const double* __restrict__ S, double* __restrict__ buf, Geo G, HaloTable T) {
  const long long s = (long long)blockIdx.x*blockDim.x + threadIdx.x;
  if (s >= G.plane) return;
  const int e = blockIdx.y;
  buf[(long long)e*G.plane + s] = S[(long long)T.comp[e]*G.vol + (long long)T.plane[e]*G.plane + s];
}
__global__ void __launch_bounds__(256) k_halo_unpack(double* __restrict__ S, const double* __restrict__ buf, Geo G, HaloTable T) {
  const long long s = (long long)blockIdx.x*blockDim.x + threadIdx.x;
  if (s >= G.plane) return;
  const int e = blockIdx.y;
  S[(long long)T.comp[e]*G.vol + (long long)T.plane[e]*G.plane + s] = buf[(long long)e*G.plane + s];
}

// one face without staging: entry e of the halo table is a contiguous plane in the source slab (table Ts, possibly the
// memory of a peer GPU) and in the destination slab (table Td)
__global__ void __launch_bounds__(256) k_halo_pull(double* __restrict__ D, const double* __restrict__ Ssrc, long long plane,
                                                   long long dvol, long long svol, HaloTable Td, HaloTable Ts) {
  const long long s = ((long long)blockIdx.x*blockDim.x + threadIdx.x) * 2;      // 16 bytes per lane
  if (s >= plane) return;
  const int e = blockIdx.y;
  const double2 v = *reinterpret_cast<const double2*>(Ssrc + (long long)Ts.comp[e]*svol + (long long)Ts.plane[e]*plane + s);
  *reinterpret_cast<double2*>(D + (long long)Td.comp[e]*dvol + (long long)Td.plane[e]*plane + s) = v;
}

// ---- slab reductions (deterministic: per-block partials, summed on the host in block order)
// out[b][0..5] = sum rho, sum phi, sum rho*i, sum rho*j, sum rho*k (global k), 0
__global__ void __launch_bounds__(256) k_reduce(const double* __restrict__ rho, const double* __restrict__ phi,
                                                double* __restrict__ partial, Geo G, int p0) {
  __shared__ double sh[5][256];
  const long long s_ = (long long)blockIdx.x*blockDim.x + threadIdx.x;
  const int p = p0 + (int)blockIdx.y;
  double v[5] = {0., 0., 0., 0., 0.};
  const int y = (int)(s_ / G.pitch);
  const int x = (int)(s_ - (long long)y*G.pitch);
  if (s_ < G.plane && x < G.nx) {
    const long long o = (long long)p*G.plane + s_;
    const double r = rho[o];
    int gz = G.z0 + (p - G.H);
    v[0] = r; v[1] = phi[o]; v[2] = r*x; v[3] = r*y; v[4] = r*gz;
  }
  for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) for (int k = 0; k < 5; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long long b = (long long)blockIdx.y*gridDim.x + blockIdx.x;
    for (int k = 0; k < 5; ++k) partial[b*5 + k] = sh[k][0];
  }
}

// largest |rho + phi| per block over the slab's own planes (what `auto` keys its stability bound on after an upload,
// bflbm_commit_upload); NaN propagates to the result so that a broken upload is not mistaken for a small one
__global__ void __launch_bounds__(256) k_total_absmax(const double* __restrict__ rho, const double* __restrict__ phi,
                                                      double* __restrict__ partial, Geo G, int p0) {
  __shared__ double sh[256];
  const long long s_ = (long long)blockIdx.x*blockDim.x + threadIdx.x;
  const int p = p0 + (int)blockIdx.y;
  double v = 0.;
  const int y = (int)(s_ / G.pitch);
  const int x = (int)(s_ - (long long)y*G.pitch);
  if (s_ < G.plane && x < G.nx) {
    const long long o = (long long)p*G.plane + s_;
    v = fabs(rho[o] + phi[o]);
  }
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { const double a = sh[threadIdx.x], b = sh[threadIdx.x + w]; sh[threadIdx.x] = (a != a || b != b) ? (a + b) : (a > b ? a : b); }
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(long long)blockIdx.y*gridDim.x + blockIdx.x] = sh[0];
}

#endif  // BFLBM_KERNELS_H_
