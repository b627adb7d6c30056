// bflbm_fused.h -- fused plane-marching collide-and-stream kernel (schedule 1).
//
// One HBM pass per LBM_timestep: read 38 populations, write 38 populations per site.
//
// Each workgroup owns a TX x TY tile in (x,y) and marches through a chunk of planes along z.
// At march position q every thread
//   1. pulls the 38 populations of its site in plane q (f_i(x) = S_i(x - c_i), coalesced along x)
//      and keeps them in registers;
//   2. sums them in the reference's order -> rho,phi of plane q (LBM_binary.H:320-330) into an
//      LDS ring slot; the lanes of the tile additionally pull-and-sum the one-site ring around
//      the tile (those loads are L2 hits: the neighbouring tile's workgroup streams the same
//      lines), so the slot covers (TX+2) x (TY+2) sites;
//   3. after one barrier, collides plane q-1 from the registers kept at the previous position,
//      with the 18-neighbour gradient stencil (LBM_binary.H:134-150) served from the LDS slots
//      of planes q-2, q-1, q, and writes the post-collision populations of plane q-1.
// The slab is split into chunks of planes so that the grid has >> 256 workgroups; a chunk of
// L planes pulls L+2 planes (the two extra only for their densities).
//
// The arithmetic is the same device functions as the two-pass schedule (bflbm_site.h), so both
// schedules and the CPU oracle agree bit for bit.
#ifndef BFLBM_FUSED_H_
#define BFLBM_FUSED_H_

#include "bflbm_kernels.h"

struct FusedGrid {
  int ntx, nty;        // tiles in x and y
  int pa, pb;          // storage planes [pa,pb) to advance
  int lz;              // planes per chunk
  int cstride;         // planes between the starts of consecutive chunks (== lz unless the chunks are disjoint)
  int nchunks;
  int ncols;           // ntx*nty
  int total;           // ncols*nchunks workgroups
  int per_xcd;         // ceil(total/8)
  int sx;              // strip width (tiles in x) of the column order, see fused_map
  int row0 = 0;        // the column order starts at this tile row (hand-over kernel on a lattice with a lower last tile row)
};

// Workgroup -> (column, chunk).  Workgroups b and b+8 share an XCD (round-robin dispatch), so give
// each XCD a contiguous band of tile columns: halo lines are then shared through that XCD's L2.
// Placement only affects speed.
__device__ __forceinline__ bool fused_map(const FusedGrid& F, int b, int& col, int& chunk) {
  const int xcd = b & 7, j = b >> 3;
  const int w = xcd * F.per_xcd + j;             // position in the XCD-major work list
  if (j >= F.per_xcd || w >= F.total) return false;
  // inside the list: chunk-major over groups of columns so that concurrently resident workgroups
  // of one XCD are neighbouring columns of the same chunk
  chunk = w / F.ncols;
  const int c = w % F.ncols;
  // column order inside the list: strips of F.sx tiles in x, tile rows (y) fastest inside a strip, so that the
  // workgroups resident together on an XCD cover a compact block whose inner ring rows are shared through L2
  // (sx == ntx is row-major order: a band of whole tile rows)
  if (F.sx >= F.ntx) { col = c; }
  else {
    const int per_strip = F.sx * F.nty;
    const int strip = c / per_strip, r = c - strip * per_strip;
    const int w_strip = min(F.sx, F.ntx - strip * F.sx);       // last strip may be narrower
    const int tiy = r / w_strip, tix = strip * F.sx + (r - tiy * w_strip);
    col = tiy * F.ntx + tix;
  }
  return true;
}


// Tile geometry: TX x TY threads, one site each, tiles aligned to TX (row segments start on
// 128-byte lines when TX is a multiple of 16).  rho,phi of the one-site ring around the tile are
// pulled and summed as 2*ring half-tasks (site x fluid) spread evenly over the waves.
template <int TX, int TY, int MODE>   // MODE 0: no noise, 1: generated noise, 2: injected noise
__global__ void __launch_bounds__(TX*TY, 2)
k_fused(const double* __restrict__ S, double* __restrict__ D,
        const double* __restrict__ injf, const double* __restrict__ injg,
        Geo G, DevParams P, FusedGrid F, uint32_t noise_index) {
  static_assert((TX * TY) % 64 == 0, "whole waves");
  constexpr int LW = TX + 2;                     // LDS row length
  constexpr int LSZ = (TX + 2) * (TY + 2);
  constexpr int NW = TX * TY / 64;
  static_assert(NW % 2 == 0 && (2 * (TX + 2) + 2 * TY) <= 64 * (NW / 2), "ring tasks of one fluid must fit one per lane of half the waves");
  __shared__ double rp[4][2][LSZ];               // ring of 4 planes x {rho,phi} x (TY+2)x(TX+2)
  __shared__ double gl[Q][TX * TY];              // g populations of the previous plane
  __shared__ double ntab[MODE == 1 ? BFLBM_NORMAL_TABLE_N : 4];
  __shared__ double n3l[3][MODE == 1 ? TX * TY : 1];   // MODE 1: momentum-mode noise of the site being collided (thread-private column)
  if (MODE == 1) d_load_normal_table(ntab, true);

  int col, chunk;
  if (!fused_map(F, (int)blockIdx.x, col, chunk)) return;   // whole workgroup leaves together
  const int tix = col % F.ntx, tiy = col / F.ntx;
  const int x0 = tix * TX, y0 = tiy * TY;
  const int aw = min(TX, G.nx - x0), ah = min(TY, G.ny - y0);   // active extent of this tile
  const int tid = threadIdx.x;
  const int tx = tid % TX, ty = tid / TX;
  auto wrapx = [&](int v) { return v < 0 ? v + G.nx : (v >= G.nx ? v - G.nx : v); };
  auto wrapy = [&](int v) { return v < 0 ? v + G.ny : (v >= G.ny ? v - G.ny : v); };

  // ---- own site: per-thread 32-bit element offsets inside a plane, fixed for the whole march
  const bool loader = (tx < aw) && (ty < ah);
  const bool interior = loader;
  const int x = loader ? x0 + tx : x0, y = loader ? y0 + ty : y0;
  // byte offsets: a plane is < 4 GB (checked at creation), so  address = wave-uniform base + 32-bit lane
  // offset  and the loads/stores use the scalar-base addressing form (no 64-bit per-lane address math)
  const unsigned xo[3] = { (unsigned)wrapx(x - 1) * 8u, (unsigned)x * 8u, (unsigned)wrapx(x + 1) * 8u };
  const unsigned yo[3] = { (unsigned)(wrapy(y - 1) * G.pitch) * 8u, (unsigned)(y * G.pitch) * 8u, (unsigned)(wrapy(y + 1) * G.pitch) * 8u };
  auto ld = [](const double* __restrict__ base, unsigned boff) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + boff); };
#ifndef BFLBM_FUSED_NT_STORES
#define BFLBM_FUSED_NT_STORES 0     // non-temporal population stores: +0.8 % at 256^3, -2.0 % at 512^3 in this kernel (profiles/r04_nt_hints.txt): off
#endif
  auto st = [](double* __restrict__ base, unsigned boff, double v) {
    double* q = reinterpret_cast<double*>(reinterpret_cast<char*>(base) + boff);
    if (BFLBM_FUSED_NT_STORES) __builtin_nontemporal_store(v, q); else *q = v;
  };
  // ---- ring half-task of this thread: lanes 0..nper-1 of every wave; the lower half of the waves sums
  // fluid f, the upper half fluid g, so the fluid (and with it the load base) is wave-uniform
  const int nring = 2 * (aw + 2) + 2 * ah;
  const int nper = (nring + NW / 2 - 1) / (NW / 2);
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hfl = wv / (NW / 2);
  const int task = (wv % (NW / 2)) * nper + lane;
  const bool has_task = lane < nper && task < nring;
  int hlx = 0, hly = 0;                          // LDS coordinates of the ring site
  if (has_task) {
    const int r = task;
    if (r < aw + 2) { hlx = r; hly = 0; }
    else if (r < 2 * (aw + 2)) { hlx = r - (aw + 2); hly = ah + 1; }
    else if (r < 2 * (aw + 2) + ah) { hlx = 0; hly = r - 2 * (aw + 2) + 1; }
    else { hlx = aw + 1; hly = r - 2 * (aw + 2) - ah + 1; }
  }
  const int hx = wrapx(x0 + hlx - 1);            // in [-1, nx]: one wrap suffices
  const int hy = wrapy(y0 + hly - 1);
  const unsigned hxo[3] = { (unsigned)wrapx(hx - 1) * 8u, (unsigned)hx * 8u, (unsigned)wrapx(hx + 1) * 8u };
  const unsigned hyo[3] = { (unsigned)(wrapy(hy - 1) * G.pitch) * 8u, (unsigned)(hy * G.pitch) * 8u, (unsigned)(wrapy(hy + 1) * G.pitch) * 8u };

  const int lown = (ty + 1) * LW + (tx + 1);
  const int lhalo = hly * LW + hlx;

  // ---- chunk of planes
  const int qa = F.pa + chunk * F.cstride;
  const int qb = min(F.pb, qa + F.lz);
  auto wrapp = [&](int q) {                      // q in [-2, nzs+1]; nzs may be 1, so use a true modulo
    if (!G.zwrap) return q;
    const int m = q % G.nzs;
    return m < 0 ? m + G.nzs : m;
  };

  // Held across one march position: f of the previous plane in registers, g of the previous
  // plane in LDS (each thread only touches its own column gl[.][tid], so no barrier is needed).
  double pf[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) pf[i] = 0.;

  int it = 0;
  for (int q = qa - 1; q <= qb; ++q, ++it) {
    const int slot = it & 3;
    // wave-uniform plane bases: every load below is  (SGPR base) + (32-bit lane offset)
    const double* __restrict__ pl[3] = { S + (long long)wrapp(q - 1) * G.plane, S + (long long)wrapp(q) * G.plane,
                                         S + (long long)wrapp(q + 1) * G.plane };
    // 1. pull plane q: the own site first, then the ring half-task (measured: +2.3 % over ring first);
    //    everything is in flight together
    double cf[Q], cg[Q];
    if (loader) {
      unsigned oo[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) { oo[a][b2] = yo[a] + xo[b2]; asm volatile("" : "+v"(oo[a][b2])); }
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const double* __restrict__ b = pl[1 - Vel::cz[i]] + (long long)i * G.vol;
        const unsigned o = oo[1 - Vel::cy[i]][1 + BFLBM_PX(Vel::cx[i])];
        cf[i] = ld(b, o);
        cg[i] = ld(b + (long long)Q * G.vol, o);
      }
    } else {
#pragma unroll
      for (int i = 0; i < Q; ++i) { cf[i] = 0.; cg[i] = 0.; }
    }
    double hv[Q];
    if (has_task) {
      // the nine (dy,dx) offsets as opaque 32-bit values INSIDE this block: instruction selection then sees
      // base + zext(offset) and uses the scalar-base addressing form instead of 64-bit per-lane adds
      unsigned ho[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 3; ++b2) { ho[a][b2] = hyo[a] + hxo[b2]; asm volatile("" : "+v"(ho[a][b2])); }
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const double* __restrict__ b = pl[1 - Vel::cz[i]] + (long long)hfl * Q * G.vol + (long long)i * G.vol;
        hv[i] = ld(b, ho[1 - Vel::cy[i]][1 + BFLBM_PX(Vel::cx[i])]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < Q; ++i) hv[i] = 0.;
    }
    // 2. densities of plane q into the ring slot.  The sums start from an opaque zero defined HERE: with a
    // literal 0.0 the compiler sinks the first addition (0.0 + f_0) into the load blocks above and waits
    // for the first load before it issues the rest.
    double zero = 0.0;
    asm volatile("" : "+v"(zero));
    auto density = [&](const double (&fs)[Q]) { double r = zero;
#pragma unroll
      for (int i = 0; i < Q; ++i) r += fs[i];
      return r; };
    // own sums first: their loads were issued first, the ring loads are still landing meanwhile
    if (loader) { rp[slot][0][lown] = density(cf); rp[slot][1][lown] = density(cg); }
    if (has_task) rp[slot][hfl][lhalo] = density(hv);
    __syncthreads();
    // 3. collide plane q-1: f from registers, g streamed out of LDS while plane q's g takes its place
    const bool do_collide = (q - 1 >= qa) && (q - 1 < qb) && interior;
    double mg[Q], jg[3];
    if (do_collide) {
      double pg[Q];
#pragma unroll
      for (int i = 0; i < Q; ++i) pg[i] = gl[i][tid];
      d_moments(pg, mg);
      d_momentum(pg, jg);
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) gl[i][tid] = cg[i];
    if (do_collide) {
      double mf[Q], jf[3];
      d_moments(pf, mf);
      d_momentum(pf, jf);
      const int sl[3] = { (it - 2) & 3, (it - 1) & 3, it & 3 };
      const double r = rp[sl[1]][0][lown], ph = rp[sl[1]][1][lown];
      double nb[Q], grad_rho[3], grad_phi[3];
#pragma unroll
      for (int i = 0; i < Q; ++i) nb[i] = rp[sl[1 + Vel::cz[i]]][0][lown + Vel::cy[i] * LW + Vel::cx[i]];
      d_gradient(P, nb, grad_rho);
#pragma unroll
      for (int i = 0; i < Q; ++i) nb[i] = rp[sl[1 + Vel::cz[i]]][1][lown + Vel::cy[i] * LW + Vel::cx[i]];
      d_gradient(P, nb, grad_phi);
      const int pc = wrapp(q - 1);
      // noise: momentum modes now (hydrovars needs them), the rest right before each relaxation
      double fn3[3] = {0., 0., 0.}, gn3[3] = {0., 0., 0.};
      NoiseAmp NA; bflbm_rng_state rst;
      const double* __restrict__ nb_f = nullptr; const double* __restrict__ nb_g = nullptr;
      long long nvol = 0; unsigned no = 0;
      if (MODE == 2) {
        nvol = (long long)(G.nzs - 2 * G.H) * G.dplane;          // injected arrays are dense
        nb_f = injf + (long long)(pc - G.H) * G.dplane;
        nb_g = injg + (long long)(pc - G.H) * G.dplane;
        no = (unsigned)(y * G.nx + x) * 8u;
#pragma unroll
        for (int k = 0; k < 3; ++k) { fn3[k] = ld(nb_f + (1 + k) * nvol, no); gn3[k] = ld(nb_g + (1 + k) * nvol, no); }
      } else if (MODE == 1) {
        d_noise_amp(P, r, ph, r + ph, NA);
        d_noise_head(P, NA, global_site(G, x, y, pc), noise_index, ntab, rst, fn3);
#pragma unroll
        for (int k = 0; k < 3; ++k) { gn3[k] = -fn3[k]; n3l[k][tid] = fn3[k]; }
      }
      double* __restrict__ Dp = D + (long long)pc * G.plane;
      unsigned o = yo[1] + xo[1];
      asm volatile("" : "+v"(o));
      unsigned os3[3] = { yo[1] + xo[0], o, yo[1] + xo[2] };       // store slots of populations with c_x = -1, 0, +1 (BFLBM_XSHIFT)
      if (BFLBM_XSHIFT) { asm volatile("" : "+v"(os3[0])); asm volatile("" : "+v"(os3[2])); }
      {
        SiteHydro Hy;
        SiteRecip R;
        d_site_recips(P, r, ph, R);
        d_hydrovars_j(P, jf, jg, r, ph, grad_rho, grad_phi, fn3, gn3, Hy, R);
        double v_b[3];
        d_barycentric(r, ph, Hy, v_b, R);
        {
          if (MODE == 2) {
            double fn[Q];
#pragma unroll
            for (int a = 0; a < Q; ++a) fn[a] = ld(nb_f + a * nvol, no);
            d_relax<true>(P, mf, r, v_b, Hy.uf, Hy.af, P.inv_tau_f_bar, fn, R.cs4);
          } else if (MODE == 1) {
            const double n3[3] = { n3l[0][tid], n3l[1][tid], n3l[2][tid] };     // reloaded: keeps the register budget
            d_relax_generated(P, mf, r, v_b, Hy.uf, Hy.af, P.inv_tau_f_bar, n3, sqrt(fabs(r)), ntab, rst, R.cs4);
          } else {
            const double zn[Q] = {0.};
            d_relax<false>(P, mf, r, v_b, Hy.uf, Hy.af, P.inv_tau_f_bar, zn, R.cs4);
          }
          double out[Q];
          d_populations(mf, out);
#pragma unroll
          for (int i = 0; i < Q; ++i) st(Dp + (long long)i * G.vol, os3[1 + BFLBM_SX(Vel::cx[i])], out[i]);
        }
        {
          if (MODE == 2) {
            double gn[Q];
#pragma unroll
            for (int a = 0; a < Q; ++a) gn[a] = ld(nb_g + a * nvol, no);
            d_relax<true>(P, mg, ph, v_b, Hy.ug, Hy.ag, P.inv_tau_g_bar, gn, R.cs4);
          } else if (MODE == 1) {
            const double n3[3] = { -n3l[0][tid], -n3l[1][tid], -n3l[2][tid] };
            d_relax_generated(P, mg, ph, v_b, Hy.ug, Hy.ag, P.inv_tau_g_bar, n3, sqrt(fabs(ph)), ntab, rst, R.cs4);
          } else {
            const double zn[Q] = {0.};
            d_relax<false>(P, mg, ph, v_b, Hy.ug, Hy.ag, P.inv_tau_g_bar, zn, R.cs4);
          }
          double out[Q];
          d_populations(mg, out);
#pragma unroll
          for (int i = 0; i < Q; ++i) st(Dp + (long long)(i + Q) * G.vol, os3[1 + BFLBM_SX(Vel::cx[i])], out[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) pf[i] = cf[i];
  }
}

#ifndef BFLBM_FUSED_TX
#define BFLBM_FUSED_TX 64
#endif
#ifndef BFLBM_FUSED_TY
#define BFLBM_FUSED_TY 8
#endif

static int g_fused_ncu = 0;     // compute units of the device (set at context creation)
// Tile shape, chunking and workgroup order of one launch of the fused kernel over the storage planes [pa, pb); returns the
// tile width (the height is 512 / width)
static inline int fused_plan(const Geo& G, int pa, int pb, int mode, int pair_len, FusedGrid& F) {
  // pair_len > 0: ONE launch over the two disjoint plane ranges [pa, pa+pair_len) and [pb-pair_len, pb)
  // (the boundary plane pairs of a slab), one chunk each.
  // Tile shape: 64 x 8 sites; lattices narrower than 64 in x get the same 512 sites as 32 x 16, 16 x 32 or 8 x 64
  // (zero noise only; e.g. the reference's 8 x 256 x 64 flat-interface box would use 8 of 64 lanes of a 64-wide tile)
  const int TX = (mode != 0 || G.nx > 32) ? BFLBM_FUSED_TX : (G.nx > 16 ? 32 : (G.nx > 8 ? 16 : 8));
  const int TY = (BFLBM_FUSED_TX * BFLBM_FUSED_TY) / TX;
  F.ntx = (G.nx + TX - 1) / TX;
  F.nty = (G.ny + TY - 1) / TY;
  F.ncols = F.ntx * F.nty;
  F.pa = pa; F.pb = pb;
  const int np = pb - pa;
  static const int want_env = [] { const char* e = getenv("BFLBM_FUSED_WG"); return e ? atoi(e) : 0; }();
  // Chunking.  One workgroup is resident per CU, so the launch runs in rounds of `ncu` workgroups and costs
  // about  rounds x (planes per chunk + 1)  plane marches (a chunk of L planes marches L+2, the two extra
  // ones pull only).  Pick the chunk count that minimises this: 256^3 -> 128 columns x 2 chunks = one full
  // round; 192^3 -> 72 columns x 7 chunks = 504 workgroups in 2 rounds of 30 planes instead of 288 in
  // "1.1" rounds of 50.  Marches longer than 256 planes are avoided on a single slab (neighbouring
  // workgroups drift apart and lose L2 sharing; measured on MI355X); a slab of a multi-GPU run is cut
  // into >= 3 rounds of shorter workgroups so that the RCCL copy kernels of the overlapped exchange, which
  // need a few CUs of their own, delay at most a short tail of the interior sweep.
  // BFLBM_FUSED_WG overrides the target workgroup count (tuning only).
  const int ncu = g_fused_ncu > 0 ? g_fused_ncu : 256;
  // BFLBM_SLAB_ROUNDS: minimum number of rounds of the interior sweep of a slab (default 3; tuning knob for
  // real multi-GPU runs: fewer rounds = less look-ahead overhead, more = shorter tail behind the RCCL kernels)
  static const int min_slab_rounds = [] { const char* e = getenv("BFLBM_SLAB_ROUNDS"); return e && atoi(e) > 0 ? atoi(e) : 3; }();
  const int maxchunks = std::max(1, np / 2);                     // small lattices: short chunks buy parallelism
  int nchunks;
  if (want_env > 0) {
    nchunks = std::min(maxchunks, std::max(1, (want_env + F.ncols - 1) / F.ncols));
  } else {
    long long best = -1; nchunks = 1;
    for (int k = 1; k <= maxchunks; ++k) {
      const int lz = (np + k - 1) / k, chunks = (np + lz - 1) / lz;
      if (chunks != k) continue;                                  // same partition as a smaller k
      if (G.zwrap && lz > 256 && k < maxchunks) continue;
      const long long total = (long long)F.ncols * chunks, rounds = (total + ncu - 1) / ncu;
      if (!G.zwrap && rounds < min_slab_rounds && k < maxchunks) continue;
      const long long cost = rounds * (lz + 1);
      if (best < 0 || cost < best) { best = cost; nchunks = k; }
    }
  }
  F.lz = (np + nchunks - 1) / nchunks;
  F.nchunks = (np + F.lz - 1) / F.lz;
  F.cstride = F.lz;
  if (pair_len > 0 && np > 2 * pair_len) { F.lz = pair_len; F.nchunks = 2; F.cstride = np - pair_len; }
  F.total = F.ncols * F.nchunks;
  F.per_xcd = (F.total + 7) / 8;
  { static const int sx_env = [] { const char* e = getenv("BFLBM_MAP_SX"); return e ? atoi(e) : 0; }(); F.sx = sx_env > 0 ? sx_env : std::min(F.ntx, 4); }   // strips of 4 tiles: +2 % at 512^3 (ntx = 8), identical at 256^3
  return TX;
}

// returns the launch's error code
static inline hipError_t fused_launch(const double* S, double* D, const double* injf, const double* injg,
                               const Geo& G, const DevParams& P, int pa, int pb,
                               uint32_t noise_index, int mode, hipStream_t stream, int pair_len = 0) {
  FusedGrid F;
  const int TX = fused_plan(G, pa, pb, mode, pair_len, F);
  const int TY = (BFLBM_FUSED_TX * BFLBM_FUSED_TY) / TX;
  dim3 grid((unsigned)(F.per_xcd * 8)), block(TX * TY);
  constexpr int TX0 = BFLBM_FUSED_TX, TY0 = BFLBM_FUSED_TY;
  if (mode == 2)      hipLaunchKernelGGL((k_fused<TX0, TY0, 2>), grid, block, 0, stream, S, D, injf, injg, G, P, F, noise_index);
  else if (mode == 1) hipLaunchKernelGGL((k_fused<TX0, TY0, 1>), grid, block, 0, stream, S, D, injf, injg, G, P, F, noise_index);
  else if (TX == 32)  hipLaunchKernelGGL((k_fused<32, (TX0 * TY0) / 32, 0>), grid, block, 0, stream, S, D, injf, injg, G, P, F, noise_index);
  else if (TX == 16)  hipLaunchKernelGGL((k_fused<16, (TX0 * TY0) / 16, 0>), grid, block, 0, stream, S, D, injf, injg, G, P, F, noise_index);
  else if (TX == 8)   hipLaunchKernelGGL((k_fused<8, (TX0 * TY0) / 8, 0>), grid, block, 0, stream, S, D, injf, injg, G, P, F, noise_index);
  else                hipLaunchKernelGGL((k_fused<TX0, TY0, 0>), grid, block, 0, stream, S, D, injf, injg, G, P, F, noise_index);
  return hipGetLastError();
}

#endif  // BFLBM_FUSED_H_
