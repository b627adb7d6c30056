// bflbm_handover.h -- fused plane-marching collide-and-stream kernel with a cross-step hand-over of the
// tile-boundary densities (schedule 3).
//
// The plane march of bflbm_fused.h needs rho,phi of the streamed state on a one-site ring around each
// workgroup's tile (gradient stencil, LBM_binary.H:134-150 applied to the densities of :315-330).  There
// the ring is pulled: 19 loads per ring site and fluid to produce one number, lines that belong to the
// neighbouring tiles (measured: 1.33x-1.6x the algorithmic read bytes, profiles/r01_*).  Here the ring
// densities of step t+1 are handed over from step t instead:
//
//   rho_{t+1}(r) = sum_i f*_i(r - c_i)           (f* = post-collision populations of step t)
//
// While the 19 outputs of a site are in registers the producing workgroup sorts them by destination:
//   x  wave shifts (DPP wave_shr/wave_shl, a tile row is one 64-lane wave)       -> 9 buckets (dy,dz)
//   z  a two-stage register/LDS pipeline along the march                          -> 3 sums (dy)
//   y  finished sums of the row next to an edge row cross through LDS
// and writes per tile, plane and fluid a FRAME of partial sums over the sources INSIDE the tile:
//   E  for its own edge sites      (2*64 + 2*(TY-2) values)   -- what the neighbours are missing
//   O  for the ring sites outside  (2*64 + 2*(TY+2) values)   -- what it contributes to the neighbours
// Step t+1 forms the ring density of a site as  E(owner tile) + O(own tile) [+ O of the two other tiles
// at the 12 sites next to a tile corner]: 2-4 loads instead of 19.  Frame traffic: 2*FR/(64*TY) values per
// site and fluid each way (64x4 tiles: +5.7 % of 608 B/site).
//
// The tile's OWN densities are still summed from the pulled populations in the reference's order.  The ring
// densities are sums of the same 19 numbers in a different, fixed order: deterministic, equal to the
// reference's to an ulp, not bit-identical; they only enter the gradient of the tile's edge sites.  The
// schedule is therefore held to the north-star tolerance (rho,phi rel 1e-12, u abs 1e-12 cs) and the two
// bit-exact schedules (0 two-pass, 1 fused with pulled ring) remain the cross-check.
//
// Frames of the first and last plane of a chunk would need sources of the neighbouring chunk: those planes
// (and every plane of the first step after an init/upload) use the pulled ring.
//
// Lattices that are not whole tiles (template flag RAG; the full-tile kernel is compiled without any of it):
//   x  the last tile of a row may be narrower (aw < 64 sites): its idle lanes load a duplicate of the last site, store
//      nothing and contribute zeros to the x shifts; its right column lane is aw-1 and its right ring column aw+1, and
//      the frames keep their layout (slots are roles, not coordinates), so a narrow tile hands over like any other.
//   y  the last tile row may be lower (ah < TY rows).  In a full tile every row has ONE y role (bottom edge, hands its -y
//      sum down, hands its +y sum up, top edge); in a tile of 2 or 3 rows a row has two (ah = 2: row 0 is the bottom
//      edge AND hands its +y sum to row 1), so the ragged kernel runs both travelling sums in every row and picks by
//      role; the frame keeps its layout with the top row / top ring row at ah-1 / ah.  (ah = 1, like aw = 1, is
//      refused by the host: a ring site owned by such a tile also collects from the tile beyond it.)
// A ring site is looked up by its coordinates RELATIVE to each of the 3 x 3 surrounding tiles (not by wrapped global
// coordinates), so a lattice one tile wide or high -- where the neighbour on both sides is the tile itself and a ring
// site is one of its own edge sites at the same time -- needs nothing special: nx >= 64 (one tile: nx = 64), ny >= TY.
#ifndef BFLBM_HANDOVER_H_
#define BFLBM_HANDOVER_H_

#include "bflbm_fused.h"

// Requests of the next plane's f half (quiet kernel): one every BFLBM_HO_SPREAD_F VALU instructions of the relaxation of
// fluid f instead of one burst of 19 before it; 0 = burst (see the comment at the call site)
#ifndef BFLBM_HO_SPREAD_F
#define BFLBM_HO_SPREAD_F 28
#endif
#ifndef BFLBM_HO_NT_STORES
#define BFLBM_HO_NT_STORES 1
#endif
#ifndef BFLBM_HO_NT_LOADS
#define BFLBM_HO_NT_LOADS 0
#endif
#ifndef BFLBM_HO_NT_FRAMES
#define BFLBM_HO_NT_FRAMES 0     // experiment: the frame stores too
#endif
#ifndef BFLBM_HO_SPREAD_F1
#define BFLBM_HO_SPREAD_F1 44     // noise kernel, round 4: with the 11-instruction normals one request every 44 VALU instructions is
#endif                            // +2.3 % at 512^3 and +2.0 % at 256^3 (16 ... 32: -1 %, 56: 0, 68: +2.5 / +0.5 %); round 3's generator: every spacing lost


// Diagnostic build (-DBFLBM_STAMP, tools/ho_stamps.py): shader-clock stamps at the phase boundaries of a march position,
// written by lane 0 of every wave of ONE workgroup for 64 steady-state positions.  Not compiled into the product.
#ifdef BFLBM_STAMP
#define HO_NSTAMP 10
#define HO_STAMP_POS 64
__device__ unsigned long long g_ho_stamps[4 * HO_STAMP_POS * HO_NSTAMP];
#define HO_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); ts[k] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define HO_STAMP(k) do { } while (0)
#endif

template <int TY> struct HoLayout {
  static constexpr int TX = 64;
  // slots of one fluid's frame (doubles): the four 64-entry rows first, each on its own 128-byte lines (they are stored
  // and loaded by whole waves), the column entries (4*TY doubles, one line for TY = 4) behind them
  static constexpr int EB = 0, ET = TX, OB = 2 * TX, OT = 3 * TX;
  static constexpr int EL = 4 * TX, ER = EL + (TY - 2), OL = ER + (TY - 2), OR_ = OL + (TY + 2);
  static constexpr int FR = OR_ + (TY + 2);            // 4*TX + 4*TY: 272 (TY=4), 288 (TY=8) -- whole 128-byte lines
  static constexpr int REC = 2 * FR;                   // both fluids
};

struct HoGrid {
  const double* fin;     // frames of the state being read   [plane][tile][fluid][FR]
  double* fout;          // frames of the state being written
  long long fplane;      // doubles per plane = ntiles * REC
  int use_frames;        // fin holds frames written by the same launch geometry one step earlier
};

// destination slot of lattice site (lx,ly), given relative to a tile's origin, in that tile's frame; -1: none
template <int TYL>
__device__ __forceinline__ int ho_frame_slot(int lx, int ly, int TX /* width */, int TY /* height of the tile that owns the frame */) {
  using L = HoLayout<TYL>;
  if (lx >= 0 && lx < TX && ly >= 0 && ly < TY) {
    if (ly == 0) return L::EB + lx;
    if (ly == TY - 1) return L::ET + lx;
    if (lx == 0) return L::EL + ly - 1;
    if (lx == TX - 1) return L::ER + ly - 1;
    return -1;
  }
  if (lx >= 0 && lx < TX) {
    if (ly == -1) return L::OB + lx;
    if (ly == TY) return L::OT + lx;
    return -1;
  }
  if (ly >= -1 && ly <= TY) {
    if (lx == -1) return L::OL + ly + 1;
    if (lx == TX) return L::OR_ + ly + 1;
  }
  return -1;
}
__device__ __forceinline__ double ho_shr(double v) {   // value of lane-1 (travelling +x), 0 in lane 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ho_shl(double v) {   // value of lane+1 (travelling -x), 0 in lane 63
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// ONE workgroup per CU at one wave per SIMD (512 registers per lane, AGPRs included, 160 KB of LDS): the pending
// plane of BOTH fluids waits in LDS, and the loads of plane q+1 are issued before plane q-1 is collided -- the f half
// before fluid f is relaxed, the g half after it is finished -- so that memory latency runs under the arithmetic
// inside one wave instead of between waves.  (At two waves per SIMD the frame producer spills 59-90 registers; that
// form, and the variants that kept the f plane in registers or requested the next plane before the density sums,
// were measured and removed: NOTES.md section 3.1b.)
// MODE 0: zero noise; MODE 1: generated thermal noise (csrc/bflbm_rng.h), drawn where it is added.
// RAG: the lattice has a narrower last tile column and/or a lower last tile row (see the header comment)
template <int TY, int MODE, bool RAG>
__global__ void __launch_bounds__(64 * TY, 1)
k_fused_ho(const double* __restrict__ S, double* __restrict__ D, Geo G, DevParams P, FusedGrid F, HoGrid Hg, uint32_t noise_index) {
  using L = HoLayout<TY>;
  constexpr int TX = 64, NT = TX * TY;
  constexpr int LW = TX + 2, LSZ = (TX + 2) * (TY + 2);
  constexpr int NRING = 2 * (TX + 2) + 2 * TY;
  static_assert(TY >= 4 && 4 + 2 * TY <= 64 && 8 * TY <= 64, "tile shape");
  __shared__ double rp[4][2][LSZ];                     // ring of 4 planes x {rho,phi} x (TY+2)x(TX+2)
  __shared__ double gl[Q][NT];                         // g populations of the previous plane
  __shared__ double fl[Q][NT];                         // f populations of the previous plane
  __shared__ double exch[2][2][2][2][TX];              // [buf][fluid][side][0 edge row's own sum, 1 sum handed over by the row next to it][lane]
  __shared__ double accs[2][2][2][TX];                 // [stage][fluid][side][lane] z pipeline of the edge rows' own sums
  __shared__ double colacc[2][2][2][TY][6];            // [stage][fluid][side][row][kind] z pipeline of the column lanes
  __shared__ double colfin[2][2][2][TY][6];            // [buf][fluid][side][row][kind] finished column sums

  __shared__ double ntab[MODE == 1 ? BFLBM_NORMAL_TABLE_N : 4];

  int col, chunk;
  if (!fused_map(F, (int)blockIdx.x, col, chunk)) return;   // whole workgroup leaves together
  if (MODE == 1) d_load_normal_table(ntab, true);
  const int tix = col % F.ntx;
  int tiy = col / F.ntx + F.row0;
  if (tiy >= F.nty) tiy -= F.nty;
  const int x0 = tix * TX, y0 = tiy * TY;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int ty = __builtin_amdgcn_readfirstlane(tid >> 6);   // a wave is a tile row
  const int tx = lane;
  auto wrapx = [&](int v) { return v < 0 ? v + G.nx : (v >= G.nx ? v - G.nx : v); };
  auto wrapy = [&](int v) { return v < 0 ? v + G.ny : (v >= G.ny ? v - G.ny : v); };
  auto ld = [](const double* __restrict__ base, unsigned boff) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + boff); };
  auto ldnt = [](const double* __restrict__ base, unsigned boff) { return __builtin_nontemporal_load(reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + boff)); };
  auto st = [](double* __restrict__ base, unsigned boff, double v) {
    double* q = reinterpret_cast<double*>(reinterpret_cast<char*>(base) + boff);
    if (BFLBM_HO_NT_FRAMES) __builtin_nontemporal_store(v, q); else *q = v;
  };

  // active extent of this tile; idle lanes / rows of a ragged tile work on a duplicate of the last active site
  const int aw = RAG ? min(TX, G.nx - x0) : TX, ah = RAG ? min(TY, G.ny - y0) : TY;
  const bool active_x = !RAG || tx < aw;
  const bool active = !RAG || (tx < aw && ty < ah);
  const int x = x0 + (RAG ? min(tx, aw - 1) : tx), y = y0 + (RAG ? min(ty, ah - 1) : ty);
  // y roles of this row in a ragged tile (wave-uniform; a full tile has exactly one per row, see row_down / row_up below)
  const bool r_bot = RAG && ty == 0, r_top = RAG && ty == ah - 1;           // edge rows: own sums -> E, travelling sum -> O
  const bool r_hdn = RAG && ah >= 2 && ty == 1;                            // its -y travelling sum completes row 0's E
  const bool r_hup = RAG && ah >= 2 && ty == ah - 2;                       // its +y travelling sum completes row ah-1's E
  const unsigned xo[3] = { (unsigned)wrapx(x - 1) * 8u, (unsigned)x * 8u, (unsigned)wrapx(x + 1) * 8u };
  const unsigned yo[3] = { (unsigned)(wrapy(y - 1) * G.pitch) * 8u, (unsigned)(y * G.pitch) * 8u, (unsigned)(wrapy(y + 1) * G.pitch) * 8u };
  const int lown = (ty + 1) * LW + (tx + 1);

  // ---- roles of this thread in the frame production
  const bool row_down = ty < 2, row_up = ty >= TY - 2;       // rows whose -y / +y travelling sums are needed
  const bool row_kind = row_down || row_up;                  // wave-uniform
  const bool edge_row = (ty == 0) || (ty == TY - 1);
  const bool is_edge = (tid < TX) || (tid >= NT - TX);       // the same as a lane predicate (keeps the producer free of uniform branches)
  const int side_y = row_down ? 0 : 1;
  const bool col_lane = (tx == 0) || (tx == aw - 1);         // aw >= 2 (host check)
  const int side_x = (tx == 0) ? 0 : 1;
  const unsigned tile_rec = (unsigned)((tiy * F.ntx + tix) * L::REC) * 8u;     // byte offset of this tile's frame in a plane
  // ---- ring site of this thread when the ring comes from frames, both fluids.  Wave 0 takes the 64 sites below
  // the tile, wave 1 the 64 above it: their two pieces (E of the neighbouring tile, O of the own one) are whole
  // 512-byte frame rows, four full lines per load.  The 4 corners and 2*TY column sites, which have up to four
  // pieces in scattered places, go to lanes of wave 2.  (Spread evenly over the four waves, every wave issued all
  // eight frame loads on parts of those rows: twice the instructions and half again the line requests.)
  const bool has_rtask = (ty < 2 && lane < aw) || (ty == 2 && lane < 4 + 2 * ah);
  int hlx = 0, hly = 0;
  if (ty == 0) { hlx = lane + 1; hly = 0; }
  else if (ty == 1) { hlx = lane + 1; hly = ah + 1; }
  else if (ty == 2 && lane < 4) { hlx = (lane & 1) ? aw + 1 : 0; hly = (lane & 2) ? ah + 1 : 0; }
  else if (ty == 2 && lane < 4 + ah) { hlx = 0; hly = lane - 4 + 1; }
  else if (ty == 2 && lane < 4 + 2 * ah) { hlx = aw + 1; hly = lane - 4 - ah + 1; }
  const int lhalo = hly * LW + hlx;
  // the frames that hold a piece of this ring site: the owner's E and the O of every other tile around it.  The site
  // is (hlx-1, hly-1) in this tile's coordinates, hence (that - offset of the tile) in the coordinates of each of the
  // 3 x 3 tiles around; when two of those are the same tile (one or two tiles per direction) they are different
  // REPRESENTATIONS of the site in that tile's frame, and at most one of them names a slot per piece.
  unsigned fo[4] = {0u, 0u, 0u, 0u};
  int nfo = 0;
  if (has_rtask) {
    const int wlast = G.nx - (F.ntx - 1) * TX, hlast = G.ny - (F.nty - 1) * TY;   // extent of the last tile column / row (TX, TY when full)
    for (int dty = -1; dty <= 1; ++dty) {
      for (int dtx = -1; dtx <= 1; ++dtx) {
        int ux = tix + dtx, uy = tiy + dty;
        ux = ux < 0 ? ux + F.ntx : (ux >= F.ntx ? ux - F.ntx : ux);
        uy = uy < 0 ? uy + F.nty : (uy >= F.nty ? uy - F.nty : uy);
        const int wu = (ux == F.ntx - 1) ? wlast : TX, hu = (uy == F.nty - 1) ? hlast : TY;
        const int lx = (hlx - 1) + (dtx < 0 ? wu : (dtx > 0 ? -aw : 0));
        const int ly = (hly - 1) + (dty < 0 ? hu : (dty > 0 ? -ah : 0));
        const int slot = ho_frame_slot<TY>(lx, ly, wu, hu);
        if (slot >= 0 && nfo < 4) { fo[nfo] = (unsigned)((uy * F.ntx + ux) * L::REC + slot) * 8u; ++nfo; }
      }
    }
  }

  const int qa = F.pa + chunk * F.cstride;
  const int qb = min(F.pb, qa + F.lz);
  auto wrapp = [&](int q) {
    if (!G.zwrap) return q;
    const int m = q % G.nzs;
    return m < 0 ? m + G.nzs : m;
  };
  // planes whose frames are complete: all three source planes collided by this workgroup
  const int fa = qa + 1, fb = qb - 2;                        // [fa, fb]

  double anb[2][2] = {{0., 0.}, {0., 0.}};                   // [fluid][stage] z pipeline of the row's travelling sum
  double aup[2][2] = {{0., 0.}, {0., 0.}};                   // RAG: the +y travelling sum runs beside the -y one (anb)

  // finish the frames of plane tpf from what the previous march position left in LDS (after a barrier)
  auto finish = [&](int tpf, int rb) {
    if (tpf < fa || tpf > fb) return;
    double* __restrict__ fp = Hg.fout + (long long)tpf * Hg.fplane;
    if (!RAG) {
      if (edge_row) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const double e = exch[rb][k][side_y][0][lane] + exch[rb][k][side_y][1][lane];
          st(fp, tile_rec + (unsigned)(k * L::FR + (side_y ? L::ET : L::EB) + lane) * 8u, e);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (r_bot) st(fp, tile_rec + (unsigned)(k * L::FR + L::EB + lane) * 8u, exch[rb][k][0][0][lane] + exch[rb][k][0][1][lane]);
        if (r_top) st(fp, tile_rec + (unsigned)(k * L::FR + L::ET + lane) * 8u, exch[rb][k][1][0][lane] + exch[rb][k][1][1][lane]);
      }
    }
    if (ty == 0 && lane < 8 * TY) {
      const int k = lane / (4 * TY), rem = lane % (4 * TY), sd = rem / (2 * TY), u = rem % (2 * TY);
      const int ne = ah > 2 ? ah - 2 : 0;                      // own edge sites of the column: rows 1..ah-2
      double v = 0.; int slot = -1;
      if (u < ne) {
        const int row = u + 1;
        v = colfin[rb][k][sd][row][0] + colfin[rb][k][sd][row - 1][1] + colfin[rb][k][sd][row + 1][2];
        slot = (sd ? L::ER : L::EL) + row - 1;
      } else if (u - ne < ah + 2) {                            // ring site beside the column, rows -1..ah
        const int row = u - ne - 1;
        if (row >= 0 && row < ah) v = colfin[rb][k][sd][row][3];
        if (row - 1 >= 0 && row - 1 < ah) v += colfin[rb][k][sd][row - 1][4];
        if (row + 1 >= 0 && row + 1 < ah) v += colfin[rb][k][sd][row + 1][5];
        slot = (sd ? L::OR_ : L::OL) + row + 1;
      }
      if (slot >= 0) st(fp, tile_rec + (unsigned)(k * L::FR + slot) * 8u, v);
    }
  };

  // issue the loads of plane q: the own site's 38 populations and, when the ring of that plane comes from frames, its pieces
  // which: 0 both fluids, 1 the f half (with the frame pieces), 2 the g half; parts: 1 the own site's loads, 2 the frame pieces
  auto pull_plane = [&](int q, double (&f)[Q], double (&g)[Q], double (&hv)[2][4], const int which = 0, const int parts = 3) {
    const double* __restrict__ pl[3] = { S + (long long)wrapp(q - 1) * G.plane, S + (long long)wrapp(q) * G.plane,
                                         S + (long long)wrapp(q + 1) * G.plane };
    unsigned oo[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 3; ++b2) { oo[a][b2] = yo[a] + xo[b2]; asm volatile("" : "+v"(oo[a][b2])); }
    if (parts & 1) {
#pragma unroll
      for (int i = 0; i < Q; ++i) {
        const double* __restrict__ b = pl[1 - Vel::cz[i]] + (long long)i * G.vol;
        const unsigned o = oo[1 - Vel::cy[i]][1 + BFLBM_PX(Vel::cx[i])];
        // BFLBM_HO_NT_LOADS: 1 = the populations that do not travel in x with the non-temporal hint (every one of their lines is read by
        // exactly one tile, once; only the x-shifted row segments of the others share a line with the neighbouring tile), 2 = all of them
        const bool nt = BFLBM_HO_NT_LOADS == 2 || (BFLBM_HO_NT_LOADS == 1 && Vel::cx[i] == 0);
        if (which != 2) f[i] = nt ? ldnt(b, o) : ld(b, o);
        if (which != 1) g[i] = nt ? ldnt(b + (long long)Q * G.vol, o) : ld(b + (long long)Q * G.vol, o);
      }
    }
    if ((parts & 2) && which != 2 && Hg.use_frames && q >= fa && q <= fb && has_rtask) {
      const double* __restrict__ fp = Hg.fin + (long long)q * Hg.fplane;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned o = fo[j];
        asm volatile("" : "+v"(o));
        if (j < 2 || j < nfo) { hv[0][j] = ld(fp, o); hv[1][j] = ld(fp + L::FR, o); }
        else { hv[0][j] = 0.; hv[1][j] = 0.; }
      }
    }
  };
  double nf[Q], ng[Q], hvn[2][4];                            // the plane in flight
  pull_plane(qa - 1, nf, ng, hvn);
  // The first plane is waited for here, outside the loop: the wait at the loop head is then computed from the
  // back edge alone, where the stores of the previous position are younger than every load it needs, and no
  // longer drains those stores (it was vmcnt(0), the join of this path and the back edge).
  __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0), expcnt and lgkmcnt untouched

  int it = 0;
  // One march position. The two template flags say at compile time whether this position collides a plane and whether it
  // requests the next one. The two leading positions, the steady state and the last position are separate
  // instances, so that the steady-state loop has a single path of memory operations: the compiler's wait at the
  // loop head is then the one the back edge needs -- the loads, not the 19 stores issued after them (it was
  // vmcnt(0): the join with the paths that end in loads).
  auto position = [&](const int q, auto col_c, auto ldn_c) {
    constexpr bool do_collide = decltype(col_c)::value, load_next = decltype(ldn_c)::value;
    // quiet kernel: the own loads of the f half are spread over the relaxation of f (below); the noise kernel keeps the burst
    constexpr bool spread_f = do_collide && load_next && (MODE == 0 ? BFLBM_HO_SPREAD_F > 0 : BFLBM_HO_SPREAD_F1 > 0);
#ifdef BFLBM_STAMP
    unsigned long long ts[HO_NSTAMP] = {0};
#endif
    HO_STAMP(0);                                             // top of the position
    const int slot = it & 3;
    const double* __restrict__ pl[3] = { S + (long long)wrapp(q - 1) * G.plane, S + (long long)wrapp(q) * G.plane,
                                         S + (long long)wrapp(q + 1) * G.plane };
    // 1. plane q: pulled while the previous plane was collided
    double cf[Q], cg[Q], hv[2][4];
#pragma unroll
    for (int i = 0; i < Q; ++i) { cf[i] = nf[i]; cg[i] = ng[i]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { hv[0][j] = hvn[0][j]; hv[1][j] = hvn[1][j]; }
    const bool ring_from_frames = Hg.use_frames && q >= fa && q <= fb;       // uniform over the workgroup
    double zero = 0.0;
    asm volatile("" : "+v"(zero));
    auto density = [&](const double (&fs)[Q]) { double r = zero;
#pragma unroll
      for (int i = 0; i < Q; ++i) r += fs[i];
      return r; };
    if (ring_from_frames) {
      if (active) { rp[slot][0][lown] = density(cf); rp[slot][1][lown] = density(cg); }
      if (has_rtask) {
        double r0 = hv[0][0] + hv[0][1], r1 = hv[1][0] + hv[1][1];
        if (nfo > 2) { r0 += hv[0][2]; r1 += hv[1][2]; }
        if (nfo > 3) { r0 += hv[0][3]; r1 += hv[1][3]; }
        rp[slot][0][lhalo] = r0; rp[slot][1][lhalo] = r1;
      }
    } else {
      if (active) { rp[slot][0][lown] = density(cf); rp[slot][1][lown] = density(cg); }
      // pulled ring (chunk-boundary planes, first step, tile rows next to a lower last row): threads 0..nring-1 pull
      // one ring site, both fluids in one batch of 38 loads (wave-uniform base + 32-bit lane offset, like the own
      // loads); a rare path
      if (tid < (RAG ? 2 * (aw + 2) + 2 * ah : NRING)) {
        const int r = tid;
        int rx, ry;
        if (r < aw + 2) { rx = r; ry = 0; }
        else if (r < 2 * (aw + 2)) { rx = r - (aw + 2); ry = ah + 1; }
        else if (r < 2 * (aw + 2) + ah) { rx = 0; ry = r - 2 * (aw + 2) + 1; }
        else { rx = aw + 1; ry = r - 2 * (aw + 2) - ah + 1; }
        const int hx = wrapx(x0 + rx - 1), hy = wrapy(y0 + ry - 1);
        const unsigned hxo[3] = { (unsigned)wrapx(hx - 1) * 8u, (unsigned)hx * 8u, (unsigned)wrapx(hx + 1) * 8u };
        const unsigned hyo[3] = { (unsigned)(wrapy(hy - 1) * G.pitch) * 8u, (unsigned)(hy * G.pitch) * 8u, (unsigned)(wrapy(hy + 1) * G.pitch) * 8u };
        double t[2][Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) {
          unsigned o = hyo[1 - Vel::cy[i]] + hxo[1 + BFLBM_PX(Vel::cx[i])];
          asm volatile("" : "+v"(o));
          const double* __restrict__ b = pl[1 - Vel::cz[i]] + (long long)i * G.vol;
          t[0][i] = ld(b, o);
          t[1][i] = ld(b + (long long)Q * G.vol, o);
        }
        // all 38 requests are out before the first sum waits (left to itself the compiler issued one load, waited
        // for it, added, and went on to the next: 38 memory latencies in a row at four positions of every chunk)
        __builtin_amdgcn_sched_barrier(0);
        rp[slot][0][ry * LW + rx] = density(t[0]);
        rp[slot][1][ry * LW + rx] = density(t[1]);
      }
    }
    HO_STAMP(1);                                             // plane q arrived, densities summed
    __syncthreads();
    HO_STAMP(2);                                             // barrier passed
    // frames of plane q-3: finished at the previous position, combined across rows now
    finish(q - 3, (it & 1) ^ 1);
    // 3. collide plane q-1
    const int pc = q - 1;
    double mg[Q], jg[3];
    if (do_collide) {
      double pg[Q];
#pragma unroll
      for (int i = 0; i < Q; ++i) pg[i] = gl[i][tid];
      d_moments(pg, mg);
      d_momentum(pg, jg);
    }
    double mf[Q], jf[3];
    if (do_collide) {
      double pfl[Q];
#pragma unroll
      for (int i = 0; i < Q; ++i) pfl[i] = fl[i][tid];
      d_moments(pfl, mf);
      d_momentum(pfl, jf);
    }
#pragma unroll
    for (int i = 0; i < Q; ++i) gl[i][tid] = cg[i];
#pragma unroll
    for (int i = 0; i < Q; ++i) fl[i][tid] = cf[i];
    HO_STAMP(3);                                             // frames finished, held plane read as moments, new plane parked in LDS
    // the f half of the next plane: in flight while plane q-1 is collided
    if (load_next) pull_plane(q + 1, nf, ng, hvn, 1, spread_f ? 2 : 3);
    HO_STAMP(4);                                             // f half of plane q+1 requested
    if (do_collide) {
      const int sl[3] = { (it - 2) & 3, (it - 1) & 3, it & 3 };
      const double r = rp[sl[1]][0][lown], ph = rp[sl[1]][1][lown];
      double nb[Q], grad_rho[3], grad_phi[3];
#pragma unroll
      for (int i = 0; i < Q; ++i) nb[i] = rp[sl[1 + Vel::cz[i]]][0][lown + Vel::cy[i] * LW + Vel::cx[i]];
      d_gradient(P, nb, grad_rho);
#pragma unroll
      for (int i = 0; i < Q; ++i) nb[i] = rp[sl[1 + Vel::cz[i]]][1][lown + Vel::cy[i] * LW + Vel::cx[i]];
      d_gradient(P, nb, grad_phi);
      const int pcw = wrapp(pc);
      double fn3[3] = {0., 0., 0.}, gn3[3] = {0., 0., 0.};
      NoiseAmp NA; bflbm_rng_state rst;
      if (MODE == 1) {
        d_noise_amp(P, r, ph, r + ph, NA);
        d_noise_head(P, NA, global_site(G, x, y, pcw), noise_index, ntab, rst, fn3);
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3) gn3[k3] = -fn3[k3];
      }
      double* __restrict__ Dp = D + (long long)pcw * G.plane;
      unsigned o = yo[1] + xo[1];
      asm volatile("" : "+v"(o));
      unsigned os3[3] = { yo[1] + xo[0], o, yo[1] + xo[2] };       // store slots of populations with c_x = -1, 0, +1 (BFLBM_XSHIFT)
      if (BFLBM_XSHIFT) { asm volatile("" : "+v"(os3[0])); asm volatile("" : "+v"(os3[2])); }
      SiteHydro Hy;
      SiteRecip R;
      d_site_recips(P, r, ph, R);
      d_hydrovars_j(P, jf, jg, r, ph, grad_rho, grad_phi, fn3, gn3, Hy, R);
      double v_b[3];
      d_barycentric(r, ph, Hy, v_b, R);
      const int tp = pc - 1;                                   // plane whose sums become complete now
      const bool tp_ok = tp >= fa && tp <= fb;
      double* __restrict__ fp = Hg.fout + (long long)tp * Hg.fplane;
      const int wb = it & 1;
      const double zn[Q] = {0.};
      // sorts the 19 outputs of one fluid by destination and advances the z pipelines (see the header comment)
      // moments -> populations from the same terms as d_populations; a group of three populations that share a
      // destination (dy,dz) is stored and folded into its x bucket right after it is formed
      auto finish_fluid = [&](const double (&mom)[Q], const int k) {
        PopTerms T;
        d_population_terms(mom, T);
        double* __restrict__ Dk = Dp + (long long)(k * Q) * G.vol;
        // population stores carry the non-temporal hint (round 4: +0.9 % at 512^3, +0.6 % at 256^3, +1.6 % with noise, runs agreeing to
        // 0.1 %: the written lines are not read again before the next step, the L2 keeps the neighbours' shared lines instead)
        auto put = [&](int i, double v) {
          if (active) {
            double* __restrict__ q = reinterpret_cast<double*>(reinterpret_cast<char*>(Dk + (long long)i * G.vol) + os3[1 + BFLBM_SX(Vel::cx[i])]);
            if (BFLBM_HO_NT_STORES) __builtin_nontemporal_store(v, q); else *q = v;
          }
        };
        // x shifts see zeros from the idle lanes of a narrow tile (they hold a duplicate of the last site)
        auto shr = [&](double v) { return ho_shr((RAG && !active_x) ? 0.0 : v); };
        auto shl = [&](double v) { return ho_shl((RAG && !active_x) ? 0.0 : v); };
        auto diag = [&](int g, int j) {          // population j of plane g (d_populations)
          return j == 0 ? T.B[g] + T.p[g] + T.q[g] + T.r[g] : j == 1 ? T.B[g] - T.p[g] - T.q[g] + T.r[g]
               : j == 2 ? T.B[g] + T.p[g] - T.q[g] - T.r[g] : T.B[g] - T.p[g] + T.q[g] - T.r[g]; };
        const double o0 = T.rest, o1 = T.E[0] + T.O[0], o2 = T.E[0] - T.O[0];
        put(0, o0); put(1, o1); put(2, o2);
        const double x00 = o0 + shr(o1) + shl(o2);
        const double o3 = T.E[1] + T.O[1], o7 = diag(0, 0), o10 = diag(0, 3);
        put(3, o3); put(7, o7); put(10, o10);
        const double xp0 = o3 + shr(o7) + shl(o10);
        const double o4 = T.E[1] - T.O[1], o9 = diag(0, 2), o8 = diag(0, 1);
        put(4, o4); put(9, o9); put(8, o8);
        const double xm0 = o4 + shr(o9) + shl(o8);
        const double o5 = T.E[2] + T.O[2], o15 = diag(2, 0), o18 = diag(2, 2);
        put(5, o5); put(15, o15); put(18, o18);
        const double x0p = o5 + shr(o15) + shl(o18);
        const double o6 = T.E[2] - T.O[2], o17 = diag(2, 3), o16 = diag(2, 1);
        put(6, o6); put(17, o17); put(16, o16);
        const double x0m = o6 + shr(o17) + shl(o16);
        const double o11 = diag(1, 0), o12 = diag(1, 1), o13 = diag(1, 2), o14 = diag(1, 3);
        put(11, o11); put(12, o12); put(13, o13); put(14, o14);
        // what leaves the tile in x (column lanes only; side_x picks the lane's outward direction)
        const double oxp = side_x ? o15 : o18, ox0 = side_x ? o1 : o2, oxm = side_x ? o17 : o16, oyp = side_x ? o7 : o10, oym = side_x ? o9 : o8;
        if (!RAG) {
        // travelling sum towards the nearer tile edge (rows 0,1: -y; rows TY-2,TY-1: +y): contributions to
        // planes p+1, p, p-1 enter a two-stage pipeline, what leaves it is complete for plane p-1
        const double np = row_down ? o14 : o11, n0 = row_down ? xm0 : xp0, nm = row_down ? o12 : o13;
        const double fin_nb = anb[k][1] + nm;
        anb[k][1] = anb[k][0] + n0;
        anb[k][0] = np;
        double hand = fin_nb;                     // rows next to an edge row hand their sum to the edge row
        if (is_edge) {                            // (lane predicate) own sums of the edge row; its travelling sum is the ring's
          const double fin_self = accs[1][k][side_y][lane] + x0m;
          accs[1][k][side_y][lane] = accs[0][k][side_y][lane] + x00;
          accs[0][k][side_y][lane] = x0p;
          hand = fin_self;
          if (tp_ok) st(fp, tile_rec + (unsigned)(k * L::FR + (side_y ? L::OT : L::OB) + lane) * 8u, fin_nb);
        }
        if (TY == 4 || row_kind) exch[wb][k][side_y][edge_row ? 0 : 1][lane] = hand;
        } else {
        // ragged tile (1-4 rows): both travelling sums run in every row; what a row does with them is its role(s)
        const double fin_dn = anb[k][1] + o12;    // -y: lands on row ty-1 (the ring row below for row 0)
        anb[k][1] = anb[k][0] + xm0;
        anb[k][0] = o14;
        const double fin_up = aup[k][1] + o13;    // +y: lands on row ty+1 (the ring row above for row ah-1)
        aup[k][1] = aup[k][0] + xp0;
        aup[k][0] = o11;
        if (r_bot || r_top) {                     // own sums of an edge row (one pipeline: a single-row tile is both edges)
          const int sd = r_bot ? 0 : 1;
          const double fin_self = accs[1][k][sd][lane] + x0m;
          accs[1][k][sd][lane] = accs[0][k][sd][lane] + x00;
          accs[0][k][sd][lane] = x0p;
          if (r_bot) { exch[wb][k][0][0][lane] = fin_self; if (tp_ok) st(fp, tile_rec + (unsigned)(k * L::FR + L::OB + lane) * 8u, fin_dn); }
          if (r_top) { exch[wb][k][1][0][lane] = fin_self; if (tp_ok) st(fp, tile_rec + (unsigned)(k * L::FR + L::OT + lane) * 8u, fin_up); }
        }
        if (r_hdn) exch[wb][k][0][1][lane] = fin_dn;
        if (r_hup) exch[wb][k][1][1][lane] = fin_up;
        }
        if (col_lane) {
          // kinds: 0..2 sums travelling dy = 0,+1,-1 inside the column; 3..5 what leaves the tile in x with dy = 0,+1,-1
          const double vp[6] = { x0p, o11, o14, oxp, 0., 0. };
          const double v0[6] = { x00, xp0, xm0, ox0, oyp, oym };
          const double vm[6] = { x0m, o13, o12, oxm, 0., 0. };
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            colfin[wb][k][side_x][ty][j] = colacc[1][k][side_x][ty][j] + vm[j];
            colacc[1][k][side_x][ty][j] = colacc[0][k][side_x][ty][j] + v0[j];
            colacc[0][k][side_x][ty][j] = vp[j];
          }
        }
      };
      HO_STAMP(5);                                           // gradient, noise head, projection done
      // Round 3 (tools/ho_stamps.py): a lone wave that issues its 19 requests as one burst stands at the issue for 2400 of
      // the 18500 clocks of a position -- the burst is longer than the CU's request queue, and an in-order wave cannot
      // compute while it waits for queue space.  In the quiet kernel the 19 own loads of the f half are therefore requested
      // one every BFLBM_HO_SPREAD_F (28) VALU instructions of the relaxation of f; the order is pinned with
      // sched_group_barrier (the compiler hoists independent loads to the top of the block otherwise).  512^3: 7817 ->
      // 8484 and 8250 -> 8442 MLUPS on two boxes, 256^3 +2.5 %; spacings of 20 and 36 and all 38 loads spread were slower
      // than the burst (NOTES.md section 3.1f).  The noise kernel: every spacing lost with round 3's generator; with round 4's
      // (11 instructions per normal) one request every 44 instructions is +2 % at both sizes (profiles/r04_noise_generator_ab.txt).
      if (spread_f) pull_plane(q + 1, nf, ng, hvn, 1, 1);
      if (MODE == 1) d_relax_generated(P, mf, r, v_b, Hy.uf, Hy.af, P.inv_tau_f_bar, fn3, NA.sr, ntab, rst, R.cs4);
      else           d_relax<false>(P, mf, r, v_b, Hy.uf, Hy.af, P.inv_tau_f_bar, zn, R.cs4);
      if (spread_f) {
#pragma unroll
        for (int s_ = 0; s_ < Q; ++s_) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                      // one vector-memory read,
          __builtin_amdgcn_sched_group_barrier(0x002, MODE == 0 ? BFLBM_HO_SPREAD_F : BFLBM_HO_SPREAD_F1, 0);      // then this many VALU instructions
        }
      }
      HO_STAMP(6);                                           // fluid f relaxed
      finish_fluid(mf, 0);
      if (load_next) pull_plane(q + 1, nf, ng, hvn, 2);       // the g half of the next plane: spreads the requests over the march position (+2.9 % at 512^3)
      HO_STAMP(7);                                           // f stored, frames produced, g half requested
      if (MODE == 1) d_relax_generated(P, mg, ph, v_b, Hy.ug, Hy.ag, P.inv_tau_g_bar, gn3, NA.sp, ntab, rst, R.cs4);
      else           d_relax<false>(P, mg, ph, v_b, Hy.ug, Hy.ag, P.inv_tau_g_bar, zn, R.cs4);
      HO_STAMP(8);                                           // fluid g relaxed
      finish_fluid(mg, 1);
      HO_STAMP(9);                                           // g stored
#ifdef BFLBM_STAMP
      if (blockIdx.x == (gridDim.x > 777u ? 777u : gridDim.x / 2u + 1u) && it >= 40 && it < 40 + HO_STAMP_POS && lane == 0) {
#pragma unroll
        for (int k = 0; k < HO_NSTAMP; ++k) g_ho_stamps[((it - 40) * 4 + ty) * HO_NSTAMP + k] = ts[k];
      }
#endif
    } else if (load_next) {
      // the first two positions of a chunk collide nothing: the g half goes now, and is waited for here (as in
      // the prologue: keeps vmcnt(0) out of the loop head)
      pull_plane(q + 1, nf, ng, hvn, 2);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
  };
  using std::bool_constant;
  {
    int q = qa - 1;                                          // qb >= qa + 1: both leading positions request a plane
    for (int k = 0; k < 2; ++k, ++q, ++it) position(q, bool_constant<false>{}, bool_constant<true>{});
    for (; q < qb; ++q, ++it) position(q, bool_constant<true>{}, bool_constant<true>{});
    position(qb, bool_constant<true>{}, bool_constant<false>{});
    ++it;
  }
  // the last complete plane (qb-2) was finished at the last position; combine it across rows
  __syncthreads();
  finish(qb - 2, (it & 1) ^ 1);
}

#ifndef BFLBM_HO_TY
#define BFLBM_HO_TY 4
#endif

// lattices the hand-over kernel takes: at least one whole tile; a narrower last tile column / lower last tile row needs
// two lanes / rows (a ring site owned by a tile ONE site wide or high also collects from the tile beyond it, which the
// 3 x 3 lookup of the consumer does not reach, and its two edge roles must be different lanes / rows)
static inline bool handover_ok(const Geo& G) {
  constexpr int TY = BFLBM_HO_TY;
  return G.nx >= 64 && G.nx % 64 != 1 && G.ny >= TY && G.ny % TY != 1;
}
static inline bool handover_ragged(const Geo& G) { return G.nx % 64 != 0 || G.ny % BFLBM_HO_TY != 0; }
static inline size_t handover_frame_doubles(const Geo& G) {
  constexpr int TY = BFLBM_HO_TY;
  return (size_t)((G.nx + 63) / 64) * (size_t)((G.ny + TY - 1) / TY) * HoLayout<TY>::REC * (size_t)G.nzs;
}

struct HoSig { int pa = -1, pb = -1, lz = -1, nchunks = -1, cstride = -1; long long step = -1;
  bool same_geometry(const HoSig& o) const { return pa == o.pa && pb == o.pb && lz == o.lz && nchunks == o.nchunks && cstride == o.cstride; } };

// Chunking and workgroup order of one launch over the storage planes [pa, pb) (pair_len > 0: the two boundary plane
// pairs of a slab, one chunk each).  One workgroup is resident per CU, so a launch runs in rounds of `slots` workgroups
// and costs about rounds x (planes per chunk + 1) march positions; the chunk count minimises that.  Round 4 scanned the
// chunk count at 256^3 ... 512^3 on two boxes (profiles/r04_chunk_scan.txt): a model fitted on the first box
// ((rounds + tail) x (planes + 4): 448^3 +5 % with 7 chunks, 512^3 +5 % with 4) changed nothing on the second (every
// lattice within the +-2 % process-to-process scatter, 512x512x128 2 % SLOWER with its choice), so the rule stays.
static inline void handover_plan(const Geo& G, int pa, int pb, int pair_len, FusedGrid& F) {
  constexpr int TX = 64, TY = BFLBM_HO_TY;
  F.ntx = (G.nx + TX - 1) / TX; F.nty = (G.ny + TY - 1) / TY;
  F.ncols = F.ntx * F.nty;
  F.pa = pa; F.pb = pb;
  const int np = pb - pa;
  static const int want_env = [] { const char* e = getenv("BFLBM_FUSED_WG"); return e ? atoi(e) : 0; }();
  const int slots = g_fused_ncu > 0 ? g_fused_ncu : 256;        // one workgroup per CU
  static const int min_slab_rounds = [] { const char* e = getenv("BFLBM_SLAB_ROUNDS"); return e && atoi(e) > 0 ? atoi(e) : 3; }();
  F.row0 = 0;
  const int maxchunks = std::max(1, np / 4);                     // a chunk shorter than 4 planes has no complete frame
  int nchunks;
  if (want_env > 0) {
    nchunks = std::min(maxchunks, std::max(1, (want_env + F.ncols - 1) / F.ncols));
  } else {
    long long best = -1; nchunks = 1;
    for (int k = 1; k <= maxchunks; ++k) {
      const int lz = (np + k - 1) / k, chunks = (np + lz - 1) / lz;
      if (chunks != k) continue;
      if (G.zwrap && lz > 256 && k < maxchunks) continue;   // one 512-plane march per column was A/B-tested: -1 %
      const long long total = (long long)F.ncols * chunks, rounds = (total + slots - 1) / slots;
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
  { static const int sx_env = [] { const char* e = getenv("BFLBM_MAP_SX"); return e ? atoi(e) : 0; }(); F.sx = sx_env > 0 ? sx_env : F.ntx; }
}

// fin/fout: frame buffers of the state read / written.  sig_in: what wrote fin (step == steps-1 required);
// sig_out receives this launch.  returns the launch's error code
static inline hipError_t handover_launch(const double* S, double* D, const double* fin, double* fout, const Geo& G, const DevParams& P,
                                  int pa, int pb, long long steps, const HoSig& sig_in, HoSig& sig_out, hipStream_t stream, int pair_len = 0, int mode = 0) {
  constexpr int TX = 64, TY = BFLBM_HO_TY;
  FusedGrid F;
  handover_plan(G, pa, pb, pair_len, F);
  sig_out.pa = pa; sig_out.pb = pb; sig_out.lz = F.lz; sig_out.nchunks = F.nchunks; sig_out.cstride = F.cstride; sig_out.step = steps;
  HoGrid Hg;
  Hg.fin = fin; Hg.fout = fout;
  Hg.fplane = (long long)F.ncols * HoLayout<TY>::REC;
  Hg.use_frames = (sig_in.step == steps - 1 && sig_in.same_geometry(sig_out)) ? 1 : 0;
  dim3 grid((unsigned)(F.per_xcd * 8)), block(TX * TY);
  const uint32_t nidx = (uint32_t)steps;
  static const bool force_rag = [] { const char* e = getenv("BFLBM_FORCE_RAG"); return e && atoi(e) != 0; }();   // diagnostics: what the ragged-tile code costs on full tiles
  const bool rag = handover_ragged(G) || force_rag;
  if (mode == 1) { if (rag) hipLaunchKernelGGL((k_fused_ho<TY, 1, true>), grid, block, 0, stream, S, D, G, P, F, Hg, nidx);
                   else     hipLaunchKernelGGL((k_fused_ho<TY, 1, false>), grid, block, 0, stream, S, D, G, P, F, Hg, nidx); }
  else           { if (rag) hipLaunchKernelGGL((k_fused_ho<TY, 0, true>), grid, block, 0, stream, S, D, G, P, F, Hg, nidx);
                   else     hipLaunchKernelGGL((k_fused_ho<TY, 0, false>), grid, block, 0, stream, S, D, G, P, F, Hg, nidx); }
  return hipGetLastError();
}

#endif  // BFLBM_HANDOVER_H_
