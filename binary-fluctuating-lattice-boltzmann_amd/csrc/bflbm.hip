// bflbm.hip -- C-ABI implementation (include/bflbm.h) over the gfx950 kernels.
// Host side: context, HBM layout, launches, FAB <-> slab copies.  No torch, no oracle.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bflbm.h"
#define BFLBM_NHYDRO_ 22
#include "bflbm_kernels.h"
#include "bflbm_fused.h"
#include "bflbm_handover.h"
#ifdef BFLBM_CALIBRATION
#include "../../tools/calibration_kernels.h"
#endif

namespace {

thread_local std::string g_err;

int fail(const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return 1;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

const double kW[Q] = {1./3., 1./18.,1./18.,1./18.,1./18.,1./18.,1./18.,
                      1./36.,1./36.,1./36.,1./36.,1./36.,1./36.,1./36.,1./36.,1./36.,1./36.,1./36.,1./36.};
// mode norms b[a], LBM_d3q19.H:56-76
const double kB[Q] = {1.0, 1./3., 1./3., 1./3., 2./3., 4./3., 4./9., 1./9., 1./9., 1./9.,
                      2./3., 2./3., 2./3., 2./9., 2./9., 2./9., 2.0, 4./3., 4./9.};
const int kCZh[Q] = BFLBM_CZ;

// Uniform sub-expressions in the reference's operation order (see DevParams).
void derive(const bflbm_params& p, DevParams& d) {
  d.cs2 = p.cs2;
  d.cs4 = p.cs2 * p.cs2;                       // LBM_d3q19.H:7 (1./3.)*(1./3.)
  d.tau_f = p.tau_f; d.tau_g = p.tau_g; d.alpha0 = p.alpha0; d.kBT = p.kBT;
  d.wcs1 = kW[1] / p.cs2;
  d.wcs2 = kW[7] / p.cs2;
  d.neg_cs2_alpha0 = -p.cs2 * p.alpha0;
  d.kf = 0.5 / (p.tau_f + 0.5);
  d.kg = 0.5 / (p.tau_g + 0.5);
  const double tau_f_bar = p.tau_f * (1. + 0.5 / p.tau_f);
  const double tau_g_bar = p.tau_g * (1. + 0.5 / p.tau_g);
  d.inv_tau_f_bar = 1. / tau_f_bar;
  d.inv_tau_g_bar = 1. / tau_g_bar;
  d.coefC = 1. / p.cs2;
  d.coefC_cs2 = d.coefC * p.cs2;
  d.two_cs4 = 2. * d.cs4;
  d.six_cs4 = 6. * d.cs4;
  d.modifactor = 1. / (1. + 1. / (2. * p.tau_f));
  d.mod2 = d.modifactor * 2.;
  const double tfb = 1. / (p.tau_f + 0.5);     // LBM_binary.H:79-82 (tau_g_bar = tau_f_bar)
  const double tfb2 = tfb * tfb;
  const double base = 2. * (tfb - 0.5 * tfb2);
  d.amp_j = base * p.kBT;
  for (int a = 0; a < Q; ++a) {
    d.amp_f[a] = base * p.kBT / p.cs2 * kB[a];
    d.amp_g[a] = d.amp_f[a];
  }
  const int rep[6] = {4, 5, 6, 7, 13, 16};       // one mode per distinct norm b[a]
  for (int k = 0; k < 6; ++k) d.samp[k] = std::sqrt(d.amp_f[rep[k]]);
  d.seed_lo = (uint32_t)p.seed;
  d.seed_hi = (uint32_t)(p.seed >> 32);
  d.noise_on = (p.kBT != 0.) ? 1 : 0;
}

}  // namespace

struct bflbm_ctx {
  bflbm_params prm;
  DevParams dp;
  bflbm_domain dom;
  Geo G;
  int nzl = 0;
  double* S[2] = {nullptr, nullptr};
  int cur = 0;
  double* rho = nullptr;
  double* phi = nullptr;
  double* injf = nullptr;
  double* injg = nullptr;
  bool inject = false;
  double* partial = nullptr;
  size_t partial_n = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  long long steps = 0;
  int schedule = 2;              // 0 two-pass, 1 fused plane-marching (pulled ring), 2 auto (default), 3 fused with density hand-over
  double* frames[2] = {nullptr, nullptr};   // schedule 3: tile-boundary density frames of S[0], S[1] (bflbm_handover.h)
  bool frames_unavailable = false;          // their allocation failed once: auto stays on the bit-exact schedules
  HoSig fsig[2][2];              // [state buffer][0 interior sweep, 1 boundary pairs]: the launch that wrote the frames
  bool step_open = false;
  double total_max = -1.;        // largest |rho + phi| of the state an upload made resident (< 0: analytic init, the parameters say it)
  float tune_ms[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // bflbm_tune_placement: step time of every candidate allocation tried, and which was kept
  int tune_n = 0, tune_kept = 0;
  bool density_valid = false;   // rho/phi arrays hold the densities of the resident state
  size_t bytes = 0;
  // USE_REF_STATE (LBM_binary.H:12, :92-107): noise amplitudes from an equilibrium state
  double* ref[3] = {nullptr, nullptr, nullptr};   // rho_eq, phi_eq, rhot_eq over the GLOBAL lattice
  bool ref_on = false;
  double com_ref[3] = {0., 0., 0.};
  int ref_kind = 0;             // what the call that made the resident state hands to thermal_noise:
  long long ref_kind_step = -1; //   0 COM - com_ref (:588, :653), 1 zero (:690, :739), 2 absolute COM (:623-625)
  double com[3] = {0., 0., 0.}; // global centre of mass (update_com) of the resident state
  bool com_valid = false;
};

namespace {

dim3 plane_grid(const bflbm_ctx* c, int nplanes) {
  return dim3((unsigned)((c->G.plane + 255) / 256), (unsigned)nplanes, 1);
}

// planes [pa,pb) of rho,phi from S[cur]
int launch_density(bflbm_ctx* c, int pa, int pb) {
  if (pb <= pa) return 0;
  hipLaunchKernelGGL(k_density, plane_grid(c, pb - pa), dim3(256), 0, c->stream, c->S[c->cur], c->rho, c->phi, c->G, pa);
  HIP_TRY(hipGetLastError());
  return 0;
}

inline bool ref_active(const bflbm_ctx* c) { return c->ref_on && c->dp.noise_on && !c->inject; }

// shift of the reference-state lookup for the noise of the resident state
RefState ref_state(const bflbm_ctx* c) {
  RefState R;
  R.rho = c->ref[0]; R.phi = c->ref[1]; R.rhot = c->ref[2];
  R.on = ref_active(c) ? 1 : 0;
  R.sx = R.sy = R.sz = 0;
  if (!R.on) return R;
  const int kind = (c->steps == c->ref_kind_step) ? c->ref_kind : 0;
  if (kind == 1) return R;
  double rel[3];
  for (int d = 0; d < 3; ++d) rel[d] = c->com[d] - (kind == 0 ? c->com_ref[d] : 0.);
  // static_cast<int> (:94-96), then reduced modulo the lattice: the reference wraps the shifted index ONCE
  // (:98-103), which is only in range for |shift| < n -- beyond that it reads out of bounds.  The reduction
  // changes nothing for |shift| < n and keeps the kernel's single wrap sufficient for any centre of mass
  // (prepare_ref refuses a non-finite one).
  const int n[3] = { c->G.nx, c->G.ny, c->G.nz };
  int sh[3];
  for (int d = 0; d < 3; ++d) {
    const double t = std::trunc(rel[d]);
    sh[d] = (int)std::fmod(t, (double)n[d]);                       // in (-n, n), sign of the shift kept
  }
  R.sx = sh[0]; R.sy = sh[1]; R.sz = sh[2];
  return R;
}

int launch_collide(bflbm_ctx* c, int pa, int pb) {
  if (pb <= pa) return 0;
  const RefState Rf = ref_state(c);
  const double* src = c->S[c->cur];
  double* dst = c->S[1 - c->cur];
  const uint32_t idx = (uint32_t)c->steps;
  dim3 g = plane_grid(c, pb - pa), b(256);
  if (c->inject)
    hipLaunchKernelGGL((k_collide<true, true>), g, b, 0, c->stream, src, dst, c->rho, c->phi, c->injf, c->injg, c->G, c->dp, pa, idx, Rf);
  else if (c->dp.noise_on)
    hipLaunchKernelGGL((k_collide<true, false>), g, b, 0, c->stream, src, dst, c->rho, c->phi, c->injf, c->injg, c->G, c->dp, pa, idx, Rf);
  else
    hipLaunchKernelGGL((k_collide<false, false>), g, b, 0, c->stream, src, dst, c->rho, c->phi, c->injf, c->injg, c->G, c->dp, pa, idx, Rf);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Frames of schedule 3: 2 x nzs x ntiles x 544 doubles = 5.6 % of the state (4.6 GB at 512^3).  An explicit request for
// schedule 3 fails loudly when they cannot be allocated; `auto` remembers the failure and stays on the bit-exact
// schedules (a lattice that fitted in HBM before the hand-over existed keeps running).
int ensure_frames(bflbm_ctx* c, bool quiet = false) {
  if (c->frames[0]) return 0;
  if (c->frames_unavailable) return quiet ? 1 : fail("hand-over frames could not be allocated earlier (out of device memory)");
  const size_t nb = handover_frame_doubles(c->G) * sizeof(double);
  static const size_t fail_above = [] { const char* e = getenv("BFLBM_DEBUG_FRAMES_LIMIT"); return e ? (size_t)atoll(e) : (size_t)0; }();   // test hook
  hipError_t e = (fail_above && 2 * nb > fail_above) ? hipErrorOutOfMemory : hipMalloc((void**)&c->frames[0], 2 * nb);
  if (e != hipSuccess) {
    c->frames[0] = nullptr;
    (void)hipGetLastError();                                     // clear the sticky error
    c->frames_unavailable = true;
    return quiet ? 1 : fail("hand-over frames: %s (%zu bytes)", hipGetErrorString(e), 2 * nb);
  }
  c->frames[1] = c->frames[0] + handover_frame_doubles(c->G);
  c->bytes += 2 * nb;
  return 0;
}

int launch_handover(bflbm_ctx* c, int pa, int pb, int pair_len = 0) {
  if (pb <= pa) return 0;
  if (ensure_frames(c)) return 1;
  const int kind = pair_len > 0 ? 1 : 0;
  const hipError_t e = handover_launch(c->S[c->cur], c->S[1 - c->cur], c->frames[c->cur], c->frames[1 - c->cur], c->G, c->dp, pa, pb,
                                       c->steps, c->fsig[c->cur][kind], c->fsig[1 - c->cur][kind], c->stream, pair_len, c->dp.noise_on ? 1 : 0);
  return e != hipSuccess ? fail("hand-over launch failed: %s", hipGetErrorString(e)) : 0;
}

int launch_fused(bflbm_ctx* c, int pa, int pb, int pair_len = 0) {
  if (pb <= pa) return 0;
  const hipError_t e = fused_launch(c->S[c->cur], c->S[1 - c->cur], c->injf, c->injg, c->G, c->dp, pa, pb,
                                    (uint32_t)c->steps, c->inject ? 2 : (c->dp.noise_on ? 1 : 0), c->stream, pair_len);
  return e != hipSuccess ? fail("fused launch failed: %s", hipGetErrorString(e)) : 0;
}

// The range of model parameters inside which `auto` uses the hand-over kernel.  Schedule 3 differs from the reference's
// doubles by a re-ordered sum of 19 numbers at tile-edge sites, i.e. by what a one-ulp change of the state does, and
// what becomes of that is the trajectory's own conditioning: measured against the oracle (tools/ho_stress.py, NOTES.md
// section 3.1c) the difference equals the oracle's own response to a one-ulp perturbation case by case, which is inside
// the north-star tolerance wherever the reference run is stable and unbounded where the reference itself diverges
// (alpha0 = 4 with rho_hi = 3: NaN within 10-40 steps on the CPU path too).  Those diverging runs all have an
// interaction strength alpha0 (rho_hi + rho_lo) >= 7.5; every parameter set the reference ships or its notebooks record
// has <= 5.1 (header defaults 4 x 1, Surface_Tension.ipynb 1.5 x 3 and 1.7 x 3).  `auto` stays bit-exact above 6.
// rho_hi + rho_lo is the total density rho + phi of every site of an analytic init (LBM_binary.H:681, :731: phi =
// rho_hi + rho_lo - rho).  After LBM_init(f0, g0) (:632-661, the restart path) the resident densities need not relate to
// the parameters at all, so the bound is then taken on the state: the largest |rho + phi| the upload made resident
// (reduced once in bflbm_commit_upload); a non-finite value keeps auto bit-exact.
inline bool handover_contract_params(const bflbm_ctx* c) {
  const bflbm_params& p = c->prm;
  const double total = c->total_max >= 0. ? c->total_max : std::fabs(p.rho_hi) + std::fabs(p.rho_lo);
  return std::fabs(p.alpha0) * total <= 6.0;       // false for NaN
}

// Where `auto` expects schedule 3 to be the faster one (A/B on MI355X, tools/ragged_ab.sh, NOTES.md section 3.1d):
// it needs marches of at least 16 planes per workgroup (64^3: 16 tile columns cut into chunks of 4 planes, of whose 6
// positions only 2 read frames: -13 %; 64 x 64 x 256 and 128 x 128 x 64, chunks of 16: equal or better), and at zero
// noise, where the alternative is the one-pass schedule 1, a last tile column that is not mostly idle lanes (96^3: 75 %
// of the lanes busy, -9 %; 200^3: 78 %, +4 %; 300^3: +13 %; 250^3 with a last tile row of two rows: +9 %).  With noise the
// alternative is the two-pass schedule and the hand-over kernel wins on every ragged lattice measured (250^3 +25 %).
// BFLBM_AUTO_MIN_LZ overrides the 16.
inline bool handover_worthwhile(const bflbm_ctx* c, bool noisy) {
  static const int min_lz = [] { const char* e = getenv("BFLBM_AUTO_MIN_LZ"); return e ? atoi(e) : 16; }();
  const int lo = c->G.H, hi = c->G.H + c->nzl;
  const int a = c->G.zwrap ? lo : lo + 2, b = c->G.zwrap ? hi : hi - 2;   // the interior sweep
  if (b - a < min_lz || b - a < 4) return false;
  FusedGrid F;
  handover_plan(c->G, a, b, 0, F);
  if (F.lz < min_lz) return false;
  if (noisy) return true;
  const double lanes_busy = (double)c->G.nx / (64.0 * F.ntx);
  return lanes_busy >= 0.77;
}

// Which bit-exact schedule `auto` takes at zero noise.  The one-pass kernel needs a workgroup per CU to be worth its two
// planes of look-ahead per chunk: a lattice whose tile columns x 2-plane chunks do not fill the device runs the two-pass
// schedule faster (same doubles): 32^3 2440 against 1000 MLUPS, 48^3 5070 / 3350, the reference's 8 x 256 x 64 flat-interface
// box 5910 / 3530; 64^3 (exactly 256 workgroups) and 96^3 are equal (tools/size_sweep.sh, NOTES.md section 3.1d).
inline int exact_quiet_schedule(const bflbm_ctx* c) {
  const int lo = c->G.H, hi = c->G.H + c->nzl;
  const int a = c->G.zwrap ? lo : lo + 2, b = c->G.zwrap ? hi : hi - 2;   // the interior sweep
  if (b - a < 2) return 0;                       // a slab of 4 or 5 planes is all boundary pairs
  FusedGrid F;
  (void)fused_plan(c->G, a, b, 0, 0, F);
  const int ncu = g_fused_ncu > 0 ? g_fused_ncu : 256;
  return F.total < ncu ? 0 : 1;
}

// 0 two-pass, 1 fused (pulled ring), 3 hand-over.  The bit-exact choice is 1 at zero noise and 0 with noise.
// auto (2): the pipelined hand-over kernel wherever the lattice has full 64 x 4 tiles with distinct neighbours (it
// generates thermal noise itself but takes no injected noise) where it is the faster one, the parameters are inside the
// range above and the frames fit in memory; BFLBM_AUTO_EXACT=1 keeps auto on the bit-exact schedules with and without noise.
inline int resolved_schedule(const bflbm_ctx* c) {
  if (ref_active(c)) return 0;                   // needs the densities and their centre of mass first
  const bool noisy = c->dp.noise_on || c->inject;
  const int exact = noisy ? 0 : 1;
  if (c->schedule == 3) return (!c->inject && handover_ok(c->G)) ? 3 : exact;
  if (c->schedule != 2) return c->schedule;
  static const int auto_noise_fused = [] { const char* e = getenv("BFLBM_AUTO_NOISE_HANDOVER"); return e ? atoi(e) != 0 : true; }();
  static const int auto_exact = [] { const char* e = getenv("BFLBM_AUTO_EXACT"); return e && atoi(e) != 0; }();
  const int auto_exact_choice = noisy ? 0 : exact_quiet_schedule(c);
  if (auto_exact || c->inject || c->frames_unavailable || !handover_ok(c->G) || !handover_contract_params(c)) return auto_exact_choice;
  if (!handover_worthwhile(c, noisy)) return auto_exact_choice;
  if (noisy && !auto_noise_fused) return 0;
  return 3;
}

// the slab's own planes are [H, H+nzl)
inline int own_lo(const bflbm_ctx* c) { return c->G.H; }
inline int own_hi(const bflbm_ctx* c) { return c->G.H + c->nzl; }

int ensure_density(bflbm_ctx* c) {
  if (c->density_valid) return 0;
  const int ext = c->G.zwrap ? 0 : 1;
  if (launch_density(c, own_lo(c) - ext, own_hi(c) + ext)) return 1;
  c->density_valid = true;
  return 0;
}

struct Overlap { int x0, x1, y0, y1, z0, z1; bool empty; };

// valid cells of the box inside the lattice and inside the z range [z0, z1)
Overlap overlap(const bflbm_ctx* c, const bflbm_fab* b, int z0, int z1) {
  Overlap o;
  o.x0 = std::max(b->vlo[0], 0); o.x1 = std::min(b->vhi[0], c->G.nx - 1);
  o.y0 = std::max(b->vlo[1], 0); o.y1 = std::min(b->vhi[1], c->G.ny - 1);
  o.z0 = std::max(b->vlo[2], z0); o.z1 = std::min(b->vhi[2], z1 - 1);
  o.empty = (o.x0 > o.x1 || o.y0 > o.y1 || o.z0 > o.z1);
  return o;
}

int check_fab(const bflbm_fab* b) {
  if (!b) return fail("null bflbm_fab");
  for (int d = 0; d < 3; ++d) {
    if (b->hi[d] < b->lo[d]) return fail("bflbm_fab: hi < lo in dim %d", d);
    if (b->vlo[d] < b->lo[d] || b->vhi[d] > b->hi[d]) return fail("bflbm_fab: valid region outside the allocated box in dim %d", d);
  }
  return 0;
}

// copy ncomp components between a host FAB and a slab-layout device array (component stride
// dvol, plane offset dplane0 = storage plane of global z0).  to_device selects direction.
// whole_lattice: the device array covers global z in [0, nz) from plane 0 (reference-state fields)
// instead of the slab's own planes.
// dense: the device array has rows of nx elements (observable outputs, injected noise, reference state)
// instead of the resident arrays' pitch.
int copy_fab(bflbm_ctx* c, double* host, const bflbm_fab* b, int ncomp, double* dev, long long dvol, int dplane0, bool to_device,
             bool whole_lattice = false, bool dense = false) {
  if (check_fab(b)) return 1;
  const int zlo = whole_lattice ? 0 : c->dom.z0, zhi = whole_lattice ? c->G.nz : c->dom.z1;
  const Overlap o = overlap(c, b, zlo, zhi);
  if (o.empty) return 0;
  const size_t fx = (size_t)(b->hi[0] - b->lo[0] + 1), fy = (size_t)(b->hi[1] - b->lo[1] + 1), fz = (size_t)(b->hi[2] - b->lo[2] + 1);
  const size_t fvol = fx * fy * fz;
  for (int k = 0; k < ncomp; ++k) {
    hipMemcpy3DParms p;
    memset(&p, 0, sizeof p);
    hipPitchedPtr hp = make_hipPitchedPtr(host + (size_t)k * fvol, fx * sizeof(double), fx * sizeof(double), fy);
    const size_t drow = (size_t)(dense ? c->G.nx : c->G.pitch) * sizeof(double);
    hipPitchedPtr dp = make_hipPitchedPtr(dev + (size_t)k * dvol, drow, drow, (size_t)c->G.ny);
    hipPos hpos = make_hipPos((size_t)(o.x0 - b->lo[0]) * sizeof(double), (size_t)(o.y0 - b->lo[1]), (size_t)(o.z0 - b->lo[2]));
    hipPos dpos = make_hipPos((size_t)o.x0 * sizeof(double), (size_t)o.y0, (size_t)(o.z0 - zlo + dplane0));
    p.extent = make_hipExtent((size_t)(o.x1 - o.x0 + 1) * sizeof(double), (size_t)(o.y1 - o.y0 + 1), (size_t)(o.z1 - o.z0 + 1));
    if (to_device) { p.srcPtr = hp; p.srcPos = hpos; p.dstPtr = dp; p.dstPos = dpos; p.kind = hipMemcpyHostToDevice; }
    else           { p.srcPtr = dp; p.srcPos = dpos; p.dstPtr = hp; p.dstPos = hpos; p.kind = hipMemcpyDeviceToHost; }
    HIP_TRY(hipMemcpy3DAsync(&p, c->stream));
  }
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// (component, storage plane) list of what crosses a z face.  pack side s gathers from the
// slab's own planes; unpack side s scatters into the halo planes on that side.
void halo_table(const bflbm_ctx* c, int kind, int side, bool pack, HaloTable& T) {
  const int H = c->G.H, nzl = c->nzl;
  int e = 0;
  if (kind == BFLBM_HALO_UPLOAD) {
    int plane;
    if (pack) plane = (side == 0) ? H : H + nzl - 1;
    else      plane = (side == 0) ? H - 1 : H + nzl;
    for (int k = 0; k < 2 * Q; ++k) { T.comp[e] = k; T.plane[e] = plane; ++e; }
    return;
  }
  // What the neighbour across `side` needs of MY planes (pack), i.e. what I receive into my
  // halo from the neighbour across `side` (unpack, mirrored):
  //  low face  (side 0, pack): plane H   : c_z in {0,-1};  plane H+1     : c_z = -1
  //  high face (side 1, pack): plane H+nzl-1: c_z in {0,+1}; plane H+nzl-2: c_z = +1
  //  unpack side 0 (from below): plane H-1: c_z in {0,+1};  plane H-2: c_z = +1
  //  unpack side 1 (from above): plane H+nzl: c_z in {0,-1}; plane H+nzl+1: c_z = -1
  int near, far, dir;
  if (pack) { dir = (side == 0) ? -1 : +1; near = (side == 0) ? H : H + nzl - 1; far = (side == 0) ? H + 1 : H + nzl - 2; }
  else      { dir = (side == 0) ? +1 : -1; near = (side == 0) ? H - 1 : H + nzl; far = (side == 0) ? H - 2 : H + nzl + 1; }
  for (int fl = 0; fl < 2; ++fl) {
    for (int i = 0; i < Q; ++i) if (kCZh[i] == 0 || kCZh[i] == dir) { T.comp[e] = fl * Q + i; T.plane[e] = near; ++e; }
    for (int i = 0; i < Q; ++i) if (kCZh[i] == dir) { T.comp[e] = fl * Q + i; T.plane[e] = far; ++e; }
  }
}

double* halo_buffer(bflbm_ctx* c, int kind) {
  return (kind == BFLBM_HALO_STATE) ? c->S[c->cur] : c->S[1 - c->cur];
}

}  // namespace

extern "C" {

void bflbm_default_params(bflbm_params* p) {
  p->tau_f = 1. / 2.; p->tau_g = 1. / 2.;
  p->alpha0 = 4.; p->alpha1 = 0.;
  p->kappa = 4;
  p->kBT = 0.;
  p->cs2 = 1. / 3.;
  p->rho_lo = 0.; p->rho_hi = 1.0;
  p->seed = 12345ULL;
}

int bflbm_abi_version(void) { return BFLBM_ABI_VERSION; }
const char* bflbm_last_error(void) { return g_err.c_str(); }

int bflbm_device_count(int* n) {
  if (!n) return fail("null argument");
  HIP_TRY(hipGetDeviceCount(n));
  return 0;
}

int bflbm_create(const bflbm_params* p, const bflbm_domain* d, bflbm_ctx** out) {
  if (!p || !d || !out) return fail("bflbm_create: null argument");
  if (d->n[0] < 1 || d->n[1] < 1 || d->n[2] < 1) return fail("bflbm_create: lattice size must be >= 1");
  if (d->nranks < 1 || d->rank < 0 || d->rank >= d->nranks) return fail("bflbm_create: bad rank/nranks");
  if (d->z0 < 0 || d->z1 > d->n[2] || d->z1 <= d->z0) return fail("bflbm_create: bad slab [%d,%d) of nz=%d", d->z0, d->z1, d->n[2]);
  if (d->nranks == 1 && (d->z0 != 0 || d->z1 != d->n[2])) return fail("bflbm_create: a single slab must cover all of z");
  if (d->nranks > 1 && d->z1 - d->z0 < 4) return fail("bflbm_create: a slab needs at least 4 planes when nranks > 1");
  if ((long long)(d->n[0] + 15) * d->n[1] * (long long)(d->z1 - d->z0 + 4) >= (1LL << 31)) return fail("bflbm_create: slab too large for 32-bit site offsets");
  if ((long long)(d->n[0] + 15) * d->n[1] >= (1LL << 28)) return fail("bflbm_create: a plane must stay below 2 GB (32-bit byte offsets inside a plane)");
  HIP_TRY(hipSetDevice(d->device));
  bflbm_ctx* c = new bflbm_ctx();
  c->prm = *p; c->dom = *d;
  derive(c->prm, c->dp);
  c->nzl = d->z1 - d->z0;
  Geo& G = c->G;
  G.nx = d->n[0]; G.ny = d->n[1]; G.nz = d->n[2];
  G.zwrap = (d->nranks == 1) ? 1 : 0;
  G.H = G.zwrap ? 0 : 2;
  G.nzs = c->nzl + 2 * G.H;
  G.z0 = d->z0;
  // rows of the resident arrays start on 128-byte lines for any nx.  Measured against dense rows on one box
  // (A/B of the two libraries, +-0.5 %): 250^3 +1.5 %, 300^3 +3.9 %, 257^3 +2.7 %, no change when nx is a
  // multiple of 16 (pitch == nx, same code path).  Most of what such sizes lose against 256^3 is partial
  // tiles, not alignment.  Dense layouts (dplane) are kept for everything that crosses the ABI.
  G.pitch = (G.nx > 16) ? ((G.nx + 15) & ~15) : G.nx;
  {                                              // tuning override: row pitch in doubles (a multiple of 16, >= nx); see NOTES.md "Round 4"
    static const int pitch_env = [] { const char* e = getenv("BFLBM_PITCH"); return e ? atoi(e) : 0; }();
    if (pitch_env > 0) {
      if (pitch_env < G.nx || (pitch_env & 15)) { delete c; return fail("BFLBM_PITCH must be a multiple of 16 and at least nx"); }
      G.pitch = pitch_env;
    }
  }
  G.plane = (long long)G.pitch * G.ny;
  G.dplane = (long long)G.nx * G.ny;
  // component stride: padded so that the 38 component arrays of a power-of-two lattice do not all start
  // on the same memory channel / L2 set.  Measured on MI355X (256^3): no pad -> 65 lines (1040 doubles):
  // pull-copy 1.96 -> 1.77 ms, fused step 2.57 -> 2.35 ms; a scan with reproducible placement (one
  // allocation, below) gives 0: 6780, 1040: 6870-6940, 65552: 6935-7000, 262160: 6970, 1048592: 6730 MLUPS
  // -> 65552 doubles = 512 KB + one line (BFLBM_PAD = doubles, tuning override).
  static const long long pad = [] { const char* e = getenv("BFLBM_PAD"); return e ? atoll(e) : 1040LL; }();
  if (pad < 0 || (pad & 1)) { delete c; return fail("BFLBM_PAD must be a non-negative even number of doubles"); }
  G.vol = G.plane * G.nzs + pad;
  const size_t sbytes = (size_t)2 * Q * G.vol * sizeof(double);
  const size_t fbytes = (size_t)G.vol * sizeof(double);
  hipError_t e = hipSuccess;
  // Both state buffers come from ONE allocation with buffer B displaced by 33280 doubles (260 KB) from the
  // end of A.  With two separate allocations the relative placement of the read and the write stream of a
  // component varied from process to process and with it the step time (6510-7190 MLUPS at 256^3 on one
  // box); inside one allocation it is reproducible to 0.3 %, 520 doubles is a bad displacement (-4 %),
  // anything from 25k to 1M doubles is equally good (BFLBM_AB_OFF = doubles, tuning override).
  static const long long ab_off = [] { const char* e = getenv("BFLBM_AB_OFF"); return e ? atoll(e) : 33280LL; }();
  if (ab_off < 0 || (ab_off & 1)) { delete c; return fail("BFLBM_AB_OFF must be a non-negative even number of doubles"); }
  e = hipMalloc((void**)&c->S[0], 2 * sbytes + (size_t)ab_off * sizeof(double));
  if (e == hipSuccess) c->S[1] = c->S[0] + (size_t)2 * Q * G.vol + ab_off;
  if (e == hipSuccess) e = hipMalloc((void**)&c->rho, fbytes);
  if (e == hipSuccess) e = hipMalloc((void**)&c->phi, fbytes);
  c->partial_n = (size_t)((G.plane + 255) / 256) * (size_t)c->nzl;
  if (e == hipSuccess) e = hipMalloc((void**)&c->partial, c->partial_n * 5 * sizeof(double));
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&c->ev0);
  if (e == hipSuccess) e = hipEventCreate(&c->ev1);
  if (e != hipSuccess) {
    fail("bflbm_create: %s", hipGetErrorString(e));
    bflbm_destroy(c);
    return 1;
  }
  c->bytes = 2 * sbytes + 2 * fbytes + c->partial_n * 5 * sizeof(double);
  // halo planes must never hold NaN garbage that a reduction could touch
  hipMemsetAsync(c->S[0], 0, sbytes, c->stream);
  { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, d->device) == hipSuccess && ncu > 0) g_fused_ncu = ncu; }
  hipMemsetAsync(c->S[1], 0, sbytes, c->stream);
  hipMemsetAsync(c->rho, 0, fbytes, c->stream);
  hipMemsetAsync(c->phi, 0, fbytes, c->stream);
  HIP_TRY(hipStreamSynchronize(c->stream));
  // draw the physical placement again where it pays (see bflbm_tune_placement): lattices of at least 128^3 sites per slab, where
  // a step is long enough to time; BFLBM_PLACEMENT_CANDIDATES=1 switches it off.  A failure here is not a failure to create.
  // Default: 4 candidates, 8 where a candidate is cheap (a state below 24 GB: 256^3 still scattered over 7950 ... 8350 MLUPS with 4,
  // profiles/r04_knobs_rescan.txt; a probe there takes 60 ms).
  static const int ncand_env = [] { const char* e = getenv("BFLBM_PLACEMENT_CANDIDATES"); return e ? std::max(1, std::min(atoi(e), 8)) : 0; }();
  const int ncand = ncand_env > 0 ? ncand_env : (2 * sbytes <= ((size_t)24 << 30) ? 8 : 4);
  if (ncand > 1 && (long long)G.nx * G.ny * c->nzl >= (1LL << 21)) (void)bflbm_tune_placement(c, ncand, nullptr, nullptr);
  *out = c;
  return 0;
}

// ---- physical placement of the state ------------------------------------------------------------------------------
// Round 4 finding (profiles/r04_level_probe.txt): the step time of a context sits on one of a few discrete levels up to 8 % apart
// (512^3: 7840 ... 8510 MLUPS), the level belongs to the ALLOCATION -- stable over every block of steps and over a re-init,
// different from one context to the next at IDENTICAL virtual addresses, alternating deterministically when a process frees
// and re-creates the context -- i.e. to the physical pages behind it, which the library cannot choose.  It can draw again:
// time a few steps of the real step kernel on an analytic state, allocate another candidate WHILE HOLDING the first (so that
// the allocator hands out other memory), time it the same way, keep the faster (only the best so far and the candidate being
// timed are ever held: twice the state at the peak).  Measured process by process on one box (profiles/r04_placement_ab.txt):
// 256^3 7580-7910 -> 7900-8110 MLUPS, 384^3 7610 -> 7905, 512^3 7940-8455 -> 8420-8490; the probe's time is the bench's time.
// (Round 3 tried this with a pull-copy on the zeroed, fresh buffers as the yardstick, which did not predict the level; the
// step kernel on an initialised state does.)
static int probe_ms(bflbm_ctx* c, float* ms) {
  if (bflbm_init_stripe(c, 0.5)) return 1;
  // a slab's faces are not exchanged here: its halo planes keep the analytic state, which changes what is computed next to the
  // faces and nothing about the time
  auto run = [&](int nsteps, float* out) -> int {
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (int s = 0; s < nsteps; ++s) if (bflbm_step_boundary(c) || bflbm_step_interior(c) || bflbm_step_finish(c)) return 1;
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(out, c->ev0, c->ev1));
    return 0;
  };
  // at least 2 warm-up and 4 timed steps, and at least ~30 ms of warm-up and ~25 ms timed, so that the first candidate of a
  // fresh context (clocks still ramping after the allocation) is not handicapped on lattices whose step takes 2 ms
  float t2 = 0.f;
  if (run(2, &t2)) return 1;
  const float step = std::max(t2 / 2.f, 1e-3f);
  const int more_warm = std::max(0, (int)std::ceil(30.f / step) - 2), timed = std::max(4, (int)std::ceil(25.f / step));
  if (more_warm > 0 && run(std::min(more_warm, 64), &t2)) return 1;
  if (run(std::min(timed, 64), ms)) return 1;
  *ms /= (float)std::min(timed, 64);
  return 0;
}

int bflbm_tune_placement(bflbm_ctx* c, int max_candidates, float* ms_per_step, int* kept) {
  if (!c) return fail("null context");
  if (c->step_open) return fail("bflbm_tune_placement inside an open step");
  if (max_candidates < 1 || max_candidates > 8) return fail("max_candidates must be 1 ... 8");
  HIP_TRY(hipSetDevice(c->dom.device));
  const Geo& G = c->G;
  const size_t sdoubles = (size_t)2 * Q * G.vol;
  const size_t state_bytes = (sdoubles + (size_t)(c->S[1] - c->S[0])) * sizeof(double);     // A, the displacement, B
  // whatever happens below, the context leaves as freshly created: zeroed buffers (halo planes must never hold garbage), nothing resident
  auto fresh = [&]() -> int {
    (void)hipStreamSynchronize(c->stream);
    const size_t bytes_now = (sdoubles + (size_t)(c->S[1] - c->S[0])) * sizeof(double);
    HIP_TRY(hipMemsetAsync(c->S[0], 0, bytes_now, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->cur = 0; c->steps = 0; c->density_valid = false; c->step_open = false; c->com_valid = false; c->total_max = -1.;
    c->ref_kind = 0; c->ref_kind_step = -1;
    for (auto& b : c->fsig) for (auto& sg : b) sg = HoSig();
    return 0;
  };
  float best_ms = 0.f;
  if (probe_ms(c, &best_ms)) { (void)fresh(); return 1; }
  if (ms_per_step) ms_per_step[0] = best_ms;
  c->tune_ms[0] = best_ms; c->tune_n = 1;
  int best = 0;
  const size_t fbytes = c->frames[0] ? 2 * handover_frame_doubles(G) * sizeof(double) : 0;
  for (int k = 1; k < max_candidates; ++k) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)free_b < (double)(state_bytes + fbytes) * 1.05) break;   // no room for a second candidate
    double* oldS[2] = { c->S[0], c->S[1] };
    double* oldF[2] = { c->frames[0], c->frames[1] };
    double* nS = nullptr; double* nF = nullptr;
    // diagnostics (tools/level_probe.py): BFLBM_TUNE_WHAT=state|frames draws only that allocation again (which of the two decides the level?)
    const char* what_env = getenv("BFLBM_TUNE_WHAT");
    const bool draw_state = !what_env || strcmp(what_env, "frames") != 0, draw_frames = fbytes && (!what_env || strcmp(what_env, "state") != 0);
    if (draw_state && hipMalloc((void**)&nS, state_bytes) != hipSuccess) { (void)hipGetLastError(); break; }
    if (draw_frames && hipMalloc((void**)&nF, fbytes) != hipSuccess) { (void)hipGetLastError(); if (nS) hipFree(nS); break; }
    if (draw_state) { c->S[0] = nS; c->S[1] = nS + (oldS[1] - oldS[0]); hipMemsetAsync(nS, 0, state_bytes, c->stream); }
    if (draw_frames) { c->frames[0] = nF; c->frames[1] = nF + handover_frame_doubles(G); }
    c->cur = 0;
    float ms = 0.f;
    const int rc = probe_ms(c, &ms);
    if (ms_per_step) ms_per_step[k] = rc ? -1.f : ms;
    if (k < 8) { c->tune_ms[k] = rc ? -1.f : ms; c->tune_n = k + 1; }
    if (!rc && ms < best_ms * 0.995f) {              // keep the new one
      if (draw_state) hipFree(oldS[0]);
      if (draw_frames && oldF[0]) hipFree(oldF[0]);
      best_ms = ms; best = k;
    } else {                                        // keep the old one
      (void)hipStreamSynchronize(c->stream);
      if (nS) hipFree(nS);
      if (nF) hipFree(nF);
      c->S[0] = oldS[0]; c->S[1] = oldS[1]; c->frames[0] = oldF[0]; c->frames[1] = oldF[1];
      if (rc) { (void)fresh(); return 1; }
    }
  }
  if (fresh()) return 1;
  if (kept) *kept = best;
  c->tune_kept = best;
  return 0;
}

int bflbm_placement_report(const bflbm_ctx* c, float ms_per_step[8], int* tried, int* kept) {
  if (!c || !ms_per_step || !tried || !kept) return fail("null argument");
  for (int k = 0; k < 8; ++k) ms_per_step[k] = c->tune_ms[k];
  *tried = c->tune_n; *kept = c->tune_kept;
  return 0;
}
int bflbm_destroy(bflbm_ctx* c) {
  if (!c) return 0;
  hipSetDevice(c->dom.device);
  if (c->stream && c->own_stream) hipStreamSynchronize(c->stream);
  if (c->S[0]) hipFree(c->S[0]);                 // S[1] lives in the same allocation
  if (c->frames[0]) hipFree(c->frames[0]);
  if (c->rho) hipFree(c->rho);
  if (c->phi) hipFree(c->phi);
  if (c->injf) hipFree(c->injf);
  if (c->injg) hipFree(c->injg);
  if (c->partial) hipFree(c->partial);
  for (int k = 0; k < 3; ++k) if (c->ref[k]) hipFree(c->ref[k]);
  if (c->ev0) hipEventDestroy(c->ev0);
  if (c->ev1) hipEventDestroy(c->ev1);
  if (c->stream && c->own_stream) hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

int bflbm_set_params(bflbm_ctx* c, const bflbm_params* p) {
  if (!c || !p) return fail("null argument");
  c->prm = *p;
  derive(c->prm, c->dp);
  return 0;
}
int bflbm_get_params(const bflbm_ctx* c, bflbm_params* p) {
  if (!c || !p) return fail("null argument");
  *p = c->prm;
  return 0;
}

int bflbm_set_stream(bflbm_ctx* c, void* s, int external) {
  if (!c) return fail("null context");
  HIP_TRY(hipSetDevice(c->dom.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (external) {
    if (c->own_stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)s;            // may be the null (legacy default) stream
    c->own_stream = false;
  } else if (!c->own_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
  }
  return 0;
}

int bflbm_resolved_schedule(const bflbm_ctx* c, int* schedule) {
  if (!c || !schedule) return fail("null argument");
  *schedule = resolved_schedule(c);
  return 0;
}

int bflbm_set_schedule(bflbm_ctx* c, int schedule) {
  if (!c) return fail("null context");
  if (schedule < 0 || schedule > 3) return fail("unknown schedule %d", schedule);
  c->schedule = schedule;
  return 0;
}

// ---- initial conditions ---------------------------------------------------------------
static int run_init(bflbm_ctx* c, int mode, const double* rho_ext_host, size_t n_ext, double rho_c, double phi_c, double rho_t) {
  HIP_TRY(hipSetDevice(c->dom.device));
  double* scratch = c->S[1 - c->cur];
  if (mode != 0) HIP_TRY(hipMemcpyAsync(scratch, rho_ext_host, n_ext * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_init, plane_grid(c, c->G.nzs), dim3(256), 0, c->stream, c->S[c->cur], scratch, c->G, mode, rho_c, phi_c, rho_t, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->steps = 0; c->density_valid = false; c->step_open = false; c->com_valid = false;
  c->total_max = -1.;                              // rho + phi = rho_hi + rho_lo at every site
  for (auto& b : c->fsig) for (auto& sg : b) sg = HoSig();     // the hand-over frames describe another state
  // the frames of schedule 3 are allocated here, where a run starts, rather than inside its first (possibly timed or
  // overlapped) step; a failure is remembered and `auto` stays bit-exact (an explicit schedule 3 reports it at the step)
  if ((c->schedule == 2 || c->schedule == 3) && resolved_schedule(c) == 3) (void)ensure_frames(c, true);
  c->ref_kind = (mode == 0) ? 2 : 1;             // thermal_noise gets the absolute COM (:623-625) or zero (:690, :739)
  c->ref_kind_step = 0;
  return 0;
}

// global z of extended plane pe (pe = storage plane + 1), wrapped
static inline int ext_global_z(const bflbm_ctx* c, int pe) {
  int gz = c->dom.z0 - c->G.H + pe - 1;
  const int nz = c->G.nz;
  gz %= nz; if (gz < 0) gz += nz;
  return gz;
}

int bflbm_init_mixture(bflbm_ctx* c) {
  if (!c) return fail("null context");
  const double C1 = 0.5, C2 = 0.5;               // LBM_binary.H:606-614
  if (run_init(c, 0, nullptr, 0, 2. * C1, 2. * C2, 0.)) return 1;
  // rho = phi = 1 at every site whatever rho_hi / rho_lo say (:613-614): the total density `auto` keys its stability bound on is 2
  // (alpha0 = 4 on this state is NaN within 50 steps on the reference's CPU path, tests/test_oracle_pins.py)
  c->total_max = 2. * C1 + 2. * C2;
  return 0;
}

int bflbm_init_stripe(bflbm_ctx* c, double frac) {
  if (!c) return fail("null context");
  const bflbm_params& P = c->prm;
  const int nz = c->G.nz;
  const double rho_t = P.rho_hi + P.rho_lo;      // LBM_binary.H:674-681
  const double pos_lo = (-0.5 * frac) * nz;
  const double pos_hi = (0.5 * frac) * nz;
  std::vector<double> tab((size_t)c->G.nzs + 2);
  for (int pe = 0; pe < c->G.nzs + 2; ++pe) {
    const int z = ext_global_z(c, pe);
    const double pos = z - nz / 2;
    tab[pe] = (P.rho_hi - P.rho_lo) * 0.5 * (std::tanh((pos - pos_lo) / std::sqrt(P.kappa)) + std::tanh((pos_hi - pos) / std::sqrt(P.kappa))) + P.rho_lo;
  }
  return run_init(c, 1, tab.data(), tab.size(), 0., 0., rho_t);
}

int bflbm_init_droplet(bflbm_ctx* c, double r_frac) {
  if (!c) return fail("null context");
  const bflbm_params P = c->prm;
  const int nx = c->G.nx, ny = c->G.ny;
  const double R = r_frac * nx;                  // LBM_binary.H:714 (box[0])
  const double rho_t = P.rho_hi + P.rho_lo;
  const int npe = c->G.nzs + 2;
  const size_t plane = (size_t)c->G.dplane;       // host table: dense planes (k_init reads it that way)
  std::vector<double> fld(plane * npe);
  auto work = [&](int pa, int pb) {
    for (int pe = pa; pe < pb; ++pe) {
      const int z = ext_global_z(c, pe);
      const double rz = z - nx / 2;              // :725 box[0], integer division (sic)
      for (int y = 0; y < ny; ++y) {
        const double ry = y - ny / 2.;
        for (int x = 0; x < nx; ++x) {
          const double rx = x - nx / 2.;
          const double r2 = rx * rx + ry * ry + rz * rz;
          const double r = std::sqrt(r2);
          fld[(size_t)pe * plane + (size_t)y * nx + x] = (P.rho_hi - P.rho_lo) * (1. + std::tanh((R - r) / std::sqrt(P.kappa))) / 2. + P.rho_lo;
        }
      }
    }
  };
  unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
  if ((size_t)npe * plane < (1u << 18)) nt = 1;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) {
    const int pa = (int)((long long)npe * t / nt), pb = (int)((long long)npe * (t + 1) / nt);
    if (pb > pa) th.emplace_back(work, pa, pb);
  }
  for (auto& t : th) t.join();
  return run_init(c, 2, fld.data(), fld.size(), 0., 0., rho_t);
}

// ---- upload / download ----------------------------------------------------------------
int bflbm_upload_fg(bflbm_ctx* c, const double* f, const double* g, const bflbm_fab* box) {
  if (!c || !f || !g) return fail("null argument");
  HIP_TRY(hipSetDevice(c->dom.device));
  double* N = c->S[1 - c->cur];
  if (copy_fab(c, const_cast<double*>(f), box, Q, N, c->G.vol, c->G.H, true)) return 1;
  if (copy_fab(c, const_cast<double*>(g), box, Q, N + (size_t)Q * c->G.vol, c->G.vol, c->G.H, true)) return 1;
  return 0;
}

int bflbm_commit_upload(bflbm_ctx* c, int reset) {
  if (!c) return fail("null context");
  HIP_TRY(hipSetDevice(c->dom.device));
  hipLaunchKernelGGL(k_unstream, plane_grid(c, c->nzl), dim3(256), 0, c->stream, c->S[1 - c->cur], c->S[c->cur], c->G, own_lo(c));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (reset) c->steps = 0;
  c->density_valid = false; c->step_open = false; c->com_valid = false;
  for (auto& b : c->fsig) for (auto& sg : b) sg = HoSig();
  c->ref_kind = 0; c->ref_kind_step = c->steps;  // LBM_init: COM relative to com_ref (:651-654)
  // what `auto` keys its stability bound on from here on: the total density the upload made resident, not rho_hi/rho_lo
  // (handover_contract_params).  The own planes' densities need no halo (the upload buffer holds the streamed populations
  // of every own site), so the slab reduces its own maximum; a ring combines the slabs' (bflbm_ring_commit_upload).
  {
    const int lo = own_lo(c);
    hipLaunchKernelGGL(k_density_streamed, plane_grid(c, c->nzl), dim3(256), 0, c->stream, c->S[1 - c->cur], c->rho, c->phi, c->G, lo);
    hipLaunchKernelGGL(k_total_absmax, plane_grid(c, c->nzl), dim3(256), 0, c->stream, c->rho, c->phi, c->partial, c->G, lo);
    HIP_TRY(hipGetLastError());
    std::vector<double> h(c->partial_n);
    HIP_TRY(hipMemcpyAsync(h.data(), c->partial, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    double m = 0.;
    for (double v : h) m = (v != v || m != m) ? (m + v) : std::max(m, v);
    c->total_max = (m == m) ? m : std::numeric_limits<double>::infinity();     // NaN in the upload: never schedule 3
  }
  if ((c->schedule == 2 || c->schedule == 3) && resolved_schedule(c) == 3) (void)ensure_frames(c, true);   // as after the analytic inits
  return 0;
}

// the total-density bound of `auto` (see handover_contract_params): < 0 = from the parameters
int bflbm_state_total_max(const bflbm_ctx* c, double* total_max) {
  if (!c || !total_max) return fail("null argument");
  *total_max = c->total_max;
  return 0;
}
int bflbm_set_state_total_max(bflbm_ctx* c, double total_max) {      // a driver that owns several slabs hands every slab the global maximum
  if (!c) return fail("null context");
  c->total_max = total_max;
  return 0;
}

int bflbm_download_fg(bflbm_ctx* c, double* f, double* g, const bflbm_fab* box) {
  if (!c || !f || !g) return fail("null argument");
  if (c->step_open) return fail("download inside an open step");
  HIP_TRY(hipSetDevice(c->dom.device));
  double* N = c->S[1 - c->cur];
  hipLaunchKernelGGL(k_pull, plane_grid(c, c->nzl), dim3(256), 0, c->stream, c->S[c->cur], N, c->G, own_lo(c));
  HIP_TRY(hipGetLastError());
  if (copy_fab(c, f, box, Q, N, c->G.vol, c->G.H, false)) return 1;
  if (copy_fab(c, g, box, Q, N + (size_t)Q * c->G.vol, c->G.vol, c->G.H, false)) return 1;
  return 0;
}

// ---- time stepping --------------------------------------------------------------------
static int prepare_ref(bflbm_ctx* c);
int bflbm_step_boundary(bflbm_ctx* c) {
  if (!c) return fail("null context");
  if (c->step_open) return fail("step already open");
  HIP_TRY(hipSetDevice(c->dom.device));
  if (prepare_ref(c)) return 1;
  if (c->schedule == 2 && resolved_schedule(c) == 3) (void)ensure_frames(c, true);   // auto: falls back when they do not fit
  // an explicit schedule 3 whose frames cannot be allocated fails HERE, before the step is opened: nothing of the resident
  // state has been touched and the caller may switch to an exact schedule and go on
  if (c->schedule == 3 && resolved_schedule(c) == 3 && ensure_frames(c)) return 1;
  c->step_open = true;
  const int lo = own_lo(c), hi = own_hi(c);
  if (c->G.zwrap) return 0;                      // single slab: everything is "interior"
  const int sch = resolved_schedule(c);
  int rc;
  if (sch == 3) {
    if (hi - lo > 4) rc = launch_handover(c, lo, hi, 2);
    else rc = launch_handover(c, lo, lo + 2, 2) || launch_handover(c, hi - 2, hi, 2);
  } else if (sch == 1) {
    if (hi - lo > 4) rc = launch_fused(c, lo, hi, 2);        // both boundary plane pairs in one launch
    else rc = launch_fused(c, lo, lo + 2) || launch_fused(c, hi - 2, hi);
  } else {
    rc = ensure_density(c) || launch_collide(c, lo, lo + 2) || launch_collide(c, hi - 2, hi);
  }
  if (rc) c->step_open = false;                  // a failed launch wrote nothing that the resident state S[cur] holds
  return rc;
}

int bflbm_step_interior(bflbm_ctx* c) {
  if (!c) return fail("null context");
  if (!c->step_open) return fail("bflbm_step_interior: call bflbm_step_boundary first");
  HIP_TRY(hipSetDevice(c->dom.device));
  const int lo = own_lo(c), hi = own_hi(c);
  const int a = c->G.zwrap ? lo : lo + 2, b = c->G.zwrap ? hi : hi - 2;
  const int sch = resolved_schedule(c);
  int rc;
  if (sch == 3) rc = launch_handover(c, a, b);
  else if (sch == 1) rc = launch_fused(c, a, b);
  else rc = ensure_density(c) || launch_collide(c, a, b);
  if (rc) c->step_open = false;                  // as in bflbm_step_boundary: S[cur] is intact, the step may be retried
  return rc;
}

int bflbm_step_finish(bflbm_ctx* c) {
  if (!c) return fail("null context");
  if (!c->step_open) return fail("no open step");
  c->cur = 1 - c->cur;
  c->steps += 1;
  c->step_open = false;
  c->density_valid = false;
  c->com_valid = false;
  if (c->inject) c->inject = false;              // injected noise feeds exactly one step
  return 0;
}

int bflbm_step(bflbm_ctx* c, int nsteps) {
  if (!c) return fail("null context");
  if (nsteps < 0) return fail("nsteps < 0");
  if (!c->G.zwrap && nsteps > 1) return fail("bflbm_step: nranks > 1 needs a halo exchange between steps; use nsteps == 1");
  for (int s = 0; s < nsteps; ++s) {
    if (bflbm_step_boundary(c)) return 1;
    if (bflbm_step_interior(c)) return 1;
    if (bflbm_step_finish(c)) return 1;
  }
  return 0;
}

#ifdef BFLBM_STAMP
// diagnostic build only: the phase stamps of the hand-over kernel (tools/ho_stamps.py)
int bflbm_debug_ho_stamps(unsigned long long* out, int n) {
  if (!out || n > 4 * HO_STAMP_POS * HO_NSTAMP) return fail("bad argument");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ho_stamps), (size_t)n * sizeof(unsigned long long)));
  return 0;
}
#endif

int bflbm_step_count(const bflbm_ctx* c, long long* n) {
  if (!c || !n) return fail("null argument");
  *n = c->steps;
  return 0;
}

// The step counter is the noise index of the counter-based generator.  A restart from a kBT > 0 checkpoint sets it
// to the checkpoint's absolute step (main_run_job.cpp:80 step_continue) so that the continued run draws fresh
// normals instead of replaying the first segment's.
int bflbm_set_step_count(bflbm_ctx* c, long long n) {
  if (!c) return fail("null context");
  if (n < 0) return fail("bflbm_set_step_count: negative step count");
  if (c->step_open) return fail("bflbm_set_step_count inside an open step");
  if (c->ref_kind_step == c->steps) c->ref_kind_step = n;
  c->steps = n;
  for (auto& b : c->fsig) for (auto& sg : b) sg = HoSig();     // frames are keyed by the step that wrote them
  return 0;
}

// ---- halo exchange support -------------------------------------------------------------
int bflbm_halo_bytes(const bflbm_ctx* c, int kind, size_t* bytes) {
  if (!c || !bytes) return fail("null argument");
  if (kind < 0 || kind > 2) return fail("unknown halo kind %d", kind);
  *bytes = (size_t)2 * Q * (size_t)c->G.plane * sizeof(double);
  return 0;
}

int bflbm_halo_pack(bflbm_ctx* c, int kind, int side, void* buf) {
  if (!c || !buf) return fail("null argument");
  if (c->G.zwrap) return fail("halo exchange on a single slab");
  if (kind < 0 || kind > 2 || side < 0 || side > 1) return fail("bad halo kind/side");
  HIP_TRY(hipSetDevice(c->dom.device));
  HaloTable T; halo_table(c, kind, side, true, T);
  hipLaunchKernelGGL(k_halo_pack, plane_grid(c, 2 * Q), dim3(256), 0, c->stream, halo_buffer(c, kind), (double*)buf, c->G, T);
  HIP_TRY(hipGetLastError());
  return 0;
}

int bflbm_halo_unpack(bflbm_ctx* c, int kind, int side, const void* buf) {
  if (!c || !buf) return fail("null argument");
  if (c->G.zwrap) return fail("halo exchange on a single slab");
  if (kind < 0 || kind > 2 || side < 0 || side > 1) return fail("bad halo kind/side");
  HIP_TRY(hipSetDevice(c->dom.device));
  HaloTable T; halo_table(c, kind, side, false, T);
  hipLaunchKernelGGL(k_halo_unpack, plane_grid(c, 2 * Q), dim3(256), 0, c->stream, halo_buffer(c, kind), (const double*)buf, c->G, T);
  HIP_TRY(hipGetLastError());
  if (kind == BFLBM_HALO_STATE) c->density_valid = false;
  return 0;
}

// Staging-free exchange for a driver that owns the transport (RCCL P2P in slab.SlabLattice): every entry of the halo
// table is ONE contiguous component plane of the state buffer, so a sender can post it straight from where the boundary
// kernels wrote it and a receiver straight into the halo plane the next step pulls from.
int bflbm_halo_planes(bflbm_ctx* c, int kind, int side, int pack, void** planes, size_t* plane_bytes, int* count) {
  if (!c || !planes || !plane_bytes || !count) return fail("null argument");
  if (c->G.zwrap) return fail("halo exchange on a single slab");
  if (kind < 0 || kind > 2 || side < 0 || side > 1) return fail("bad halo kind/side");
  HaloTable T; halo_table(c, kind, side, pack != 0, T);
  double* base = halo_buffer(c, kind);
  for (int e = 0; e < 2 * Q; ++e) planes[e] = base + (size_t)T.comp[e] * (size_t)c->G.vol + (size_t)T.plane[e] * (size_t)c->G.plane;
  *plane_bytes = (size_t)c->G.plane * sizeof(double);
  *count = 2 * Q;
  if (!pack && kind == BFLBM_HALO_STATE) c->density_valid = false;    // the caller is about to overwrite halo planes of the resident state
  return 0;
}

// ---- observables -----------------------------------------------------------------------
static int observe(bflbm_ctx* c, int what, int ncomp_out, double* dst, double* dst2, int ncomp_host, const bflbm_fab* box) {
  if (c->step_open) return fail("observables requested inside an open step");
  HIP_TRY(hipSetDevice(c->dom.device));
  if (what == 2 && ensure_density(c)) return 1;
  if (what != 0 && prepare_ref(c)) return 1;
  const RefState Rf = ref_state(c);
  double* out = c->S[1 - c->cur];
  dim3 g = plane_grid(c, c->nzl), b(256);
  const uint32_t idx = (uint32_t)c->steps;
  const int inj = c->inject ? 1 : 0;
  if (what == 0) hipLaunchKernelGGL((k_observe<0>), g, b, 0, c->stream, c->S[c->cur], c->rho, c->phi, c->injf, c->injg, out, c->G, c->dp, own_lo(c), idx, ncomp_out, inj, Rf);
  if (what == 1) hipLaunchKernelGGL((k_observe<1>), g, b, 0, c->stream, c->S[c->cur], c->rho, c->phi, c->injf, c->injg, out, c->G, c->dp, own_lo(c), idx, ncomp_out, inj, Rf);
  if (what == 2) hipLaunchKernelGGL((k_observe<2>), g, b, 0, c->stream, c->S[c->cur], c->rho, c->phi, c->injf, c->injg, out, c->G, c->dp, own_lo(c), idx, ncomp_out, inj, Rf);
  HIP_TRY(hipGetLastError());
  const long long ovol = (long long)c->nzl * c->G.dplane;
  if (what == 1) {
    if (dst && copy_fab(c, dst, box, Q, out, ovol, 0, false, false, true)) return 1;
    if (dst2 && copy_fab(c, dst2, box, Q, out + (size_t)Q * ovol, ovol, 0, false, false, true)) return 1;
    return 0;
  }
  return copy_fab(c, dst, box, ncomp_host, out, ovol, 0, false, false, true);
}

int bflbm_get_hydrovsbar(bflbm_ctx* c, double* dst, int ncomp, const bflbm_fab* box) {
  if (!c || !dst) return fail("null argument");
  if (ncomp < 1) return fail("ncomp < 1");
  return observe(c, 0, BFLBM_NHYDROBAR, dst, nullptr, std::min(ncomp, BFLBM_NHYDROBAR), box);
}
int bflbm_get_hydrovs(bflbm_ctx* c, double* dst, int ncomp, const bflbm_fab* box) {
  if (!c || !dst) return fail("null argument");
  if (ncomp < 1) return fail("ncomp < 1");
  const int n = std::min(ncomp, BFLBM_NHYDRO);
  return observe(c, 2, n, dst, nullptr, n, box);
}
int bflbm_get_noise(bflbm_ctx* c, double* fn, double* gn, const bflbm_fab* box) {
  if (!c) return fail("null context");
  return observe(c, 1, 2 * Q, fn, gn, Q, box);
}

int bflbm_inject_noise(bflbm_ctx* c, const double* fn, const double* gn, const bflbm_fab* box) {
  if (!c) return fail("null context");
  if (!fn || !gn) { c->inject = false; return 0; }
  HIP_TRY(hipSetDevice(c->dom.device));
  const size_t nb = (size_t)Q * c->nzl * (size_t)c->G.dplane * sizeof(double);
  if (!c->injf) { HIP_TRY(hipMalloc((void**)&c->injf, nb)); HIP_TRY(hipMalloc((void**)&c->injg, nb)); c->bytes += 2 * nb; }
  const long long ovol = (long long)c->nzl * c->G.dplane;
  if (copy_fab(c, const_cast<double*>(fn), box, Q, c->injf, ovol, 0, true, false, true)) return 1;
  if (copy_fab(c, const_cast<double*>(gn), box, Q, c->injg, ovol, 0, true, false, true)) return 1;
  c->inject = true;
  return 0;
}

static int reduce5(bflbm_ctx* c, double out[5]) {
  if (c->step_open) return fail("reduction requested inside an open step");
  HIP_TRY(hipSetDevice(c->dom.device));
  if (ensure_density(c)) return 1;
  hipLaunchKernelGGL(k_reduce, plane_grid(c, c->nzl), dim3(256), 0, c->stream, c->rho, c->phi, c->partial, c->G, own_lo(c));
  HIP_TRY(hipGetLastError());
  std::vector<double> h(c->partial_n * 5);
  HIP_TRY(hipMemcpyAsync(h.data(), c->partial, h.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int k = 0; k < 5; ++k) out[k] = 0.;
  for (size_t b = 0; b < c->partial_n; ++b) for (int k = 0; k < 5; ++k) out[k] += h[b * 5 + k];
  return 0;
}

int bflbm_com_sums(bflbm_ctx* c, double sums[4]) {
  if (!c || !sums) return fail("null argument");
  double r[5];
  if (reduce5(c, r)) return 1;
  sums[0] = r[0]; sums[1] = r[2]; sums[2] = r[3]; sums[3] = r[4];
  return 0;
}

int bflbm_mass(bflbm_ctx* c, double* rho_sum, double* phi_sum) {
  if (!c || !rho_sum || !phi_sum) return fail("null argument");
  double r[5];
  if (reduce5(c, r)) return 1;
  *rho_sum = r[0]; *phi_sum = r[1];
  return 0;
}

// ---- reference-state noise (USE_REF_STATE) ------------------------------------------------------
// the centre of mass the noise of the resident state needs (LBM_binary.H:585-588); a single slab
// reduces it itself, a slab of a decomposed lattice gets the global one from its driver
static int prepare_ref(bflbm_ctx* c) {
  if (!ref_active(c)) return 0;
  const int kind = (c->steps == c->ref_kind_step) ? c->ref_kind : 0;
  if (kind == 1 || c->com_valid) return 0;
  if (!c->G.zwrap) return fail("reference-state noise on a slab: hand over the global centre of mass of the resident state first (bflbm_set_com)");
  double r[5];
  if (reduce5(c, r)) return 1;
  for (int d = 0; d < 3; ++d) c->com[d] = r[2 + d] / r[0];
  for (int d = 0; d < 3; ++d)
    if (!std::isfinite(c->com[d])) return fail("reference-state noise: the centre of mass of fluid f is not finite (total mass %.3g)", r[0]);
  c->com_valid = true;
  return 0;
}

int bflbm_set_ref_state(bflbm_ctx* c, const double* rho_eq, const double* phi_eq, const double* rhot_eq, const bflbm_fab* box) {
  if (!c || !rho_eq || !phi_eq || !rhot_eq) return fail("null argument");
  if (check_fab(box)) return 1;
  HIP_TRY(hipSetDevice(c->dom.device));
  const size_t nb = (size_t)c->G.dplane * c->G.nz * sizeof(double);
  const double* src[3] = { rho_eq, phi_eq, rhot_eq };
  // the whole lattice on every slab: the lookup is shifted by the drifting centre of mass
  for (int k = 0; k < 3; ++k) {
    if (!c->ref[k]) { HIP_TRY(hipMalloc((void**)&c->ref[k], nb)); HIP_TRY(hipMemsetAsync(c->ref[k], 0, nb, c->stream)); c->bytes += nb; }
    if (copy_fab(c, const_cast<double*>(src[k]), box, 1, c->ref[k], 0, 0, true, true, true)) return 1;
  }
  return 0;
}

int bflbm_enable_ref_state(bflbm_ctx* c, int on, const double com_ref[3]) {
  if (!c) return fail("null context");
  if (on && (!c->ref[0] || !com_ref)) return fail("bflbm_enable_ref_state: upload the reference state first (bflbm_set_ref_state) and give com_ref");
  c->ref_on = on != 0;
  if (on) for (int d = 0; d < 3; ++d) c->com_ref[d] = com_ref[d];
  return 0;
}

int bflbm_ref_state_active(const bflbm_ctx* c, int* active) {
  if (!c || !active) return fail("null argument");
  *active = ref_active(c) ? 1 : 0;
  return 0;
}

int bflbm_set_com(bflbm_ctx* c, const double com[3]) {
  if (!c || !com) return fail("null argument");
  if (c->step_open) return fail("bflbm_set_com inside an open step");
  for (int d = 0; d < 3; ++d) if (!std::isfinite(com[d])) return fail("bflbm_set_com: centre of mass is not finite");
  for (int d = 0; d < 3; ++d) c->com[d] = com[d];
  c->com_valid = true;
  return 0;
}

int bflbm_sync(bflbm_ctx* c) {
  if (!c) return fail("null context");
  HIP_TRY(hipSetDevice(c->dom.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int bflbm_timer_start(bflbm_ctx* c) {
  if (!c) return fail("null context");
  HIP_TRY(hipEventRecord(c->ev0, c->stream));
  return 0;
}
int bflbm_timer_stop(bflbm_ctx* c, float* ms) {
  if (!c || !ms) return fail("null argument");
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return 0;
}

int bflbm_rng_site_normals(uint64_t seed, uint64_t site, uint32_t noise_index, double* out36) {
  if (!out36) return fail("null argument");
  static const double tab[BFLBM_NORMAL_TABLE_N] = BFLBM_NORMAL_TABLE_VALUES;
  bflbm_rng_state st;
  bflbm_rng_seed((uint32_t)seed, (uint32_t)(seed >> 32), site, noise_index, st);
  for (int k = 0; k < 36; ++k) out36[k] = (k < 33) ? bflbm_normal_from_bits(bflbm_rng_next(st), tab) : 0.;
  return 0;
}

int bflbm_debug_time_kernel(bflbm_ctx* c, int which, int reps, float* ms) {
  if (!c || !ms || reps < 1) return fail("bad argument");
  if (c->step_open) return fail("diagnostics inside an open step");
  HIP_TRY(hipSetDevice(c->dom.device));
  const size_t sbytes = (size_t)2 * Q * c->G.vol * sizeof(double);
  for (int r = -1; r < reps; ++r) {
    if (r == 0) HIP_TRY(hipEventRecord(c->ev0, c->stream));
    if (which == 0) hipLaunchKernelGGL(k_pull, plane_grid(c, c->nzl), dim3(256), 0, c->stream, c->S[c->cur], c->S[1 - c->cur], c->G, own_lo(c));
    else if (which == 1) hipLaunchKernelGGL(k_density, plane_grid(c, c->nzl), dim3(256), 0, c->stream, c->S[c->cur], c->rho, c->phi, c->G, own_lo(c));
#ifdef BFLBM_CALIBRATION
    else if (which == 3 && c->G.pitch == c->G.nx) hipLaunchKernelGGL(k_pull2, dim3((unsigned)((c->G.plane / 2 + 255) / 256), (unsigned)c->nzl), dim3(256), 0, c->stream, c->S[c->cur], c->S[1 - c->cur], c->G, own_lo(c));
    else if (which == 4 && c->G.zwrap && c->G.pitch == c->G.nx) hipLaunchKernelGGL(k_pull_rows, plane_grid(c, c->nzl), dim3(256), 0, c->stream, c->S[c->cur], c->S[1 - c->cur], c->G, own_lo(c));
#endif
    else if (which == 2) HIP_TRY(hipMemcpyAsync(c->S[1 - c->cur], c->S[c->cur], sbytes, hipMemcpyDeviceToDevice, c->stream));
    else return fail("unknown diagnostic kernel %d", which);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev1, c->stream));
  HIP_TRY(hipEventSynchronize(c->ev1));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, c->ev0, c->ev1));
  *ms = t / reps;
  c->density_valid = false;
  return 0;
}


// diagnostics (tools/level_probe.py): device addresses of the state buffers A and B, the density arrays, the reduction scratch
// and the two frame buffers (0 when not allocated)
int bflbm_debug_addresses(const bflbm_ctx* c, unsigned long long out[8]) {
  if (!c || !out) return fail("null argument");
  out[0] = (unsigned long long)c->S[0]; out[1] = (unsigned long long)c->S[1]; out[2] = (unsigned long long)c->rho; out[3] = (unsigned long long)c->phi;
  out[4] = (unsigned long long)c->partial; out[5] = (unsigned long long)c->frames[0]; out[6] = (unsigned long long)c->frames[1]; out[7] = (unsigned long long)c->G.vol;
  return 0;
}

int bflbm_device_bytes(const bflbm_ctx* c, size_t* bytes) {
  if (!c || !bytes) return fail("null argument");
  *bytes = c->bytes;
  return 0;
}


// ---- single-process ring of slabs -----------------------------------------------------------
struct bflbm_ring {
  std::vector<bflbm_ctx*> ctx;
  std::vector<hipStream_t> comm;        // per slab: copies + unpack, concurrent with the interior kernel
  std::vector<hipEvent_t> packed;       // per slab: the planes its neighbours copy are final (recorded on the slab's main stream)
  std::vector<hipEvent_t> unpacked;     // per slab: both halo faces stored (recorded on the comm stream)
  std::vector<std::array<bool, 2>> peer_ok;   // per slab: kernels on its device may read the lower / upper neighbour's memory
  size_t bytes = 0;
  bool overlap = true;                  // false: the faces move after the interior sweep (measures what the overlap buys)
  int transport = 0;                    // 0: gather kernel reading the neighbour in place where reachable; 1: copy engine, one copy per plane
  int last_kernel_faces = 0, last_copy_faces = 0;   // what the last exchange used
};

// Halo exchange of the ring without staging: every (component, plane) entry of the halo table is one
// contiguous plane (pitch*ny doubles) in the source slab and in the destination slab, so each face is moved by
// 38 direct copies  neighbour's state -> my halo planes  on my comm stream (SDMA / xGMI between GPUs, a device
// copy on one GPU): 2 x plane bytes of traffic per entry instead of the 6 x of pack -> copy -> unpack.
static int ring_mark(bflbm_ring* r) {
  const int n = (int)r->ctx.size();
  if (n == 1) return 0;
  for (int k = 0; k < n; ++k) {
    bflbm_ctx* c = r->ctx[k];
    HIP_TRY(hipSetDevice(c->dom.device));
    HIP_TRY(hipEventRecord(r->packed[k], c->stream));            // my face planes are final from here on
  }
  return 0;
}
static int ring_copy(bflbm_ring* r, int kind) {
  const int n = (int)r->ctx.size();
  if (n == 1) return 0;
  r->last_kernel_faces = r->last_copy_faces = 0;
  for (int k = 0; k < n; ++k) {
    bflbm_ctx* c = r->ctx[k];
    const int lower = (k + n - 1) % n, upper = (k + 1) % n;
    HIP_TRY(hipSetDevice(c->dom.device));
    HIP_TRY(hipStreamWaitEvent(r->comm[k], r->packed[lower], 0));
    HIP_TRY(hipStreamWaitEvent(r->comm[k], r->packed[upper], 0));
    HIP_TRY(hipStreamWaitEvent(r->comm[k], r->packed[k], 0));     // my own halo planes must not be overwritten earlier
    const size_t pbytes = (size_t)c->G.plane * sizeof(double);
    for (int side = 0; side < 2; ++side) {
      // my low halo (side 0) <- lower neighbour's high face (its pack side 1); my high halo <- upper neighbour's low face
      bflbm_ctx* src = r->ctx[side == 0 ? lower : upper];
      HaloTable Tu, Tp;
      halo_table(c, kind, side, false, Tu);
      halo_table(src, kind, 1 - side, true, Tp);
      double* dst_base = halo_buffer(c, kind);
      const double* src_base = halo_buffer(src, kind);
      for (int e = 0; e < 2 * Q; ++e) if (Tu.comp[e] != Tp.comp[e]) return fail("halo tables of neighbouring slabs disagree");
      // BFLBM_RING_COPY_FALLBACK=1 forces the per-plane copies (what a pair of GPUs without peer mapping gets), so that
      // both branches are exercised on a one-GPU box (tests/test_gpu_slabs.py)
      static const bool force_copies = [] { const char* e = getenv("BFLBM_RING_COPY_FALLBACK"); return e && atoi(e) != 0; }();
      const bool reachable = !force_copies && r->transport == 0 && ((src->dom.device == c->dom.device) || r->peer_ok[k][side]);
      if (reachable && (c->G.plane % 2 == 0) && (c->G.vol % 2 == 0) && (src->G.vol % 2 == 0)) {
        ++r->last_kernel_faces;
        // one gather kernel per face reads the neighbour's planes in place (peer memory over xGMI between GPUs)
        dim3 g((unsigned)((c->G.plane / 2 + 255) / 256), (unsigned)(2 * Q));
        hipLaunchKernelGGL(k_halo_pull, g, dim3(256), 0, r->comm[k], dst_base, src_base, c->G.plane, c->G.vol, src->G.vol, Tu, Tp);
        HIP_TRY(hipGetLastError());
      } else {
        ++r->last_copy_faces;
        for (int e = 0; e < 2 * Q; ++e) {
          double* d = dst_base + (size_t)Tu.comp[e] * c->G.vol + (size_t)Tu.plane[e] * c->G.plane;
          const double* sp = src_base + (size_t)Tp.comp[e] * src->G.vol + (size_t)Tp.plane[e] * src->G.plane;
          HIP_TRY(hipMemcpyPeerAsync(d, c->dom.device, sp, src->dom.device, pbytes, r->comm[k]));
        }
      }
    }
    HIP_TRY(hipEventRecord(r->unpacked[k], r->comm[k]));
    if (kind == BFLBM_HALO_STATE) c->density_valid = false;
  }
  return 0;
}
static int ring_exchange(bflbm_ring* r, int kind) { return ring_mark(r) || ring_copy(r, kind); }

// everything later on a slab's main stream sees the stored halos
static int ring_join(bflbm_ring* r) {
  const int n = (int)r->ctx.size();
  if (n == 1) return 0;
  for (int k = 0; k < n; ++k) {
    HIP_TRY(hipSetDevice(r->ctx[k]->dom.device));
    HIP_TRY(hipStreamWaitEvent(r->ctx[k]->stream, r->unpacked[k], 0));
    // a neighbour must not overwrite the planes I copy from before my copies of them are done
    HIP_TRY(hipStreamWaitEvent(r->ctx[k]->stream, r->unpacked[(k + 1) % n], 0));
    HIP_TRY(hipStreamWaitEvent(r->ctx[k]->stream, r->unpacked[(k + n - 1) % n], 0));
  }
  return 0;
}

int bflbm_ring_create(const bflbm_params* p, const int n[3], int nslabs, const int* devices, int ndevices, bflbm_ring** out) {
  if (!p || !n || !out || nslabs < 1) return fail("bflbm_ring_create: bad argument");
  if (nslabs > 1 && n[2] / nslabs < 4) return fail("bflbm_ring_create: every slab needs at least 4 planes");
  bflbm_ring* r = new bflbm_ring();
  for (int k = 0; k < nslabs; ++k) {
    bflbm_domain d;
    for (int a = 0; a < 3; ++a) d.n[a] = n[a];
    d.z0 = (int)((long long)n[2] * k / nslabs); d.z1 = (int)((long long)n[2] * (k + 1) / nslabs);
    d.rank = k; d.nranks = nslabs;
    d.device = (devices && ndevices > 0) ? devices[k % ndevices] : 0;
    bflbm_ctx* c = nullptr;
    if (bflbm_create(p, &d, &c)) { bflbm_ring_destroy(r); return 1; }
    r->ctx.push_back(c);
  }
  if (nslabs > 1) {
    size_t b = 0;
    bflbm_halo_bytes(r->ctx[0], BFLBM_HALO_STATE, &b);
    r->bytes = b;
    hipError_t e = hipSuccess;
    for (int k = 0; k < nslabs && e == hipSuccess; ++k) {
      e = hipSetDevice(r->ctx[k]->dom.device);
      hipStream_t st = nullptr; hipEvent_t e1 = nullptr, e2 = nullptr;
      // the comm stream at the device's highest priority: its gather kernels (or copies) should take the first CU a finishing
      // workgroup of the 512-register interior sweep frees, not queue behind the sweep's next round
      int prio_lo = 0, prio_hi = 0;
      if (e == hipSuccess && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) { (void)hipGetLastError(); prio_hi = 0; }
      if (e == hipSuccess) e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_hi);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&e1, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&e2, hipEventDisableTiming);
      r->comm.push_back(st); r->packed.push_back(e1); r->unpacked.push_back(e2);
      // peer mapping of the ring neighbours' state buffers, which k_halo_pull reads in place (nothing to do when they share the device)
      std::array<bool, 2> ok = {false, false};
      const int nbs[2] = { (k + nslabs - 1) % nslabs, (k + 1) % nslabs };       // lower, upper
      for (int sd = 0; sd < 2; ++sd) {
        const int pd = r->ctx[nbs[sd]]->dom.device;
        if (pd == r->ctx[k]->dom.device) { ok[sd] = true; continue; }
        int can = 0; hipDeviceCanAccessPeer(&can, r->ctx[k]->dom.device, pd);
        if (can) { const hipError_t pe = hipDeviceEnablePeerAccess(pd, 0); ok[sd] = (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled); }
        (void)hipGetLastError();
      }
      r->peer_ok.push_back(ok);
    }
    if (e != hipSuccess) { fail("bflbm_ring_create: %s", hipGetErrorString(e)); bflbm_ring_destroy(r); return 1; }
  }
  *out = r;
  return 0;
}

int bflbm_ring_destroy(bflbm_ring* r) {
  if (!r) return 0;
  for (size_t k = 0; k < r->ctx.size(); ++k) {
    hipSetDevice(r->ctx[k]->dom.device);
    if (k < r->comm.size() && r->comm[k]) { hipStreamSynchronize(r->comm[k]); hipStreamDestroy(r->comm[k]); }
    if (k < r->packed.size() && r->packed[k]) hipEventDestroy(r->packed[k]);
    if (k < r->unpacked.size() && r->unpacked[k]) hipEventDestroy(r->unpacked[k]);
    bflbm_destroy(r->ctx[k]);
  }
  delete r;
  return 0;
}

int bflbm_ring_size(const bflbm_ring* r, int* nslabs) {
  if (!r || !nslabs) return fail("null argument");
  *nslabs = (int)r->ctx.size();
  return 0;
}

int bflbm_ring_slab(bflbm_ring* r, int slab, bflbm_ctx** ctx) {
  if (!r || !ctx || slab < 0 || slab >= (int)r->ctx.size()) return fail("bflbm_ring_slab: bad argument");
  *ctx = r->ctx[slab];
  return 0;
}

int bflbm_ring_set_params(bflbm_ring* r, const bflbm_params* p) {
  if (!r) return fail("null ring");
  for (bflbm_ctx* c : r->ctx) if (bflbm_set_params(c, p)) return 1;
  return 0;
}
int bflbm_ring_set_schedule(bflbm_ring* r, int schedule) {
  if (!r) return fail("null ring");
  for (bflbm_ctx* c : r->ctx) if (bflbm_set_schedule(c, schedule)) return 1;
  return 0;
}
int bflbm_ring_init_mixture(bflbm_ring* r) {
  if (!r) return fail("null ring");
  for (bflbm_ctx* c : r->ctx) if (bflbm_init_mixture(c)) return 1;
  return 0;
}
int bflbm_ring_init_stripe(bflbm_ring* r, double frac) {
  if (!r) return fail("null ring");
  for (bflbm_ctx* c : r->ctx) if (bflbm_init_stripe(c, frac)) return 1;
  return 0;
}
int bflbm_ring_init_droplet(bflbm_ring* r, double radius) {
  if (!r) return fail("null ring");
  for (bflbm_ctx* c : r->ctx) if (bflbm_init_droplet(c, radius)) return 1;
  return 0;
}

int bflbm_ring_commit_upload(bflbm_ring* r, int reset) {
  if (!r) return fail("null ring");
  if (ring_exchange(r, BFLBM_HALO_UPLOAD) || ring_join(r)) return 1;
  for (bflbm_ctx* c : r->ctx) if (bflbm_commit_upload(c, reset)) return 1;
  {                                                         // every slab resolves `auto` on the whole lattice's total density
    double m = 0.;
    for (bflbm_ctx* c : r->ctx) m = std::max(m, c->total_max);
    for (bflbm_ctx* c : r->ctx) c->total_max = m;
  }
  if (ring_exchange(r, BFLBM_HALO_STATE) || ring_join(r)) return 1;
  return bflbm_ring_sync(r);
}

// 1 (default): the faces move behind the interior sweep; 0: after it (what bench.py reports as halo_overlap)
int bflbm_ring_set_overlap(bflbm_ring* r, int on) {
  if (!r) return fail("null ring");
  r->overlap = on != 0;
  return 0;
}
// 0 (default): one gather kernel per face reads the neighbour's planes in place (peer memory over xGMI) where the
// neighbour is reachable, per-plane copies otherwise; 1: always the copy engine (hipMemcpyPeerAsync, 38 copies per face:
// no compute units, so nothing competes with the one-workgroup-per-CU interior sweep)
int bflbm_ring_set_transport(bflbm_ring* r, int transport) {
  if (!r) return fail("null ring");
  if (transport < 0 || transport > 1) return fail("unknown ring transport %d", transport);
  r->transport = transport;
  return 0;
}
// faces the last exchange moved with the gather kernel / with the copy engine (2 per slab in total)
int bflbm_ring_last_transport(const bflbm_ring* r, int* kernel_faces, int* copy_faces) {
  if (!r || !kernel_faces || !copy_faces) return fail("null argument");
  *kernel_faces = r->last_kernel_faces; *copy_faces = r->last_copy_faces;
  return 0;
}

int bflbm_ring_set_step_count(bflbm_ring* r, long long n) {
  if (!r) return fail("null ring");
  for (bflbm_ctx* c : r->ctx) if (bflbm_set_step_count(c, n)) return 1;
  return 0;
}

int bflbm_ring_com_sums(bflbm_ring* r, double sums[4]);
// reference-state noise: the slabs need the GLOBAL centre of mass of the resident state
static int ring_prepare_ref(bflbm_ring* r) {
  bflbm_ctx* c0 = r->ctx[0];
  if (!ref_active(c0) || c0->com_valid) return 0;
  if (c0->steps == c0->ref_kind_step && c0->ref_kind == 1) return 0;
  double s[4];
  if (bflbm_ring_com_sums(r, s)) return 1;
  const double com[3] = { s[1] / s[0], s[2] / s[0], s[3] / s[0] };
  for (bflbm_ctx* c : r->ctx) if (bflbm_set_com(c, com)) return 1;
  return 0;
}

int bflbm_ring_set_ref_state(bflbm_ring* r, const double* rho_eq, const double* phi_eq, const double* rhot_eq, const bflbm_fab* box) {
  if (!r) return fail("null ring");
  for (bflbm_ctx* c : r->ctx) if (bflbm_set_ref_state(c, rho_eq, phi_eq, rhot_eq, box)) return 1;
  return 0;
}
int bflbm_ring_enable_ref_state(bflbm_ring* r, int on, const double com_ref[3]) {
  if (!r) return fail("null ring");
  for (bflbm_ctx* c : r->ctx) if (bflbm_enable_ref_state(c, on, com_ref)) return 1;
  return 0;
}
int bflbm_ring_prepare_ref(bflbm_ring* r) {     // before reading noise / hydrovs of the slabs
  if (!r) return fail("null ring");
  return r->ctx.size() > 1 ? ring_prepare_ref(r) : 0;
}

int bflbm_ring_step(bflbm_ring* r, int nsteps) {
  if (!r) return fail("null ring");
  if (nsteps < 0) return fail("nsteps < 0");
  if (r->ctx.size() == 1) return bflbm_step(r->ctx[0], nsteps);
  for (int s = 0; s < nsteps; ++s) {
    if (ring_prepare_ref(r)) return 1;
    for (bflbm_ctx* c : r->ctx) if (bflbm_step_boundary(c)) return 1;
    if (r->overlap && ring_mark(r)) return 1;               // boundary planes final: the neighbours may copy them
    for (bflbm_ctx* c : r->ctx) if (bflbm_step_interior(c)) return 1;   // main streams sweep the interior ...
    if (!r->overlap && ring_mark(r)) return 1;              // (measurement mode: the faces wait for the whole sweep)
    if (ring_copy(r, BFLBM_HALO_NEXT)) return 1;            // ... while the comm streams move the faces (enqueued after the
                                                            // interior launches so that the 76 copy calls per slab delay nothing)
    if (ring_join(r)) return 1;
    for (bflbm_ctx* c : r->ctx) if (bflbm_step_finish(c)) return 1;
  }
  return 0;
}

int bflbm_ring_com_sums(bflbm_ring* r, double sums[4]) {
  if (!r || !sums) return fail("null argument");
  for (int k = 0; k < 4; ++k) sums[k] = 0.;
  for (bflbm_ctx* c : r->ctx) { double s[4]; if (bflbm_com_sums(c, s)) return 1; for (int k = 0; k < 4; ++k) sums[k] += s[k]; }
  return 0;
}
int bflbm_ring_mass(bflbm_ring* r, double* rho_sum, double* phi_sum) {
  if (!r || !rho_sum || !phi_sum) return fail("null argument");
  *rho_sum = 0.; *phi_sum = 0.;
  for (bflbm_ctx* c : r->ctx) { double a, b; if (bflbm_mass(c, &a, &b)) return 1; *rho_sum += a; *phi_sum += b; }
  return 0;
}
int bflbm_ring_sync(bflbm_ring* r) {
  if (!r) return fail("null ring");
  for (size_t k = 0; k < r->ctx.size(); ++k) {
    if (bflbm_sync(r->ctx[k])) return 1;
    if (k < r->comm.size()) HIP_TRY(hipStreamSynchronize(r->comm[k]));
  }
  return 0;
}

}  // extern "C"

#include "bflbm_sf.h"
#include "bflbm_sf_ring.h"
#include "bflbm_droplet.h"
