// bflbm_sf.h -- running structure factors <a^(k) b^*(k)>/N of pairs of hydrodynamic fields on the GPU
// (SURVEY 8f rank 2).  The reference accumulates them with FHDeX's StructFact::FortStructure every
// out_SF_step steps (main_run_job.cpp:301-310, :342-349) and writes them with WritePlotFile (:50-54);
// FHDeX is un-vendored, so the definition follows the call sites and what Mixture.ipynb reads (see
// structfact.py, the host-side twin this is tested against).
//
// One frame: k_observe materialises hydrovs (or hydrovsbar) of the resident state densely in the scratch
// state buffer, hipFFT D2Z transforms every variable that occurs in a pair (half spectrum, nx/2+1), and
// k_sf_accumulate adds a^ conj(b^)/N for every pair.  k_sf_expand produces the mean over the frames as
// the full fft-shifted spectrum (Hermitian completion; k = 0 optionally removed) for download.
// hipFFT is resolved with dlopen at first use: libbflbm.so itself does not depend on it.
// Included by bflbm.hip inside its extern "C" block's translation unit (needs bflbm_ctx).
#ifndef BFLBM_SF_H_
#define BFLBM_SF_H_

#include <dlfcn.h>
#include <hipfft/hipfft.h>

namespace {

struct FftApi {
  void* handle = nullptr;
  hipfftResult (*plan3d)(hipfftHandle*, int, int, int, hipfftType) = nullptr;
  hipfftResult (*set_stream)(hipfftHandle, hipStream_t) = nullptr;
  hipfftResult (*exec_d2z)(hipfftHandle, hipfftDoubleReal*, hipfftDoubleComplex*) = nullptr;
  hipfftResult (*destroy)(hipfftHandle) = nullptr;
  bool tried = false;
};
FftApi g_fft;

int load_fft() {
  if (g_fft.plan3d) return 0;
  if (g_fft.tried) return fail("hipFFT is not available (libhipfft.so.0 could not be loaded)");
  g_fft.tried = true;
  for (const char* name : {"libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so.0"}) {
    g_fft.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_fft.handle) break;
  }
  if (!g_fft.handle) return fail("hipFFT is not available: %s", dlerror());
  g_fft.plan3d = (decltype(g_fft.plan3d))dlsym(g_fft.handle, "hipfftPlan3d");
  g_fft.set_stream = (decltype(g_fft.set_stream))dlsym(g_fft.handle, "hipfftSetStream");
  g_fft.exec_d2z = (decltype(g_fft.exec_d2z))dlsym(g_fft.handle, "hipfftExecD2Z");
  g_fft.destroy = (decltype(g_fft.destroy))dlsym(g_fft.handle, "hipfftDestroy");
  if (!g_fft.plan3d || !g_fft.set_stream || !g_fft.exec_d2z || !g_fft.destroy) {
    g_fft.plan3d = nullptr;
    return fail("hipFFT: missing symbols in the loaded library");
  }
  return 0;
}

// acc[p][k] += scale[p] * a^(k) conj(b^(k)) / N over the half spectrum
struct SfPairs { int n; int a[32]; int b[32]; double scale[32]; };
__global__ void __launch_bounds__(256) k_sf_accumulate(const double2* __restrict__ hat, double2* __restrict__ acc,
                                                       long long nk, SfPairs P, double inv_n) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nk) return;
  const int p = blockIdx.y;
  const double2 a = hat[(long long)P.a[p] * nk + k], b = hat[(long long)P.b[p] * nk + k];
  const double s = P.scale[p];
  const double ar = s * a.x, ai = s * a.y;
  double2 v = acc[(long long)p * nk + k];
  v.x += (ar * b.x + ai * b.y) * inv_n;
  v.y += (ai * b.x - ar * b.y) * inv_n;
  acc[(long long)p * nk + k] = v;
}

// full, fft-shifted mean spectrum of the pairs from the half-spectrum accumulators:
// out[p][z][y][x] with k = 0 at (nz/2, ny/2, nx/2); what: 0 |S|, 1 Re S, 2 Im S
__global__ void __launch_bounds__(256) k_sf_expand(const double2* __restrict__ acc, double* __restrict__ out,
                                                   int nx, int ny, int nz, double inv_samples, int what, int zero_avg) {
  const long long n = (long long)nx * ny * nz;
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int p = blockIdx.y;
  const int xs = (int)(s % nx), ys = (int)((s / nx) % ny), zs = (int)(s / ((long long)nx * ny));
  // fftshift: shifted index i holds frequency index (i - n/2) mod n  (numpy: shift by n//2)
  int kx = xs - nx / 2; if (kx < 0) kx += nx;
  int ky = ys - ny / 2; if (ky < 0) ky += ny;
  int kz = zs - nz / 2; if (kz < 0) kz += nz;
  const int nxc = nx / 2 + 1;
  double re, im;
  if (kx < nxc) {
    const double2 v = acc[(long long)p * nxc * ny * nz + ((long long)kz * ny + ky) * nxc + kx];
    re = v.x; im = v.y;
  } else {                                           // S(-k) = conj(S(k)) for real fields
    const int mx = nx - kx, my = (ny - ky) % ny, mz = (nz - kz) % nz;
    const double2 v = acc[(long long)p * nxc * ny * nz + ((long long)mz * ny + my) * nxc + mx];
    re = v.x; im = -v.y;
  }
  re *= inv_samples; im *= inv_samples;
  if (zero_avg && kx == 0 && ky == 0 && kz == 0) { re = 0.; im = 0.; }
  out[(long long)p * n + s] = (what == 0) ? hypot(re, im) : (what == 1 ? re : im);
}

}  // namespace

struct bflbm_sf {
  bflbm_ctx* c = nullptr;
  SfPairs pairs;
  std::vector<int> vars;          // distinct variables, in the order of their spectra in `hat`
  int nvar_fields = 0;            // components the observation must produce
  long long nk = 0;               // half-spectrum size
  double2* hat = nullptr;
  double2* acc = nullptr;
  hipfftHandle plan = nullptr;
  long long nsamples = 0;
  size_t bytes = 0;
};

extern "C" {

int bflbm_sf_create(bflbm_ctx* c, int npairs, const int* var_a, const int* var_b, const double* scale, bflbm_sf** out) {
  if (!c || !var_a || !var_b || !out) return fail("null argument");
  if (npairs < 1 || npairs > 32) return fail("bflbm_sf_create: 1..32 pairs");
  if (!c->G.zwrap) return fail("bflbm_sf_create: structure factors need the whole lattice in one context (nranks == 1)");
  if (load_fft()) return 1;
  HIP_TRY(hipSetDevice(c->dom.device));
  bflbm_sf* s = new bflbm_sf;
  s->c = c;
  s->pairs.n = npairs;
  for (int p = 0; p < npairs; ++p) {
    if (var_a[p] < 0 || var_a[p] >= BFLBM_NHYDRO || var_b[p] < 0 || var_b[p] >= BFLBM_NHYDRO) { delete s; return fail("bflbm_sf_create: variable index outside hydrovs"); }
    for (int v : {var_a[p], var_b[p]})
      if (std::find(s->vars.begin(), s->vars.end(), v) == s->vars.end()) s->vars.push_back(v);
    s->pairs.scale[p] = scale ? scale[p] : 1.0;
  }
  std::sort(s->vars.begin(), s->vars.end());
  for (int p = 0; p < npairs; ++p) {
    s->pairs.a[p] = (int)(std::find(s->vars.begin(), s->vars.end(), var_a[p]) - s->vars.begin());
    s->pairs.b[p] = (int)(std::find(s->vars.begin(), s->vars.end(), var_b[p]) - s->vars.begin());
  }
  s->nvar_fields = s->vars.back() + 1;
  const int nx = c->G.nx, ny = c->G.ny, nz = c->G.nz;
  s->nk = (long long)(nx / 2 + 1) * ny * nz;
  const size_t hb = (size_t)s->vars.size() * s->nk * sizeof(double2), ab = (size_t)npairs * s->nk * sizeof(double2);
  if (hipMalloc((void**)&s->hat, hb) != hipSuccess || hipMalloc((void**)&s->acc, ab) != hipSuccess) {
    if (s->hat) hipFree(s->hat);
    delete s;
    return fail("bflbm_sf_create: out of device memory (%zu bytes)", hb + ab);
  }
  s->bytes = hb + ab;
  hipMemsetAsync(s->acc, 0, ab, c->stream);
  if (g_fft.plan3d(&s->plan, nz, ny, nx, HIPFFT_D2Z) != HIPFFT_SUCCESS) { hipFree(s->hat); hipFree(s->acc); delete s; return fail("hipfftPlan3d failed for %d x %d x %d", nx, ny, nz); }
  g_fft.set_stream(s->plan, c->stream);
  HIP_TRY(hipStreamSynchronize(c->stream));
  *out = s;
  return 0;
}

int bflbm_sf_destroy(bflbm_sf* s) {
  if (!s) return 0;
  hipSetDevice(s->c->dom.device);
  hipStreamSynchronize(s->c->stream);
  if (s->plan) g_fft.destroy(s->plan);
  if (s->hat) hipFree(s->hat);
  if (s->acc) hipFree(s->acc);
  delete s;
  return 0;
}

int bflbm_sf_reset(bflbm_sf* s) {
  if (!s) return fail("null argument");
  HIP_TRY(hipSetDevice(s->c->dom.device));
  HIP_TRY(hipMemsetAsync(s->acc, 0, (size_t)s->pairs.n * s->nk * sizeof(double2), s->c->stream));
  s->nsamples = 0;
  return 0;
}

// FortStructure(fields, reset): one frame of the resident state.  lb_hydrovars != 0 takes hydrovsbar
// (the shipped STRUCT_LB_HYDROVARS build, main_run_job.cpp:19, :344) instead of hydrovs.
int bflbm_sf_accumulate(bflbm_sf* s, int lb_hydrovars, int reset) {
  if (!s) return fail("null argument");
  bflbm_ctx* c = s->c;
  if (c->step_open) return fail("structure factor requested inside an open step");
  if (lb_hydrovars && s->nvar_fields > BFLBM_NHYDROBAR) return fail("bflbm_sf_accumulate: pair variables outside hydrovsbar");
  if (reset && bflbm_sf_reset(s)) return 1;
  HIP_TRY(hipSetDevice(c->dom.device));
  g_fft.set_stream(s->plan, c->stream);              // the context's stream may have been replaced (bflbm_set_stream) since the plan was made
  if (!lb_hydrovars && ensure_density(c)) return 1;
  if (!lb_hydrovars && prepare_ref(c)) return 1;
  const RefState Rf = ref_state(c);
  double* fields = c->S[1 - c->cur];                 // dense [comp][z][y][x]
  dim3 g = plane_grid(c, c->nzl), b(256);
  const uint32_t idx = (uint32_t)c->steps;
  const int inj = c->inject ? 1 : 0;
  if (lb_hydrovars) hipLaunchKernelGGL((k_observe<0>), g, b, 0, c->stream, c->S[c->cur], c->rho, c->phi, c->injf, c->injg, fields, c->G, c->dp, own_lo(c), idx, BFLBM_NHYDROBAR, inj, Rf);
  else              hipLaunchKernelGGL((k_observe<2>), g, b, 0, c->stream, c->S[c->cur], c->rho, c->phi, c->injf, c->injg, fields, c->G, c->dp, own_lo(c), idx, s->nvar_fields, inj, Rf);
  HIP_TRY(hipGetLastError());
  const long long n = (long long)c->nzl * c->G.dplane;
  for (size_t v = 0; v < s->vars.size(); ++v)
    if (g_fft.exec_d2z(s->plan, fields + (long long)s->vars[v] * n, (hipfftDoubleComplex*)(s->hat + (long long)v * s->nk)) != HIPFFT_SUCCESS)
      return fail("hipfftExecD2Z failed");
  dim3 ga((unsigned)((s->nk + 255) / 256), (unsigned)s->pairs.n);
  hipLaunchKernelGGL(k_sf_accumulate, ga, dim3(256), 0, c->stream, s->hat, s->acc, s->nk, s->pairs, 1.0 / (double)n);
  HIP_TRY(hipGetLastError());
  s->nsamples += 1;
  return 0;
}

int bflbm_sf_nsamples(const bflbm_sf* s, long long* n) {
  if (!s || !n) return fail("null argument");
  *n = s->nsamples;
  return 0;
}

// mean over the accumulated frames, fft-shifted (k = 0 at cell n/2), dst[npairs][nz][ny][nx] on the host;
// what: 0 magnitude, 1 real part, 2 imaginary part; zero_avg != 0 removes the k = 0 mode (WritePlotFile's flag)
int bflbm_sf_get(bflbm_sf* s, int what, int zero_avg, double* dst) {
  if (!s || !dst) return fail("null argument");
  if (what < 0 || what > 2) return fail("bflbm_sf_get: what must be 0, 1 or 2");
  bflbm_ctx* c = s->c;
  if (c->step_open) return fail("structure factor requested inside an open step");
  HIP_TRY(hipSetDevice(c->dom.device));
  const long long n = (long long)c->nzl * c->G.dplane;
  double* out = c->S[1 - c->cur];                    // 38 component volumes of scratch >= 32 pairs
  dim3 g((unsigned)((n + 255) / 256), (unsigned)s->pairs.n);
  hipLaunchKernelGGL(k_sf_expand, g, dim3(256), 0, c->stream, s->acc, out, c->G.nx, c->G.ny, c->G.nz,
                     1.0 / (double)std::max(s->nsamples, 1LL), what, zero_avg);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(dst, out, (size_t)s->pairs.n * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"

#endif  // BFLBM_SF_H_
