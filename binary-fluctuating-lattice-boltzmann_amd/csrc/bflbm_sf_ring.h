// bflbm_sf_ring.h -- structure factors of a lattice that is decomposed into z-slabs (bflbm_ring), needed by
// configs[4]-style multi-GPU runs whose primary fluctuation check is FHDeX's StructFact on hydrovs
// (main_run_job.cpp:301-310, :342-349).  Same definition and normalisation as bflbm_sf.h, computed as a slab FFT:
//   1. every slab materialises the pair variables of its own planes and transforms each plane in (x,y)
//      (hipFFT D2Z, batched over the slab's planes): h2[var][z local][ky][kx'], kx' = 0..nx/2;
//   2. transpose: the ky range is split over the slabs; slab s collects the rows ky in [ky0_s, ky1_s) of every plane
//      of every slab with strided (peer) copies: zb[var][z global][ky local][kx'];
//   3. every slab transforms its columns along z (hipFFT Z2Z, stride = row block, batched) and adds
//      scale * a^ conj(b^) / N for every pair to its part of the accumulator acc[pair][kz][ky local][kx'].
// bflbm_ring_sf_get assembles the half spectrum on the host and expands it (Hermitian completion, fft shift, optional
// k = 0 removal) exactly like k_sf_expand.  A ring of one slab delegates to bflbm_sf.
#ifndef BFLBM_SF_RING_H_
#define BFLBM_SF_RING_H_

namespace {

struct FftApiMany {
  hipfftResult (*plan_many)(hipfftHandle*, int, int*, int*, int, int, int*, int, int, hipfftType, int) = nullptr;
  hipfftResult (*exec_z2z)(hipfftHandle, hipfftDoubleComplex*, hipfftDoubleComplex*, int) = nullptr;
};
FftApiMany g_fftm;

int load_fft_many() {
  if (load_fft()) return 1;
  if (g_fftm.plan_many) return 0;
  g_fftm.plan_many = (decltype(g_fftm.plan_many))dlsym(g_fft.handle, "hipfftPlanMany");
  g_fftm.exec_z2z = (decltype(g_fftm.exec_z2z))dlsym(g_fft.handle, "hipfftExecZ2Z");
  if (!g_fftm.plan_many || !g_fftm.exec_z2z) { g_fftm.plan_many = nullptr; return fail("hipFFT: hipfftPlanMany / hipfftExecZ2Z missing"); }
  return 0;
}

struct SlabSf {
  int ky0 = 0, ky1 = 0;             // rows of the (ky, kx') plane this slab transforms along z
  hipfftHandle plan2d = nullptr, plan1d = nullptr;
  double2* h2 = nullptr;            // [var][nzl][ny][nxc]
  double2* zb = nullptr;            // [var][nz][nky][nxc]
  double2* acc = nullptr;           // [pair][nz][nky][nxc]
};

}  // namespace

struct bflbm_ring_sf {
  bflbm_ring* r = nullptr;
  bflbm_sf* single = nullptr;       // ring of one slab
  SfPairs pairs;
  std::vector<int> vars;
  int nvar_fields = 0;
  int nxc = 0;
  std::vector<SlabSf> slab;
  long long nsamples = 0;
};

extern "C" {

int bflbm_ring_sf_destroy(bflbm_ring_sf* s) {
  if (!s) return 0;
  if (s->single) bflbm_sf_destroy(s->single);
  for (size_t k = 0; k < s->slab.size(); ++k) {
    hipSetDevice(s->r->ctx[k]->dom.device);
    hipStreamSynchronize(s->r->ctx[k]->stream);
    SlabSf& q = s->slab[k];
    if (q.plan2d) g_fft.destroy(q.plan2d);
    if (q.plan1d) g_fft.destroy(q.plan1d);
    if (q.h2) hipFree(q.h2);
    if (q.zb) hipFree(q.zb);
    if (q.acc) hipFree(q.acc);
  }
  delete s;
  return 0;
}

int bflbm_ring_sf_create(bflbm_ring* r, int npairs, const int* var_a, const int* var_b, const double* scale, bflbm_ring_sf** out) {
  if (!r || !var_a || !var_b || !out) return fail("null argument");
  if (npairs < 1 || npairs > 32) return fail("bflbm_ring_sf_create: 1..32 pairs");
  bflbm_ring_sf* s = new bflbm_ring_sf;
  s->r = r;
  const int n = (int)r->ctx.size();
  if (n == 1) {
    if (bflbm_sf_create(r->ctx[0], npairs, var_a, var_b, scale, &s->single)) { delete s; return 1; }
    *out = s;
    return 0;
  }
  if (load_fft_many()) { delete s; return 1; }
  s->pairs.n = npairs;
  for (int p = 0; p < npairs; ++p) {
    if (var_a[p] < 0 || var_a[p] >= BFLBM_NHYDRO || var_b[p] < 0 || var_b[p] >= BFLBM_NHYDRO) { delete s; return fail("bflbm_ring_sf_create: variable index outside hydrovs"); }
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
  const Geo& G0 = r->ctx[0]->G;
  const int nx = G0.nx, ny = G0.ny, nz = G0.nz, nxc = nx / 2 + 1;
  s->nxc = nxc;
  if (ny < n) { delete s; return fail("bflbm_ring_sf_create: fewer rows (ny = %d) than slabs (%d)", ny, n); }
  const size_t nv = s->vars.size();
  s->slab.resize((size_t)n);
  for (int k = 0; k < n; ++k) {
    bflbm_ctx* c = r->ctx[k];
    SlabSf& q = s->slab[k];
    q.ky0 = (int)((long long)ny * k / n); q.ky1 = (int)((long long)ny * (k + 1) / n);
    const int nky = q.ky1 - q.ky0;
    hipError_t e = hipSetDevice(c->dom.device);
    if (e == hipSuccess) e = hipMalloc((void**)&q.h2, nv * (size_t)c->nzl * ny * nxc * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void**)&q.zb, nv * (size_t)nz * nky * nxc * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void**)&q.acc, (size_t)npairs * nz * nky * nxc * sizeof(double2));
    if (e == hipSuccess) e = hipMemsetAsync(q.acc, 0, (size_t)npairs * nz * nky * nxc * sizeof(double2), c->stream);
    if (e != hipSuccess) { fail("bflbm_ring_sf_create: %s", hipGetErrorString(e)); bflbm_ring_sf_destroy(s); return 1; }
    int n2[2] = { ny, nx };
    if (g_fftm.plan_many(&q.plan2d, 2, n2, nullptr, 1, ny * nx, nullptr, 1, ny * nxc, HIPFFT_D2Z, c->nzl) != HIPFFT_SUCCESS) {
      fail("hipfftPlanMany (2-D, %d planes) failed", c->nzl); bflbm_ring_sf_destroy(s); return 1;
    }
    int n1[1] = { nz };
    int emb[1] = { nz };
    const int col = nky * nxc;                         // columns of one variable: stride between the elements of a column
    if (g_fftm.plan_many(&q.plan1d, 1, n1, emb, col, 1, emb, col, 1, HIPFFT_Z2Z, col) != HIPFFT_SUCCESS) {
      fail("hipfftPlanMany (z, %d columns) failed", col); bflbm_ring_sf_destroy(s); return 1;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  *out = s;
  return 0;
}

int bflbm_ring_sf_reset(bflbm_ring_sf* s) {
  if (!s) return fail("null argument");
  s->nsamples = 0;
  if (s->single) return bflbm_sf_reset(s->single);
  const int nz = s->r->ctx[0]->G.nz;
  for (size_t k = 0; k < s->slab.size(); ++k) {
    bflbm_ctx* c = s->r->ctx[k];
    HIP_TRY(hipSetDevice(c->dom.device));
    HIP_TRY(hipMemsetAsync(s->slab[k].acc, 0, (size_t)s->pairs.n * nz * (s->slab[k].ky1 - s->slab[k].ky0) * s->nxc * sizeof(double2), c->stream));
  }
  return 0;
}

int bflbm_ring_sf_nsamples(const bflbm_ring_sf* s, long long* n) {
  if (!s || !n) return fail("null argument");
  if (s->single) return bflbm_sf_nsamples(s->single, n);
  *n = s->nsamples;
  return 0;
}

int bflbm_ring_sf_accumulate(bflbm_ring_sf* s, int lb_hydrovars, int reset) {
  if (!s) return fail("null argument");
  if (s->single) return bflbm_sf_accumulate(s->single, lb_hydrovars, reset);
  bflbm_ring* r = s->r;
  const int n = (int)r->ctx.size();
  if (lb_hydrovars && s->nvar_fields > BFLBM_NHYDROBAR) return fail("bflbm_ring_sf_accumulate: pair variables outside hydrovsbar");
  if (reset && bflbm_ring_sf_reset(s)) return 1;
  if (!lb_hydrovars && ring_prepare_ref(r)) return 1;
  const Geo& G0 = r->ctx[0]->G;
  const int ny = G0.ny, nz = G0.nz, nxc = s->nxc;
  const size_t nv = s->vars.size();
  // 1. observe + 2-D transforms of the own planes
  for (int k = 0; k < n; ++k) {
    bflbm_ctx* c = r->ctx[k];
    if (c->step_open) return fail("structure factor requested inside an open step");
    HIP_TRY(hipSetDevice(c->dom.device));
    if (!lb_hydrovars && ensure_density(c)) return 1;
    const RefState Rf = ref_state(c);
    double* fields = c->S[1 - c->cur];
    dim3 g = plane_grid(c, c->nzl), b(256);
    const uint32_t idx = (uint32_t)c->steps;
    const int inj = c->inject ? 1 : 0;
    if (lb_hydrovars) hipLaunchKernelGGL((k_observe<0>), g, b, 0, c->stream, c->S[c->cur], c->rho, c->phi, c->injf, c->injg, fields, c->G, c->dp, own_lo(c), idx, BFLBM_NHYDROBAR, inj, Rf);
    else              hipLaunchKernelGGL((k_observe<2>), g, b, 0, c->stream, c->S[c->cur], c->rho, c->phi, c->injf, c->injg, fields, c->G, c->dp, own_lo(c), idx, s->nvar_fields, inj, Rf);
    HIP_TRY(hipGetLastError());
    g_fft.set_stream(s->slab[k].plan2d, c->stream);
    const long long nloc = (long long)c->nzl * c->G.dplane;
    for (size_t v = 0; v < nv; ++v)
      if (g_fft.exec_d2z(s->slab[k].plan2d, fields + (long long)s->vars[v] * nloc,
                         (hipfftDoubleComplex*)(s->slab[k].h2 + (long long)v * c->nzl * ny * nxc)) != HIPFFT_SUCCESS)
        return fail("hipfftExecD2Z (planes of slab %d) failed", k);
  }
  for (int k = 0; k < n; ++k) { HIP_TRY(hipSetDevice(r->ctx[k]->dom.device)); HIP_TRY(hipStreamSynchronize(r->ctx[k]->stream)); }
  // 2. transpose: slab d collects its ky rows of every plane of every slab t
  for (int d = 0; d < n; ++d) {
    bflbm_ctx* cd = r->ctx[d];
    SlabSf& qd = s->slab[d];
    const int nky = qd.ky1 - qd.ky0;
    HIP_TRY(hipSetDevice(cd->dom.device));
    for (int t = 0; t < n; ++t) {
      bflbm_ctx* ct = r->ctx[t];
      for (size_t v = 0; v < nv; ++v) {
        const double2* src = s->slab[t].h2 + ((long long)v * ct->nzl * ny + qd.ky0) * nxc;               // plane 0 of slab t, row ky0
        double2* dst = qd.zb + ((long long)v * nz + ct->dom.z0) * (long long)nky * nxc;                  // global plane z0_t
        HIP_TRY(hipMemcpy2DAsync(dst, (size_t)nky * nxc * sizeof(double2), src, (size_t)ny * nxc * sizeof(double2),
                                 (size_t)nky * nxc * sizeof(double2), (size_t)ct->nzl, hipMemcpyDefault, cd->stream));
      }
    }
  }
  // 3. z transforms and pair products on every slab's rows
  const double inv_n = 1.0 / ((double)G0.nx * ny * nz);
  for (int d = 0; d < n; ++d) {
    bflbm_ctx* cd = r->ctx[d];
    SlabSf& qd = s->slab[d];
    const long long col = (long long)(qd.ky1 - qd.ky0) * nxc, nk = col * nz;
    HIP_TRY(hipSetDevice(cd->dom.device));
    g_fft.set_stream(qd.plan1d, cd->stream);
    for (size_t v = 0; v < nv; ++v) {
      hipfftDoubleComplex* p = (hipfftDoubleComplex*)(qd.zb + (long long)v * nk);
      if (g_fftm.exec_z2z(qd.plan1d, p, p, HIPFFT_FORWARD) != HIPFFT_SUCCESS) return fail("hipfftExecZ2Z (columns of slab %d) failed", d);
    }
    dim3 ga((unsigned)((nk + 255) / 256), (unsigned)s->pairs.n);
    hipLaunchKernelGGL(k_sf_accumulate, ga, dim3(256), 0, cd->stream, qd.zb, qd.acc, nk, s->pairs, inv_n);
    HIP_TRY(hipGetLastError());
  }
  for (int k = 0; k < n; ++k) { HIP_TRY(hipSetDevice(r->ctx[k]->dom.device)); HIP_TRY(hipStreamSynchronize(r->ctx[k]->stream)); }
  s->nsamples += 1;
  return 0;
}

// mean over the accumulated frames, fft-shifted, dst[npairs][nz][ny][nx] on the host (like bflbm_sf_get)
int bflbm_ring_sf_get(bflbm_ring_sf* s, int what, int zero_avg, double* dst) {
  if (!s || !dst) return fail("null argument");
  if (s->single) return bflbm_sf_get(s->single, what, zero_avg, dst);
  if (what < 0 || what > 2) return fail("bflbm_ring_sf_get: what must be 0, 1 or 2");
  bflbm_ring* r = s->r;
  const Geo& G0 = r->ctx[0]->G;
  const int nx = G0.nx, ny = G0.ny, nz = G0.nz, nxc = s->nxc, np = s->pairs.n;
  // half spectrum acc[pair][kz][ky][kx'] assembled from the slabs' row blocks
  std::vector<double2> acc((size_t)np * nz * ny * nxc);
  for (size_t k = 0; k < s->slab.size(); ++k) {
    bflbm_ctx* c = r->ctx[k];
    const SlabSf& q = s->slab[k];
    const int nky = q.ky1 - q.ky0;
    HIP_TRY(hipSetDevice(c->dom.device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // [pair][kz] blocks of nky*nxc contiguous -> rows ky0.. of the full plane
    HIP_TRY(hipMemcpy2D(acc.data() + (size_t)q.ky0 * nxc, (size_t)ny * nxc * sizeof(double2), q.acc, (size_t)nky * nxc * sizeof(double2),
                        (size_t)nky * nxc * sizeof(double2), (size_t)np * nz, hipMemcpyDeviceToHost));
  }
  const double inv_samples = 1.0 / (double)std::max(s->nsamples, 1LL);
  const long long nn = (long long)nx * ny * nz;
  for (int p = 0; p < np; ++p)
    for (int zs = 0; zs < nz; ++zs) for (int ys = 0; ys < ny; ++ys) for (int xs = 0; xs < nx; ++xs) {
      int kx = xs - nx / 2; if (kx < 0) kx += nx;
      int ky = ys - ny / 2; if (ky < 0) ky += ny;
      int kz = zs - nz / 2; if (kz < 0) kz += nz;
      double re, im;
      if (kx < nxc) {
        const double2 v = acc[(size_t)p * nxc * ny * nz + ((size_t)kz * ny + ky) * nxc + kx];
        re = v.x; im = v.y;
      } else {                                         // S(-k) = conj(S(k)) for real fields
        const int mx = nx - kx, my = (ny - ky) % ny, mz = (nz - kz) % nz;
        const double2 v = acc[(size_t)p * nxc * ny * nz + ((size_t)mz * ny + my) * nxc + mx];
        re = v.x; im = -v.y;
      }
      re *= inv_samples; im *= inv_samples;
      if (zero_avg && kx == 0 && ky == 0 && kz == 0) { re = 0.; im = 0.; }
      dst[(size_t)p * nn + ((size_t)zs * ny + ys) * nx + xs] = (what == 0) ? hypot(re, im) : (what == 1 ? re : im);
    }
  return 0;
}

}  // extern "C"

#endif  // BFLBM_SF_RING_H_
