"""Running structure factors <a^(k) b^*(k)>/N of pairs of hydrodynamic fields.

The reference accumulates them with FHDeX's `StructFact` (main_run_job.cpp:301-310 sets the 22 variable
pairs, :342-349 calls `FortStructure(hydrovsbar, 0)` every out_SF_step steps inside the window, :50-54
`WritePlotFile(step, time, root + "_SF", zero_avg)`), and Mixture.ipynb cell 2 reads the result.  FHDeX is
an un-vendored dependency (GNUmakefile:2, version unpinned), so this is a re-statement from the call
sites and from what the notebook consumes -- "parity unpinned" for anything beyond that:
  * plotfiles `<root>_mag%09d` (magnitude) and `<root>_real_imag%09d`, fields `struct_fact_<B>_<A>`
    (e.g. pair A=ufx, B=ugx -> `struct_fact_ugx_ufx`; `_real` / `_imag` suffixes in the second file);
  * k = 0 at cell n/2 (fftshift), domain [-n/2 - 1/2, n/2 - 1/2] (the notebook's domain_left/right_edge);
  * normalisation: S(k) = a^(k) conj(b^(k)) / N with un-normalised FFTs, so that S_rho ~ rho kBT/cs2;
  * zero_avg != 0 removes the k = 0 mode.
`StructFact` is host-side (numpy FFT of downloaded fields); `DeviceStructFact` is the same accumulator on
the GPU (hipFFT D2Z on the resident state, csrc/bflbm_sf.h) with the same interface and file output.
"""
import numpy as np

from . import plotfile as pf

# main_run_job.cpp:301-306
PAIR_A = [0, 1, 0, 2, 3, 4, 6, 7, 8, 2, 9, 15, 16, 17, 15, 18, 19, 20, 21, 20, 20, 21]
PAIR_B = [0, 1, 1, 2, 3, 4, 6, 7, 8, 6, 9, 15, 16, 17, 16, 18, 19, 20, 21, 21, 18, 18]


class StructFact:
    def __init__(self, var_names, pair_a=PAIR_A, pair_b=PAIR_B, var_scaling=None):
        self.names = list(var_names)
        self.pairs = [(a, b) for a, b in zip(pair_a, pair_b) if a < len(self.names) and b < len(self.names)]
        self.scale = [1.0] * len(self.pairs) if var_scaling is None else list(var_scaling)
        self.acc = None
        self.nsamples = 0

    def pair_names(self):
        return ["struct_fact_%s_%s" % (self.names[b], self.names[a]) for a, b in self.pairs]

    def reset(self):
        self.acc = None
        self.nsamples = 0

    def fort_structure(self, fields, reset=0):
        """Accumulate one frame; `fields` is (ncomp, nz, ny, nx) like hydrovs / hydrovsbar."""
        if reset:
            self.reset()
        need = sorted({i for p in self.pairs for i in p})
        hat = {i: np.fft.fftn(fields[i]) for i in need}
        n = fields[0].size
        cur = np.stack([s * hat[a] * np.conj(hat[b]) / n for (a, b), s in zip(self.pairs, self.scale)])
        self.acc = cur if self.acc is None else self.acc + cur
        self.nsamples += 1

    def mean(self, zero_avg=1):
        s = self.acc / max(self.nsamples, 1)
        if zero_avg:
            s = s.copy()
            s[:, 0, 0, 0] = 0.0
        return np.fft.fftshift(s, axes=(1, 2, 3))

    def _parts(self, zero_avg):
        s = self.mean(zero_avg)
        return np.abs(s), s.real, s.imag

    def write_plotfile(self, step, time, root, zero_avg=1, max_grid_size=None):
        mag, re, im = self._parts(zero_avg)
        nz, ny, nx = mag.shape[1:]
        lo = (-nx / 2 - 0.5, -ny / 2 - 0.5, -nz / 2 - 0.5)
        hi = (nx / 2 - 0.5, ny / 2 - 0.5, nz / 2 - 0.5)
        names = self.pair_names()
        pf.write_plotfile(pf.concatenate(root + "_mag", step, 9), mag, names, time, step, max_grid_size, lo, hi)
        ri = np.concatenate([re, im])
        pf.write_plotfile(pf.concatenate(root + "_real_imag", step, 9), ri,
                          [n + "_real" for n in names] + [n + "_imag" for n in names], time, step, max_grid_size, lo, hi)
        return re + 1j * im


class DeviceStructFact(StructFact):
    """The accumulator on the device(s): no field leaves the GPU until the mean is written.
    `lbm` is a single-context BinaryLBM or a RingLBM (slab-decomposed lattice: distributed slab FFT,
    csrc/bflbm_sf_ring.h); frames are taken from its resident state."""

    def __init__(self, lbm, var_names, pair_a=PAIR_A, pair_b=PAIR_B, var_scaling=None, lb_hydrovars=False):
        import ctypes
        from . import _lib
        super().__init__(var_names, pair_a, pair_b, var_scaling)
        self.lbm, self.lb = lbm, bool(lb_hydrovars)
        self._ct, self._check, self._libh = ctypes, _lib.check, lbm.lib
        n = len(self.pairs)
        a = (ctypes.c_int * n)(*[p[0] for p in self.pairs])
        b = (ctypes.c_int * n)(*[p[1] for p in self.pairs])
        sc = (ctypes.c_double * n)(*[float(v) for v in self.scale])
        h = ctypes.c_void_p()
        self._pre = "bflbm_ring_sf_" if type(lbm).__name__ == "RingLBM" else "bflbm_sf_"
        _lib.check(getattr(lbm.lib, self._pre + "create")(lbm._h, n, a, b, sc, ctypes.byref(h)))
        self._h = h
        if not hasattr(lbm, "_dependents"):
            lbm._dependents = []
        lbm._dependents.append(self)             # closed before the context it lives on

    def close(self):
        if getattr(self, "_h", None):
            getattr(self._libh, self._pre + "destroy")(self._h)
            self._h = None
            deps = getattr(self.lbm, "_dependents", [])
            if self in deps:
                deps.remove(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        if getattr(self, "_h", None):
            self._check(getattr(self._libh, self._pre + "reset")(self._h))
        self.nsamples = 0

    def fort_structure(self, fields=None, reset=0):
        """FortStructure on the resident state (`fields` is ignored: nothing is downloaded)."""
        self._check(getattr(self._libh, self._pre + "accumulate")(self._h, int(self.lb), int(bool(reset))))
        self.nsamples = 1 if reset else self.nsamples + 1

    def _get(self, what, zero_avg):
        nx, ny, nz = self.lbm.n
        out = np.empty((len(self.pairs), nz, ny, nx))
        self._check(getattr(self._libh, self._pre + "get")(self._h, what, int(bool(zero_avg)), out.ctypes.data_as(self._ct.c_void_p)))
        return out

    def mean(self, zero_avg=1):
        return self._get(1, zero_avg) + 1j * self._get(2, zero_avg)

    def _parts(self, zero_avg):
        return self._get(0, zero_avg), self._get(1, zero_avg), self._get(2, zero_avg)

    def magnitude(self, zero_avg=1):
        return self._get(0, zero_avg)
