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
Host-side (numpy FFT of downloaded fields), like the reference's own host-gathered FFT (SURVEY 8f rank 2
lists a rocFFT version as "next").
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

    def write_plotfile(self, step, time, root, zero_avg=1, max_grid_size=None):
        s = self.mean(zero_avg)
        nz, ny, nx = s.shape[1:]
        lo = (-nx / 2 - 0.5, -ny / 2 - 0.5, -nz / 2 - 0.5)
        hi = (nx / 2 - 0.5, ny / 2 - 0.5, nz / 2 - 0.5)
        names = self.pair_names()
        pf.write_plotfile(pf.concatenate(root + "_mag", step, 9), np.abs(s), names, time, step, max_grid_size, lo, hi)
        ri = np.concatenate([s.real, s.imag])
        pf.write_plotfile(pf.concatenate(root + "_real_imag", step, 9), ri,
                          [n + "_real" for n in names] + [n + "_imag" for n in names], time, step, max_grid_size, lo, hi)
        return s
