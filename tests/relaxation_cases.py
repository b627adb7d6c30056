"""Shared set-up of the tau != 1/2 invariant tests (tests/test_oracle_pins.py on the CPU oracle,
tests/test_gpu_relaxation.py on the GPU path).

Every number the reference's authors recorded comes from tau = 1/2, where 1/tau_bar = 1 and the relaxation of
LBM_binary.H:504-511 is a full overwrite, so no recorded reference output pins the partial relaxation
("parity unpinned for tau_bar != 1").  Two closed-form consequences of :504-511 stand in:
  * a spatially uniform perturbation along one non-conserved moment a of one fluid (no density, no momentum, hence
    no change of the equilibrium, no force, nothing for streaming to move) decays by EXACTLY 1 - 1/tau_bar per
    step, tau_bar = tau + 1/2, separately for f (tau_f) and g (tau_g);
  * a transverse shear wave u_y = A sin(k x) of the co-moving mixture decays like exp(-nu k^2 t) with the
    kinematic viscosity nu = cs2 tau (up to the O(k^2) lattice correction).
"""
import ctypes

import numpy as np

Q = 19


def _vec(ob, fn, v):
    out = np.zeros(Q)
    v = np.ascontiguousarray(v, dtype=np.float64)
    getattr(ob.lib(), fn)(v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out


def mode_pattern(ob, a):
    """populations of the unit moment e_a"""
    m = np.zeros(Q)
    m[a] = 1.0
    return _vec(ob, "orc_populations", m)


def moment(ob, f19, a):
    return _vec(ob, "orc_moments", f19)[a]


def uniform_mode_state(ob, n, a, eps_f=1e-3, eps_g=-5e-4):
    _, w, _ = ob.lattice_tables()
    w = np.asarray(w)
    pat = mode_pattern(ob, a)
    f = np.broadcast_to((w + eps_f * pat)[:, None, None, None], (Q, n, n, n)).copy()
    g = np.broadcast_to((w + eps_g * pat)[:, None, None, None], (Q, n, n, n)).copy()
    return f, g


def shear_wave_state(ob, nx, ny, nz, amp=1e-4):
    c, w, _ = ob.lattice_tables()
    c = np.asarray(c).reshape(Q, 3)
    uy = amp * np.sin(2 * np.pi * np.arange(nx) / nx)[None, None, :] * np.ones((nz, ny, 1))
    f = np.stack([w[i] * (1.0 + 3.0 * c[i, 1] * uy) for i in range(Q)])
    return f, f.copy()


def shear_amplitude(ob, f, g):
    c, _, _ = ob.lattice_tables()
    c = np.asarray(c).reshape(Q, 3)
    jy = sum(c[i, 1] * (f[i] + g[i]) for i in range(Q))[0, 0, :]
    return 2.0 * np.abs(np.fft.fft(jy)[1]) / jy.size
