"""Droplet observables of the validation notebooks, as host-side post-processing of a density field.

The reference analyses its plotfiles in Droplet_Fluctuation.ipynb / Surface_Tension.ipynb (cell 3 there
defines the quantities; its C++ twin is LBM_hydrovs.H:117-335, off by default, main_run_job.cpp:111).
These are the same definitions re-stated on numpy arrays so that the numbers the notebooks record can
be checked directly (tests/test_gpu_fullsize.py).  The *_from_moments functions evaluate the same
quantities from the 20 raw moments the device reduces (BinaryLBM.droplet_moments, csrc/bflbm_droplet.h),
so that no field has to leave the GPU; BinaryLBM.fit_droplet is the device version of fit_droplet.

Conventions of the notebooks: coordinates are cell centres (i + 1/2)/n of a unit box; fields are
indexed [x, y, z] there -- pass `rho_xyz = rho.transpose(2, 1, 0)` for an array in (z, y, x) order.
"""
import numpy as np


def _centres(shape):
    return [(np.arange(n) + 0.5) / n for n in shape]


def centre_of_mass(rho_xyz):
    """sum(r rho)/sum(rho) with cell-centred coordinates."""
    x, y, z = _centres(rho_xyz.shape)
    m = rho_xyz.sum()
    return np.array([(rho_xyz * x[:, None, None]).sum(), (rho_xyz * y[None, :, None]).sum(),
                     (rho_xyz * z[None, None, :]).sum()]) / m


def radial_profile(rho_xyz, r0=None):
    """Flattened densities and their distance from r0 (default: the centre of mass)."""
    r0 = centre_of_mass(rho_xyz) if r0 is None else np.asarray(r0)
    x, y, z = _centres(rho_xyz.shape)
    r = np.sqrt((x[:, None, None] - r0[0]) ** 2 + (y[None, :, None] - r0[1]) ** 2 + (z[None, None, :] - r0[2]) ** 2)
    return rho_xyz.ravel(), r.ravel()


def fit_droplet(rho_xyz, r0=None):
    """Least-squares fit of rho(r) = hi - (hi - lo)/2 (1 + tanh((r - R)/W)); returns (hi, lo, R, W).
    Start values as in the notebook: (max rho, min rho, 0.5, 0.5)."""
    from scipy.optimize import curve_fit

    def profile(r, hi, lo, radius, width):
        return hi - (hi - lo) / 2 * (1 + np.tanh((r - radius) / width))

    rho, r = radial_profile(rho_xyz, r0)
    popt, _ = curve_fit(profile, r, rho, p0=[rho.max(), rho.min(), 0.5, 0.5])
    return tuple(popt)


def mass_covariance(rho_xyz):
    """Second central moments of the mass distribution, trapezoid-weighted like the notebook
    (end planes count half), about the (unweighted) centre of mass."""
    x, y, z = _centres(rho_xyz.shape)
    wt = np.ones(rho_xyz.shape)
    for ax in range(3):
        sl = [slice(None)] * 3
        for end in (0, -1):
            sl[ax] = end
            wt[tuple(sl)] *= 0.5
    r0 = centre_of_mass(rho_xyz)
    w = rho_xyz * wt
    m = w.sum()
    d = [x[:, None, None] - r0[0], y[None, :, None] - r0[1], z[None, None, :] - r0[2]]
    c = np.empty((3, 3))
    for a in range(3):
        for b in range(a, 3):
            c[a, b] = c[b, a] = (d[a] * d[b] * w).sum() / m
    return c


def principal_axes(rho_xyz, radius):
    """Eigen-decomposition of the mass covariance and the semi-axes of the equal-volume ellipsoid."""
    ev, vec = np.linalg.eig(mass_covariance(rho_xyz))
    axes = np.array([ev[k] ** (1. / 3.) * radius / (ev[(k + 1) % 3] * ev[(k + 2) % 3]) ** (1. / 6.) for k in range(3)])
    return axes, ev, vec


# ---- the same observables from the device-reduced raw moments (BinaryLBM.droplet_moments) ----------------
# m[0:10] = sum rho {1, x, y, z, xx, xy, xz, yy, yz, zz} in cell indices, m[10:20] trapezoid-weighted.

def com_from_moments(m, n, weighted=False):
    """Centre of mass in unit-box cell-centre coordinates; weighted=True is the reference's C++ getCenterOfMass
    (trapezoid weights, LBM_hydrovs.H:62-113), False the notebooks' plain sum."""
    o = 10 if weighted else 0
    return np.array([(m[o + 1 + d] / m[o] + 0.5) / n[d] for d in range(3)])


def covariance_from_moments(m, n, kind="notebook"):
    """Second central mass moments. kind="notebook": trapezoid-weighted sums about the unweighted COM,
    normalised by the weighted mass (mass_covariance above); kind="reference": plain sums about the
    trapezoid-weighted COM, normalised by the plain mass (fittingDropletCovariance, LBM_hydrovs.H:262-335)."""
    if kind == "notebook":
        r0, o = com_from_moments(m, n, False), 10
    else:
        r0, o = com_from_moments(m, n, True), 0
    s0 = m[o]
    s1 = np.array([(m[o + 1 + d] + 0.5 * s0) / n[d] for d in range(3)])          # sum w (i+1/2)/n
    idx = {(0, 0): 4, (0, 1): 5, (0, 2): 6, (1, 1): 7, (1, 2): 8, (2, 2): 9}
    c = np.empty((3, 3))
    for (a, b), k in idx.items():
        # sum w (i+1/2)(j+1/2) = S_ab + (S_a + S_b)/2 + S_0/4
        s2 = (m[o + k] + 0.5 * (m[o + 1 + a] + m[o + 1 + b]) + 0.25 * s0) / (n[a] * n[b])
        c[a, b] = c[b, a] = (s2 - r0[a] * s1[b] - r0[b] * s1[a] + r0[a] * r0[b] * s0) / s0
    return c


def principal_axes_from_moments(m, n, radius, kind="notebook"):
    ev, vec = np.linalg.eig(covariance_from_moments(m, n, kind))
    axes = np.array([ev[k] ** (1. / 3.) * radius / (ev[(k + 1) % 3] * ev[(k + 2) % 3]) ** (1. / 6.) for k in range(3)])
    return axes, ev, vec
