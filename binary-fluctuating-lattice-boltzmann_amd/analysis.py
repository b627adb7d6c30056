"""Droplet observables of the validation notebooks, as host-side post-processing of a density field.

The reference analyses its plotfiles in Droplet_Fluctuation.ipynb / Surface_Tension.ipynb (cell 3 there
defines the quantities; its C++ twin is LBM_hydrovs.H:117-335, off by default, main_run_job.cpp:111).
These are the same definitions re-stated on numpy arrays so that the numbers the notebooks record can
be checked directly (tests/test_gpu_fullsize.py); SURVEY.md 8f ranks an on-device version as "next".

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
