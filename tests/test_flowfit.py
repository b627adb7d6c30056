"""The closed forms of the reference's gradient-flow droplet fit (externlib.H:108-253, :344-371; restated on the host in
csrc/bflbm_droplet.h, C-ABI bflbm_flowfit_coefficients) against numerical quadrature -- no GPU needed (the library loads
without one).  No output of this fit is recorded anywhere in the reference ("parity unpinned"), but its building blocks are
closed-form integrals, so they can be checked against their definitions:

    I_n(c)  = int_{-c}^{inf} (x + c)^n sech^4(x) dx, n = 2, 3, 4       (integral_func2_series, d = delta = 1)
    K_R / 2 = 1/s  int 4 pi r^2 m(r) sech^2((R - r)/s) dr              (KRn: the model's own Mf_R)
    K_W / 2 = 1/s^3 int 4 pi r^2 m(r) (R - r) sech^2((R - r)/s) dr     (KWn: the model's own Mf_W)

with m(r) = 1/2 (1 + tanh((R - r)/s)), s = sqrt(2W).  The reference truncates its series (20 Taylor terms of sech^4 up
to |x| = 1, 50 / 100 terms of the alternating 1/k^2 sums), which is what the tolerances below reflect."""
import ctypes

import numpy as np
import pytest
from scipy import integrate


def _coef(pkg, W, R, eta_W=0.2, eta_R=0.2, dt=0.02, C0=1.0):
    out = (ctypes.c_double * 9)()
    assert pkg._lib.load().bflbm_flowfit_coefficients(W, R, eta_W, eta_R, dt, C0, out) == 0
    return dict(zip("Jrr Jwr Jrw Jww Kw Kr I2 I3 I4".split(), list(out)))


@pytest.mark.parametrize("W,R", [(0.02, 0.3), (0.005, 0.2), (0.0008, 0.17), (0.003, 0.28)])
def test_sech4_moment_series_against_quadrature(pkg, W, R):
    c = R / np.sqrt(2 * W)
    k = _coef(pkg, W, R)
    for n in (2, 3, 4):
        q, _ = integrate.quad(lambda x: (x + c) ** n / np.cosh(x) ** 4, -c, 60, epsabs=1e-13, epsrel=1e-13, limit=400)
        assert abs(k[f"I{n}"] - q) < 5e-6 * abs(q), (n, k[f"I{n}"], q)


@pytest.mark.parametrize("W,R", [(0.02, 0.3), (0.005, 0.2), (0.0008, 0.17)])
def test_model_integrals_against_quadrature(pkg, W, R):
    """At rho = the model profile the flow's right-hand side Mf - K/2 vanishes: K/2 is the model's own lattice integral."""
    s = np.sqrt(2 * W)
    m = lambda r: 0.5 * (1 + np.tanh((R - r) / s))
    sech2 = lambda r: 1 / np.cosh((R - r) / s) ** 2
    kr, _ = integrate.quad(lambda r: 4 * np.pi * r * r * m(r) * sech2(r), 0, R + 40 * s, epsabs=1e-14, epsrel=1e-13, limit=400)
    kw, _ = integrate.quad(lambda r: 4 * np.pi * r * r * m(r) * (R - r) * sech2(r), 0, R + 40 * s, epsabs=1e-14, epsrel=1e-13, limit=400)
    k = _coef(pkg, W, R)
    assert abs(0.5 * k["Kr"] - kr / s) < 3e-4 * abs(kr / s)
    assert abs(0.5 * k["Kw"] - kw / s ** 3) < 1e-2 * abs(kw / s ** 3)


def test_linearisation_scales_with_its_prefactors(pkg):
    """J_RR, J_RW carry eta_R dt C0 and J_WR, J_WW eta_W dt C0 (externlib.H:199-253): doubling one doubles exactly those."""
    a = _coef(pkg, 0.01, 0.25)
    b = _coef(pkg, 0.01, 0.25, eta_R=0.4)
    c = _coef(pkg, 0.01, 0.25, C0=3.0)
    for key in ("Jrr", "Jrw"):
        assert abs(b[key] - 2 * a[key]) < 1e-15 * abs(a[key]) + 1e-300
    for key in ("Jwr", "Jww", "Kw", "Kr"):
        assert b[key] == a[key]
    for key in ("Jrr", "Jwr", "Jrw", "Jww"):
        assert abs(c[key] - 3 * a[key]) < 4e-16 * abs(3 * a[key])
    assert pkg._lib.load().bflbm_flowfit_coefficients(-1.0, 0.2, 0.2, 0.2, 0.02, 1.0, (ctypes.c_double * 9)()) != 0
