"""GPU: the numbers Surface_Tension.ipynb records from runs of the REFERENCE itself (cells 13, 17, 18, 19;
executed 2025-11-17): nine 32^3 droplets (rho_hi = 3, tau = 1/2, kBT = 0; alpha0 = 1.5 with five initial
radii and alpha0 = 1.7 with four), frame 20000.  The notebook prints, with all
digits, rho and phi at the droplet centre and at the box edge, the Shan-Chen force integral along x
(sum of rho*afx + phi*agx), the ideal-gas and interaction pressure differences, and the surface tension
from the Laplace-law regression of Delta P against 1/R.

These are integrated observables of 20000-step trajectories of the real reference.  Two inputs are not
printed in the notebook and were identified by reproducing its numbers (tools/surface_tension_probe.py):
kappa = 0.1 for the alpha0 = 1.5 set and 1.0 for the alpha0 = 1.7 set (LBM_binary.H:30 lists 4, 3 and 0.1 as
values in use), and, for the alpha0 = 1.5 set, the initial radii 0.2, 0.225, 0.25, 0.275, 0.3 -- the
directory names the notebook builds with "{:.2f}" show 0.225 and 0.275 as 0.23 and 0.28 (the alpha0 = 1.7
set really used 0.23 and 0.28).  With them all nine systems agree in every printed digit but the last two.
The notebook's run was made with another build of the reference (compiler / FMA contraction unknown; SURVEY
8d measured 1e-16 ... 4e-14 drift between FMA-on and FMA-off builds of the reference itself), so the
densities are required to agree to 1e-12 relative -- they do to 1e-13 or better -- and the derived numbers
to 1e-10."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CS2 = 1.0 / 3.0
N = 32

# (r_init, rho_out, rho_in, phi_out, phi_in, deltaP_SC, ideal_gas_deltaP, pressure_SC_difference)
CELL13 = [   # alpha0 = 1.5, kappa = 0.1, centre index nc = [16, 16, 16]
    (0.20, 0.015052155677499688, 3.507447510454257, 3.0222255117146184, -0.003077451215475287,
     0.00489361852887258, 0.15569746394888787, 0.028142503749551406),
    (0.225, 0.013621050407596126, 3.4578172393862876, 3.034387245103636, -0.002739184961826447,
     0.004265141328972856, 0.13568991963774302, 0.025401571302297265),
    (0.25, 0.012456369275410611, 3.425288786290128, 3.0470747495470656, -0.002430603580569477,
     0.003823620345275787, 0.12110902129569408, 0.023140503739289407),
    (0.275, 0.011473863283384058, 3.4036412715508835, 3.0604556059868004, -0.0021676762686879836,
     0.0034987793282993396, 0.10984804200400362, 0.021246620809713353),
    (0.30, 0.01068003965435518, 3.38999906807189, 3.0742614432910687, -0.0019692496094205047,
     0.003237157857496349, 0.10102944517234849, 0.019754494231470122),
]
CELL18 = [   # alpha0 = 1.7, kappa = 1.0, centre index nc = [15, 15, 15]
    (0.20, -0.003565045170913126, 3.4858659715293148, 3.0438690233711494, -0.004324518882526176,
     0.004655645674758513, 0.14707915814885078, 0.0023931255034644753),
    (0.23, -0.003909066109038856, 3.443755869229836, 3.055102632724311, -0.004992601208137102,
     0.004087472378379656, 0.1291899004688089, 0.002975397546149272),
    (0.25, -0.003930582655519604, 3.4231219737779863, 3.06257884738863, -0.005188115733803547,
     0.003758232250920278, 0.11976186443702419, 0.0032423724142349097),
    (0.28, -0.004116095171956733, 3.401966974378537, 3.075401082560991, -0.005631951517444264,
     0.0034014332845998807, 0.1083500118240194, 0.0036839393923357782),
]
# cells 17 / 19: fitted equilibrium radii used by the notebook and the regression it prints
R_15 = [0.1760534, 0.20426208, 0.23111422, 0.25739767, 0.2831091]
R_17 = [0.19280456, 0.2219286, 0.24129567, 0.27032776]
GAMMA_15 = (0.021567889346707517, 0.004859111326412077, 0.010783944673353758)   # slope, intercept, slope / 2
GAMMA_17 = (0.026914662086370552, 0.0050272470551585985, 0.013457331043185276)


def _run(pkg, alpha0, kappa, r):
    lbm = pkg.BinaryLBM(N, N, N, params=pkg.default_params(rho_hi=3.0, alpha0=alpha0, kappa=kappa))
    lbm.LBM_init_droplet(r)
    lbm.LBM_timestep(20000)
    h = lbm.LBM_hydrovars()
    fit = lbm.fit_droplet()
    lbm.close()
    return {k: np.ascontiguousarray(h[c].transpose(2, 1, 0)) for k, c in
            dict(rho=0, phi=1, rhot=5, afx=9, agx=12).items()}, fit      # notebook indexing [x, y, z]


def _check_system(fields, alpha0, nc, rec):
    r, rho_out, rho_in, phi_out, phi_in, dp_sc, dp_ideal, dp_int = rec
    rho, phi, rhot, afx, agx = (fields[k] for k in ("rho", "phi", "rhot", "afx", "agx"))
    got = np.array([rho[0, nc, nc], rho[nc, nc, nc], phi[0, nc, nc], phi[nc, nc, nc]])
    want = np.array([rho_out, rho_in, phi_out, phi_in])
    assert np.all(np.abs(got - want) <= 1e-12 * np.abs(want)), (r, got, want)   # all printed digits but the last two or three
    dr = 1.0 / N
    force_rho = force_phi = 0.0
    for ix in range(N // 2):                                              # the notebook's loop, same order
        force_phi = force_phi + phi[ix, nc, nc] * agx[ix, nc, nc] * dr
        force_rho = force_rho + rho[ix, nc, nc] * afx[ix, nc, nc] * dr
    assert abs((force_rho + force_phi) - dp_sc) <= 1e-10 * abs(dp_sc), (r, force_rho + force_phi)
    assert abs((rhot[nc, nc, nc] - rhot[0, nc, nc]) * CS2 - dp_ideal) <= 1e-11 * dp_ideal
    got_int = alpha0 * CS2 * got[0] * got[2] - alpha0 * CS2 * got[1] * got[3]
    assert abs(got_int - dp_int) <= 1e-10 * abs(dp_int)
    p_in = rhot[nc, nc, nc] * CS2 + alpha0 * CS2 * rho[nc, nc, nc] * phi[nc, nc, nc]
    p_out = rhot[0, nc, nc] * CS2 + alpha0 * CS2 * rho[0, nc, nc] * phi[0, nc, nc]
    return p_in - p_out


def _regression(radii, dps):
    x = 1.0 / np.array(radii)
    y = np.array(dps)
    k = np.sum((x - x.mean()) * (y - y.mean())) / np.sum((x - x.mean()) ** 2)
    return k, y.mean() - k * x.mean()


@pytest.mark.parametrize("alpha0,kappa,nc,systems,radii,gamma",
                         [(1.5, 0.1, 16, CELL13, R_15, GAMMA_15), (1.7, 1.0, 15, CELL18, R_17, GAMMA_17)])
def test_surface_tension_notebook_numbers(pkg, alpha0, kappa, nc, systems, radii, gamma):
    dps, fits = [], []
    for rec in systems:
        fields, fit = _run(pkg, alpha0, kappa, rec[0])
        dps.append(_check_system(fields, alpha0, nc, rec))
        fits.append(fit[2])
    k, b = _regression(radii, dps)                                        # with the notebook's own radii
    assert abs(k - gamma[0]) <= 1e-9 * gamma[0] and abs(b - gamma[1]) <= 1e-8 * gamma[1], (k, b)
    assert abs(k / 2 - gamma[2]) <= 1e-9 * gamma[2]
    # the radii the notebook quotes are tanh fits of the same fields (cell 8: scipy curve_fit); the device fit
    # of the same model finds them to the digits quoted (8 significant)
    np.testing.assert_allclose(fits, radii, rtol=2e-6)


def test_surface_tension_notebook_cell4_maxima(pkg):
    """Cell 4 (alpha0 = 2, rho_hi = 3, r = 0.2, 32^3; kappa = 3 identified from the printed initial maximum):
    'max rho at step0: 2.998 dimensionless, step1: 3.648 dimensionless' and the conserved-mass ratio '1.000e+00'."""
    lbm = pkg.BinaryLBM(N, N, N, params=pkg.default_params(rho_hi=3.0, alpha0=2.0, kappa=3.0))
    lbm.LBM_init_droplet(0.2)
    rho0 = lbm.LBM_hydrovars(ncomp=1)[0]
    m0 = lbm.mass()[0]
    lbm.LBM_timestep(20000)
    rho1 = lbm.LBM_hydrovars(ncomp=1)[0]
    assert "%.3f" % rho0.max() == "2.998" and "%.3f" % rho1.max() == "3.648"
    assert "%.3e" % (lbm.mass()[0] / m0) == "1.000e+00"
    lbm.close()
