"""GPU: the two noise-run observables the reference's authors recorded.  The generated
noise is pinned against the reference only statistically (amrex::RandomNormal is not reproducible),
so these are physics checks with stated statistical bands; the runs are deterministic (fixed seed).

 * Droplet_Fluctuation.ipynb cells 8-19 (run of the reference dated 2025-12-23): 64^3 droplet, header defaults
   (alpha0 = 4, kappa = 4, rho_hi = 1, tau = 1/2), r_init = 0.2, kBT = 5e-5, continued from the kBT = 0 state;
   centre of mass of where(rho > 0.06, rho, 0) (COM_PARAM = 2, COM_THRESHOLD = 0.06) every 50 steps from step
   40000 to 200000, mean square displacement over lags <= 100 frames, linear fit:
       Slope 5.574605966831772e-06, D_fit = 9.291009944719619e-07   (Stokes-Hasimoto estimate 9.4629e-07).
 * Flat_Interface.ipynb cells 7-9 (2025-11-14): 8x256x64 stripe, alpha0 = 1.5, rho_lo = 0.1, rho_hi = 3
   (kappa = 0.1, identified in tools/flat_interface_probe.py), kBT = 1e-5; height of the rho = 1.55 contour at
   x = 4 every 2000 steps from 500000 to 800000; <|h_k|^2> (numpy's unnormalised DFT, mean height removed) is
   plotted against kBT / (gamma k^2) with gamma = 0.012162.  The notebook records the line, not numbers; the
   printed heights lie between 47.50 and 47.55.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_droplet_diffusion_coefficient_of_the_notebook(pkg):
    """D_fit within 25 % of the recorded 9.291e-7: the three Cartesian components of one such trajectory scatter by
    +-30 % around their mean (each is an independent estimate), so the mean of the three carries about +-17 %."""
    n, kBT, every = 64, 5e-5, 50
    lbm = pkg.BinaryLBM(n, n, n)
    lbm.LBM_init_droplet(0.2)
    lbm.LBM_timestep(20000)                                  # the kBT = 0 state the noise run continues from
    hb = lbm.LBM_hydrovars_density()
    com_ref = lbm.update_com()
    f, g = lbm.populations()
    lbm.set_params(kBT=kBT)
    lbm.set_ref_state(hb[0], hb[1], hb[5], com_ref)          # the droplet recipe's reference-state noise (ReadMe.ipynb#3 case B)
    lbm.LBM_init(f, g)
    lbm.LBM_timestep(40000)
    idx = np.arange(n, dtype=np.float64)
    r = []
    for _ in range(40000, 200001, every):
        rho = lbm.LBM_hydrovars_density(ncomp=1)[0]          # [z, y, x]
        fld = np.where(rho > 0.06, rho, 0.0)                 # COM_PARAM = 2, COM_THRESHOLD = 0.06 (cells 8, 10)
        m = fld.sum()
        r.append([(fld.sum(axis=(0, 1)) * idx).sum() / m, (fld.sum(axis=(0, 2)) * idx).sum() / m, (fld.sum(axis=(1, 2)) * idx).sum() / m])
        lbm.LBM_timestep(every)
    lbm.close()
    r = np.array(r)
    assert len(r) == 3201
    msd = np.insert(np.array([np.mean(np.sum((r[k:3200] - r[:3200 - k]) ** 2, axis=1)) for k in range(1, 101)]), 0, 0.0)   # compute_msd, cell 18
    slope = np.polyfit(np.arange(101) * every, msd, 1)[0]
    d_fit = slope / 6.0
    assert abs(d_fit / 9.291009944719619e-07 - 1.0) < 0.25, d_fit
    assert abs(d_fit / 9.4629e-07 - 1.0) < 0.30, d_fit       # Stokes-Einstein with the Hasimoto correction (cell 19)
    assert np.abs(r - r[0]).max() < 4.0                       # the droplet stays near the box centre (no wrap in the estimator)


def test_capillary_wave_spectrum_of_the_notebook(pkg):
    """<|h_k|^2> k^2 gamma / kBT with the notebook's gamma = 0.012162: 151 frames of an exponentially distributed
    |h_k|^2 give +-8 % per mode if the frames were independent; the slow modes are not, so modes 2..8 are required
    within [0.6, 1.5] each and their mean within [0.8, 1.2]."""
    nx, ny, nz = 8, 256, 64
    kBT, gamma, level = 1e-5, 0.012162, 1.55
    lbm = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(alpha0=1.5, rho_lo=0.1, rho_hi=3.0, kappa=0.1))
    lbm.LBM_init_stripe(0.5)
    lbm.LBM_timestep(2000)
    f, g = lbm.populations()
    lbm.set_params(kBT=kBT)
    lbm.LBM_init(f, g)

    def heights():
        up = lbm.LBM_hydrovars_density(ncomp=1)[0][nz // 2:, :, 4]      # upper half [z, y] at x = 4: rho falls through the level
        k = np.argmax(up < level, axis=0)
        a, b = up[k - 1, np.arange(ny)], up[k, np.arange(ny)]
        return nz // 2 + (k - 1) + (level - a) / (b - a)                 # linear interpolation like skimage.find_contours
    lbm.LBM_timestep(500000)
    H = []
    for _ in range(500000, 800001, 2000):
        H.append(heights())
        lbm.LBM_timestep(2000)
    lbm.close()
    H = np.array(H)
    assert len(H) == 151
    assert 47.45 < H.mean() < 47.60                                       # the notebook's heights: 47.50 ... 47.55
    h = H - H.mean(axis=0)
    hk2 = np.mean(np.abs(np.fft.fft(h, axis=1)) ** 2, axis=0)            # fft norm "backward", cell 9
    k = 2 * np.pi * np.fft.fftfreq(ny)
    ratio = hk2[2:9] * gamma * k[2:9] ** 2 / kBT
    assert np.all((ratio > 0.6) & (ratio < 1.5)), ratio
    assert 0.8 < ratio.mean() < 1.2, ratio
