#!/usr/bin/env python3
"""Exploration: the two noise-run observables the reference's notebooks record.
 (1) Droplet_Fluctuation.ipynb cells 17-19: 64^3 droplet (header defaults, r_init = 0.2), kBT = 5e-5, continued from the
     kBT = 0 equilibrium with reference-state noise; centre of mass every 50 steps from 40000 to 200000, MSD over lags
     <= 100 frames, D_fit = slope / 6 = 9.29e-7 (theory with Hasimoto correction 9.46e-7).
 (2) Flat_Interface.ipynb cells 7-9: 8x256x64 stripe (alpha0 = 1.5, rho_lo = 0.1, rho_hi = 3, kappa = 0.1), kBT = 1e-5;
     interface height h(y) at x = 4 every 2000 steps from 500000 to 800000; <|h_k|^2> against kBT / (gamma k^2) with
     gamma = 0.012162 (a figure only, no printed number).
usage: tools/noise_anchor_probe.py droplet|interface [scale]   (scale < 1 shortens the run)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
what = sys.argv[1]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0

if what.startswith("droplet"):
    n, kBT = 64, 5e-5
    use_ref = what != "droplet_noref"
    t0 = time.time()
    lbm = pkg.BinaryLBM(n, n, n)
    lbm.LBM_init_droplet(0.2)
    lbm.LBM_timestep(20000)
    hb = lbm.LBM_hydrovars_density()
    com_ref = lbm.update_com()
    f, g = lbm.populations()
    lbm.set_params(kBT=kBT)
    if use_ref:
        lbm.set_ref_state(hb[0], hb[1], hb[5], com_ref)
    lbm.LBM_init(f, g)
    skip, total, every = int(40000 * scale), int(200000 * scale), 50
    lbm.LBM_timestep(skip)
    idx = np.arange(n, dtype=np.float64)
    r_plain, r_thr = [], []
    for s in range(skip, total + 1, every):
        rho = lbm.LBM_hydrovars_density(ncomp=1)[0]                      # [z, y, x]
        for store, fld in ((r_plain, rho), (r_thr, np.where(rho > 0.06, rho, 0.0))):   # COM_PARAM = 0 / 2 (threshold 0.06)
            m = fld.sum()
            store.append([(fld.sum(axis=(0, 1)) * idx).sum() / m, (fld.sum(axis=(0, 2)) * idx).sum() / m, (fld.sum(axis=(1, 2)) * idx).sum() / m])
        lbm.LBM_timestep(every)
    print("reference-state noise:", use_ref, " elapsed %.1f s" % (time.time() - t0))
    for name, r in (("plain centre of mass", np.array(r_plain)), ("thresholded (rho > 0.06) centre of mass, the notebook's COM_PARAM = 2", np.array(r_thr))):
        lags = np.arange(1, 101)
        msd = np.insert(np.array([np.mean(np.sum((r[k:] - r[:-k]) ** 2, axis=1)) for k in lags]), 0, 0.0)
        slope, icpt = np.polyfit(np.arange(0, 101) * every, msd, 1)
        per_axis = [np.polyfit(lags * every, [np.mean((r[k:, ax] - r[:-k, ax]) ** 2) for k in lags], 1)[0] / 2 for ax in range(3)]
        print("  %s: frames %d  D_fit = %.4e (notebook 9.291e-7, Stokes-Hasimoto 9.463e-7)  msd(lag 100) %.4e  per axis %s" % (name, len(r), slope / 6, msd[100], ["%.3e" % v for v in per_axis]))
else:
    nx, ny, nz = 8, 256, 64
    kBT, gamma, level = 1e-5, 0.012162, 1.55
    t0 = time.time()
    par = dict(alpha0=1.5, rho_lo=0.1, rho_hi=3.0, kappa=0.1)
    lbm = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(**par))
    lbm.LBM_init_stripe(0.5)
    lbm.LBM_timestep(2000)
    f, g = lbm.populations()
    lbm.set_params(kBT=kBT)
    lbm.LBM_init(f, g)

    def heights():
        rho = lbm.LBM_hydrovars_density(ncomp=1)[0][:, :, 4]            # [z, y] at x = 4
        up = rho[nz // 2:, :]                                            # upper interface: rho falls through the level
        k = np.argmax(up < level, axis=0)                                # first cell below the level
        a, b = up[k - 1, np.arange(ny)], up[k, np.arange(ny)]
        return nz // 2 + (k - 1) + (level - a) / (b - a)
    step1, step2, every = int(500000 * scale), int(800000 * scale), 2000
    lbm.LBM_timestep(step1)
    H = []
    for s in range(step1, step2 + 1, every):
        H.append(heights())
        lbm.LBM_timestep(every)
    H = np.array(H)
    h = H - H.mean(axis=0)
    hk2 = np.mean(np.abs(np.fft.fft(h, axis=1)) ** 2, axis=0)
    k = 2 * np.pi * np.fft.fftfreq(ny)
    print("frames %d  mean height %.4f  elapsed %.1f s" % (len(H), H.mean(), time.time() - t0))
    for m in range(1, 13):
        th_nb = kBT / (gamma * k[m] ** 2)                                # what the notebook plots
        th_eq = ny * kBT / (gamma * k[m] ** 2 * nx)                      # equipartition for numpy's unnormalised DFT, area nx*ny
        print("  mode %2d  <|h_k|^2> = %.4e   notebook line %.4e (ratio %.3f)   N_y kBT/(gamma k^2 L_x) %.4e (ratio %.3f)" % (m, hk2[m], th_nb, hk2[m] / th_nb, th_eq, hk2[m] / th_eq))
