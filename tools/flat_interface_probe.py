#!/usr/bin/env python3
"""Which parameters reproduce Flat_Interface.ipynb cell 4 (height 47.86628666 of the rho = (rho_lo+rho_hi)/2
contour, 8x256x64, alpha0 = 1.5, rho_lo = 0.1, rho_hi = 2, frame 2000)?  Exploration tool."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()


def height(rho_z, level):
    """upper crossing of `level` along z by linear interpolation between cells (skimage find_contours)"""
    out = []
    for k in range(len(rho_z) - 1):
        a, b = rho_z[k], rho_z[k + 1]
        if (a - level) * (b - level) < 0:
            out.append(k + (level - a) / (b - a))
    return out


for kappa, tau, frac in itertools.product((0.1, 1.0, 3.0, 4.0), (0.5,), (0.5,)):
    lbm = pkg.BinaryLBM(8, 256, 64, params=pkg.default_params(alpha0=1.5, rho_lo=0.1, rho_hi=3.0, kappa=kappa, tau_f=tau, tau_g=tau))
    lbm.LBM_init_stripe(frac)
    lbm.LBM_timestep(2000)
    rho = lbm.LBM_hydrovars(ncomp=1)[0]          # [z, y, x]
    print("kappa %.1f tau %.1f frac %.2f: level 1.05 %s  level 1.55 %s" % (kappa, tau, frac, ["%.8f" % v for v in height(rho[:, 0, 0], 1.05)], ["%.8f" % v for v in height(rho[:, 0, 0], 1.55)]))
    lbm.close()
