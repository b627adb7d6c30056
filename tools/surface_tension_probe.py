#!/usr/bin/env python3
"""Which parameters reproduce the densities Surface_Tension.ipynb prints (cells 13, 18)?  Exploration tool."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import __graft_entry__ as ge
import test_gpu_notebook_surface_tension as T
pkg = ge.load_package()
def run(alpha0, nc, r, kappa, want, **kw):
    lbm = pkg.BinaryLBM(32, 32, 32, params=pkg.default_params(rho_hi=3.0, alpha0=alpha0, kappa=kappa, **kw))
    lbm.LBM_init_droplet(r); lbm.LBM_timestep(20000)
    h = lbm.LBM_hydrovars(ncomp=2)
    got = np.array([h[0][nc, nc, 0], h[0][nc, nc, nc], h[1][nc, nc, 0], h[1][nc, nc, nc]])
    print("alpha0 %.1f r %.3f kappa %.2f %s: max rel err %.2e  rho_in %.12f want %.12f" % (alpha0, r, kappa, kw, (np.abs(got - want) / np.abs(want)).max(), got[1], want[1]))
    lbm.close()

for rec, rs in ((T.CELL18[1], (0.23, 0.2275)), (T.CELL18[2], (0.25,)), (T.CELL18[3], (0.28, 0.2775))):
    for r in rs:
        run(1.7, 15, r, 1.0, np.array(rec[1:5]))
