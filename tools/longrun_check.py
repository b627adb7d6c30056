import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
n = 256
t0 = time.time()
lbm = pkg.BinaryLBM(n, n, n, params=pkg.default_params(kBT=1e-5, alpha0=0.0, tau_f=1.0, tau_g=1.0))
print("schedule:", lbm.resolved_schedule())
lbm.LBM_init_mixture()
m0 = lbm.mass()
lbm.LBM_timestep(3000)
m1 = lbm.mass()
hb = lbm.LBM_hydrovars_density()
rho = hb[0]
print("mass drift rel", abs(m1[0]-m0[0])/m0[0], abs(m1[1]-m0[1])/m0[1], "finite", np.isfinite(hb).all())
print("<drho^2> / (rho kBT/cs2) =", rho.var() / (1.0 * 1e-5 * 3.0), " <ux^2>/(kBT/rho) =", (hb[2]**2).mean() / 1e-5, "elapsed", time.time()-t0)
lbm.close()
z = pkg.BinaryLBM(512, 512, 512)
print("schedule:", z.resolved_schedule())
z.LBM_init_droplet(0.2)
m0 = z.mass(); z.LBM_timestep(1500); m1 = z.mass()
print("512^3 1500 steps: mass drift rel", abs(m1[0]-m0[0])/m0[0], abs(m1[1]-m0[1])/m0[1], "com", z.update_com())
z.close()
