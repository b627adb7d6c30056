import sys, time
sys.path.insert(0, ".")
import __graft_entry__ as ge
pkg = ge.load_package()
lbm = pkg.BinaryLBM(256, 256, 256)
lbm.LBM_init_stripe(0.5)
lbm.sync()
for k in range(12):
    lbm.timer_start(); lbm.LBM_timestep(20); ms = lbm.timer_stop() / 20
    print("chunk %2d (steps %3d-%3d): %.4f ms/step  %.0f MLUPS" % (k, 20*k, 20*k+19, ms, 256**3 / ms / 1e3))
