#!/usr/bin/env python3
"""Where a march position of the hand-over kernel spends its time: shader-clock stamps at the phase boundaries, written
by a diagnostic build of the library (-DBFLBM_STAMP -> csrc/build/libbflbm_stamp.so, never the product).

  BFLBM_LIB=binary-fluctuating-lattice-boltzmann_amd/csrc/build/libbflbm_stamp.so python tools/ho_stamps.py [--size 512] [--noise]

Prints, per phase, the median duration in shader clocks over 64 steady-state positions x 4 waves of one workgroup, and the
position time."""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=512); ap.add_argument("--noise", action="store_true")
a = ap.parse_args()
pkg = ge.load_package()
lib = pkg._lib.load()
par = dict(kBT=1e-5, alpha0=0.0) if a.noise else {}
n = a.size
lbm = pkg.BinaryLBM(n, n, n, params=pkg.default_params(**par), schedule="handover")
lbm.LBM_init_stripe(0.5)
lbm.LBM_timestep(4); lbm.sync()
NST, NPOS = 10, 64
buf = (ctypes.c_ulonglong * (4 * NPOS * NST))()
lib.bflbm_debug_ho_stamps.restype = ctypes.c_int
assert lib.bflbm_debug_ho_stamps(buf, 4 * NPOS * NST) == 0
t = np.array(list(buf), dtype=np.float64).reshape(NPOS, 4, NST)
names = ["top -> plane arrived, densities summed", "barrier", "frames finished, hold swapped (76 LDS ops), moments", "request f half of q+1",
         "gradient, (noise head,) projection", "relax f", "store f, produce frames, request g half", "relax g", "store g, produce frames"]
d = np.diff(t, axis=2)
pos = t[1:, :, 0] - t[:-1, :, 0]
print(f"{n}^3 {'noise' if a.noise else 'quiet'}: position {np.median(pos):8.0f} clocks (min {pos.min():.0f}, max {pos.max():.0f}); between positions (store g -> next top) {np.median(t[1:, :, 0] - t[:-1, :, 9]):6.0f}")
for k, nm in enumerate(names):
    print(f"  {np.median(d[:, :, k]):8.0f}  {100 * np.median(d[:, :, k]) / np.median(pos):5.1f} %   {nm}")
lbm.close()
