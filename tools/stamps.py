#!/usr/bin/env python3
"""Phase shares of the fused march loop from a -DBFLBM_STAMP diagnostic build (never quote its run time)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lbm = pkg.BinaryLBM(n, n, n)
lbm.LBM_init_stripe(0.5)
lbm.LBM_timestep(5); lbm.sync()
lib = pkg._lib.load()
buf = np.zeros((8192 * 16, 8), dtype=np.uint64)
lib.bflbm_debug_stamps.restype = ctypes.c_int
k = lib.bflbm_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.shape[0])
b = buf[:k].astype(np.float64)
b = b[b[:, 6] > 0]
its = b[:, 6]
names = ["issue loads", "wait data + sums + LDS write", "barrier", "collide + stores", "-", "loop top"]
tot = b[:, :6].sum(1)
print(f"waves {len(b)}, iterations/wave {its.mean():.1f}, cycles/iteration {np.mean(tot / its):.0f}")
for i, nm in enumerate(names):
    if nm == "-": continue
    print(f"  {nm:32s} {np.mean(b[:, i] / its):9.0f} cycles/iter  {100 * b[:, i].sum() / tot.sum():5.1f} %")
