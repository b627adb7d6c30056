#!/usr/bin/env python3
"""Streaming calibration on the GPU box: what this device sustains for the path's access pattern."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lbm = pkg.BinaryLBM(n, n, n)
lbm.LBM_init_stripe(0.5)
sites = n ** 3
out = {}
for which, name, bytes_per_site in [(0, "pull_copy", 608), (1, "density", 320), (2, "memcpy_d2d", 608), (3, "pull_copy_2sites_per_thread", 608),
                                     (4, "pull_copy_row_interleaved_layout", 608)]:
    ms = lbm.debug_time_kernel(which, 20)
    out[name] = {"ms": round(ms, 4), "GBps": round(sites * bytes_per_site / ms / 1e6, 1)}
for sched in ("two_pass", "fused"):
    lbm.set_schedule(sched)
    lbm.LBM_timestep(5); lbm.sync()
    lbm.timer_start(); lbm.LBM_timestep(30); ms = lbm.timer_stop() / 30
    out["step_" + sched] = {"ms": round(ms, 4), "MLUPS": round(sites / ms / 1e3, 1), "GBps_alg": round(sites * 608 / ms / 1e6, 1)}
print(json.dumps(out))
