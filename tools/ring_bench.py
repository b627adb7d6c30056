#!/usr/bin/env python3
"""Per-slab cost of the decomposed path on ONE GPU: RingLBM with k slabs of S^3 each (peer copies are
device copies here) against a single S^3 context.  usage: tools/ring_bench.py [S] [steps]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
out = {}
for k in (1, 2, 4):
    r = pkg.RingLBM(S, S, S * k, nslabs=k)
    r.LBM_init_stripe(0.5)
    r.LBM_timestep(5); r.sync()
    t0 = time.perf_counter(); r.LBM_timestep(steps); r.sync(); dt = time.perf_counter() - t0
    out["slabs_%d" % k] = {"ms_per_step": round(dt / steps * 1e3, 4), "ms_per_slab_step": round(dt / steps / k * 1e3, 4),
                           "MLUPS": round(S ** 3 * k * steps / dt / 1e6, 1)}
    r.close()
print(json.dumps(out))
