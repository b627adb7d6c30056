#!/usr/bin/env python3
"""Per-slab cost of the decomposed path on ONE GPU: RingLBM with k slabs of NX x NY x NZ each (peer copies are
device copies here) against a single context of one slab's size.
usage: tools/ring_bench.py [NX NY NZ] [steps] [k,k,...]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
a = sys.argv[1:]
nx, ny, nz = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (256, 256, 256)
steps = int(a[3]) if len(a) > 3 else 50
ks = [int(v) for v in a[4].split(",")] if len(a) > 4 else [1, 2, 4]
out = {"slab": f"{nx}x{ny}x{nz}"}
for k in ks:
    r = pkg.RingLBM(nx, ny, nz * k, nslabs=k)
    r.LBM_init_stripe(0.5)
    r.LBM_timestep(5); r.sync()
    t0 = time.perf_counter(); r.LBM_timestep(steps); r.sync(); dt = time.perf_counter() - t0
    out["slabs_%d" % k] = {"ms_per_step": round(dt / steps * 1e3, 4), "ms_per_slab_step": round(dt / steps / k * 1e3, 4),
                           "MLUPS": round(float(nx) * ny * nz * k * steps / dt / 1e6, 1)}
    r.close()
base = out.get("slabs_1", {}).get("ms_per_slab_step")
if base:
    for k in ks:
        out["slabs_%d" % k]["overhead_vs_undecomposed"] = round(out["slabs_%d" % k]["ms_per_slab_step"] / base - 1, 4)
print(json.dumps(out))
