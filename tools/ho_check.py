#!/usr/bin/env python3
"""GPU check of the density hand-over schedule against the bit-exact fused schedule (same device functions,
ring densities pulled): after n steps the populations must agree to rounding, and the frames must be in use."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()

def run(shape, sched, steps, init):
    with pkg.BinaryLBM(*shape, schedule=sched) as l:
        getattr(l, "LBM_init_" + init[0])(*init[1:])
        l.LBM_timestep(steps)
        f, g = l.populations()
    return f, g

for shape, init in [((128, 8, 16), ("stripe", 0.5)), ((128, 16, 12), ("droplet", 0.3)), ((192, 24, 40), ("droplet", 0.25)), ((256, 64, 64), ("droplet", 0.2))]:
    for steps in (1, 2, 3, 10, 40):
        fe, ge_ = run(shape, "fused_exact", steps, init)
        fh, gh = run(shape, "handover", steps, init)
        d = max(np.abs(fe - fh).max(), np.abs(ge_ - gh).max())
        nd = int((fe != fh).sum() + (ge_ != gh).sum())
        print(f"{shape} {init[0]} steps {steps:3d}: max|diff| {d:.3e}  differing doubles {nd} of {fe.size*2}", flush=True)
        assert d < 1e-13, "hand-over schedule deviates"
print("ok")
