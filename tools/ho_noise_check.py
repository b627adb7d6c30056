#!/usr/bin/env python3
"""GPU check: the pipelined hand-over kernel with thermal noise against the two-pass schedule (same stream, same
amplitudes; the tile-ring densities are summed in another order, so agreement is to rounding, not bitwise)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
for shape in ((128, 16, 12), (256, 32, 40)):
    for steps in (1, 2, 10, 30):
        out = {}
        for sched in ("two_pass", "handover"):
            with pkg.BinaryLBM(*shape, params=pkg.default_params(kBT=1e-5, alpha0=1.0), schedule=sched) as l:
                l.LBM_init_droplet(0.25)
                l.LBM_timestep(steps)
                out[sched] = l.populations()
        d = max(np.abs(out["two_pass"][0] - out["handover"][0]).max(), np.abs(out["two_pass"][1] - out["handover"][1]).max())
        print(shape, steps, "max |diff| %.3e" % d, "noise present:", bool(np.abs(out["handover"][0] - out["handover"][0].mean(axis=(1, 2, 3), keepdims=True)).max() > 0), flush=True)
        assert d < 1e-13
print("ok")
