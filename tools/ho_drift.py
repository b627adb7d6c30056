#!/usr/bin/env python3
"""How far the hand-over schedule (3) drifts from the bit-exact schedule (1) over a LONG run, in the metric of
tests/tolerances.py: 128^3 and 256^3 droplets (header defaults, r = 0.2) after 100 ... 20000 steps.  The reference CPU path
shows the same kind of drift between an FMA and a non-FMA build (rho 1e-16 at 10 steps, 4e-14 at 1000)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import tolerances
pkg = ge.load_package()
for n, marks in ((128, (100, 1000, 5000, 20000)), (256, (100, 1000, 4000))):
    a = pkg.BinaryLBM(n, n, n, schedule="fused"); b = pkg.BinaryLBM(n, n, n, schedule="handover")
    a.LBM_init_droplet(0.2); b.LBM_init_droplet(0.2)
    done = 0
    for m in marks:
        a.LBM_timestep(m - done); b.LBM_timestep(m - done); done = m
        e = tolerances.errors(b.LBM_hydrovars(ncomp=9), a.LBM_hydrovars(ncomp=9))
        print(f"{n}^3 droplet, {m:6d} steps: " + "  ".join(f"{k} {v:.1e}" for k, v in e.items()), flush=True)
    a.close(); b.close()
