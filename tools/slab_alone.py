#!/usr/bin/env python3
"""What one rank of an N > 1 run computes per step, alone on the GPU: a 512x512x512 (or NX NY NZ) slab of a decomposed lattice stepped without
its neighbours (boundary pairs + interior sweep, halo planes left as initialised) against the undecomposed box of the same size.
usage: slab_alone.py [NX NY NZ]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
a = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else [512, 512, 512]
nx, ny, nz = a
for tag, kw in (("undecomposed", {}), ("slab of a 2-slab lattice", dict(z0=0, z1=nz, rank=0, nranks=2))):
    l = pkg.BinaryLBM(nx, ny, nz * (2 if kw else 1), **kw)
    l.LBM_init_stripe(0.5)
    def step(k):
        for _ in range(k):
            l.step_boundary(); l.step_interior(); l.step_finish()
    step(5); l.sync()
    out = []
    for _ in range(3):
        l.timer_start(); step(20); out.append(round(l.timer_stop() / 20, 4))
    print(f"{nx}x{ny}x{nz} {tag:26s} ms/step {out}  MLUPS {nx * ny * nz / min(out) / 1e3:.0f}  schedule {l.resolved_schedule()}  placement {l.placement_report()}", flush=True)
    l.close()
