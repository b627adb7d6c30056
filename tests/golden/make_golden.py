#!/usr/bin/env python3
"""Writes tests/golden/oracle_trajectories.npz from the CPU oracle.

The reference itself cannot be run here (it needs AMReX, which is absent, and stand-in headers are
not allowed), so these vectors are NOT reference outputs: they freeze the oracle at the commit where
it reproduced the outputs the reference's authors recorded in their notebooks (Flat_Interface.ipynb cell 4,
Surface_Tension.ipynb cells 13-19, Droplet_Fluctuation.ipynb cell 5; see oracle/bflbm_oracle.c).  The GPU path is compared with
them on the GPU box, where /root/reference does not exist either.
Compiler: gcc 11.4 -O3 -ffp-contract=off (oracle/Makefile).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402

out = {}
for name, n, steps, init in [("stripe", (8, 8, 8), 10, ("stripe", 0.5)),
                             ("droplet", (12, 10, 8), 5, ("droplet", 0.3)),
                             ("mixture", (6, 6, 6), 3, ("mixture",))]:
    ref = ob.OracleLattice(*n)
    getattr(ref, "init_" + init[0])(*init[1:])
    for _ in range(steps):
        ref.timestep()
    key = f"{name}-{'x'.join(map(str, n))}-{steps}"
    out[key + "_f"] = ref.f
    out[key + "_g"] = ref.g
    out[key + "_h"] = ref.h
np.savez_compressed(os.path.join(HERE, "oracle_trajectories.npz"), **out)
print("wrote", sorted(out))
