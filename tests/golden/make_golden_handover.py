#!/usr/bin/env python3
"""Writes tests/golden/oracle_handover_trajectories.npz: oracle trajectories on the two smallest lattices the
hand-over schedule (3, csrc/bflbm_handover.h) accepts with distinct neighbour tiles in both directions --
128 x 8 x 8 (2 x 2 tiles of 64 x 4) and 192 x 12 x 10 (3 x 3 tiles) -- so that the DEFAULT kernel is checked
against committed data and not only against another GPU schedule.

NOT reference outputs (the reference cannot be built here): they freeze the CPU oracle (oracle/bflbm_oracle.c, gcc
-O3 -ffp-contract=off) at the commit where it reproduced the reference authors' notebook numbers.  Stored per case:
  h/<steps>      hydrovs comps 0..8 (rho, phi, uf, rho+phi, ug) after 10 and 50 steps (the droplet: 50 only), float64
                 (tolerance checks)
  fg1            SHA-256 of f and g after step 1 (the first step of schedule 3 pulls its ring: bit-exact)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402
from make_golden_v2 import digest  # noqa: E402

PATH = os.path.join(HERE, "oracle_handover_trajectories.npz")
CASES = {
    "stripe_128x8x8": dict(shape=(128, 8, 8), init=("stripe", 0.5), par={}),
    "droplet_192x12x10": dict(shape=(192, 12, 10), init=("droplet", 0.485), par=dict(alpha0=2.0, kappa=2.0), steps=(50,)),
    "noise_128x8x8": dict(shape=(128, 8, 8), init=("mixture",), par=dict(kBT=1e-5, alpha0=0.0)),
}
STEPS = (10, 50)


def load():
    return {k.replace("__", "/"): v for k, v in np.load(PATH).items()}


def build():
    out = {}
    for name, c in CASES.items():
        ref = ob.OracleLattice(*c["shape"], params=ob.default_params(**c["par"]))
        getattr(ref, "init_" + c["init"][0])(*c["init"][1:])
        ref.timestep()
        out[f"{name}/fg1"] = np.concatenate([digest(ref.f), digest(ref.g)])
        done = 1
        for steps in c.get("steps", STEPS):
            while done < steps:
                ref.timestep(); done += 1
            out[f"{name}/h/{steps}"] = ref.h[:9].copy()
    return out


if __name__ == "__main__":
    o = build()
    np.savez_compressed(PATH, **{k.replace("/", "__"): v for k, v in o.items()})
    print(PATH, os.path.getsize(PATH), "bytes,", len(o), "entries")
