#!/usr/bin/env python3
"""Writes tests/golden/oracle_golden_v2.npz: the fixture list of SURVEY.md 8c, from the CPU oracle.

Like make_golden.py these are NOT reference outputs (the reference cannot be built here): they freeze
the oracle at the commit where it reproduced the notebook outputs of the reference (oracle/bflbm_oracle.c).
  * trajectories  N in {8, 12, 16} x {stripe 0.5, droplet 0.3, mixture} x steps {1, 3, 10, 100}, kBT = 0:
    f, g, hydrovsbar[0..8], hydrovs[0..21] -- full arrays for N = 8, SHA-256 of the little-endian bytes for
    N = 12 and 16 (bit-exact comparisons need nothing more and the file stays small);
  * noise         8^3 droplet, kBT = 1e-5, three steps with INJECTED noise arrays (stored), full results;
                  and three steps with the project's generated stream (seed 12345), digests;
  * units         moments / populations of 64 random 19-vectors; hydrovars (gradient, projection) and one
                  collide+stream of a random 4x5x6 state at two parameter sets.
Compiler: gcc 11.4 -O3 -ffp-contract=off (oracle/Makefile).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob  # noqa: E402


def digest(a):
    # + 0.0 maps -0.0 to +0.0: the parity claim is equality of the doubles as numbers (np.array_equal); the
    # HIP path's shared-reciprocal division returns +0 for -0/b where IEEE division gives -0
    a = np.ascontiguousarray(a, dtype="<f8") + 0.0
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)


INITS = {"stripe": ("stripe", 0.5), "droplet": ("droplet", 0.3), "mixture": ("mixture",)}
NOISE_PAR = dict(kBT=1e-5, alpha0=2.0)
UNIT_PARS = (("default", {}), ("tau", dict(tau_f=0.8, tau_g=0.6, alpha0=2.5, kappa=3.0)))
PATH = os.path.join(HERE, "oracle_golden_v2.npz")


def load():
    return {k.replace("__", "/"): v for k, v in np.load(PATH).items()}


def build():
  out = {}
  for n in (8, 12, 16):
      for name, init in INITS.items():
          ref = ob.OracleLattice(n, n, n)
          getattr(ref, "init_" + init[0])(*init[1:])
          done = 0
          for steps in (1, 3, 10, 100):
              while done < steps:
                  ref.timestep(); done += 1
              key = f"traj/{name}/{n}/{steps}/"
              for nm, arr in (("f", ref.f), ("g", ref.g), ("hbar", ref.hbar[:9]), ("h", ref.h)):
                  out[key + nm] = arr.copy() if n == 8 else digest(arr)

  # noise: injected arrays
  rng = np.random.default_rng(2024)
  par = NOISE_PAR
  ref = ob.OracleLattice(8, 8, 8, ob.default_params(**par))
  ref.init_droplet(0.3)
  amp = np.sqrt(1e-5)
  for step in range(3):
      fn = amp * rng.standard_normal(ref.fn.shape); gn = amp * rng.standard_normal(ref.gn.shape)
      fn[0] = 0; gn[0] = 0; gn[1:4] = -fn[1:4]
      out[f"noise/injected/{step}/fn"] = fn; out[f"noise/injected/{step}/gn"] = gn
      ref.timestep_injected(fn, gn)
      out[f"noise/injected/{step}/f"] = ref.f.copy(); out[f"noise/injected/{step}/g"] = ref.g.copy()
      out[f"noise/injected/{step}/hbar"] = ref.hbar[:9].copy()
  # noise: generated stream
  ref = ob.OracleLattice(12, 12, 12, ob.default_params(**par))
  ref.init_droplet(0.3)
  for step in range(1, 4):
      ref.timestep()
      for nm, arr in (("f", ref.f), ("g", ref.g), ("fn", ref.fn), ("gn", ref.gn), ("h", ref.h)):
          out[f"noise/generated/{step}/{nm}"] = digest(arr)

  # units
  rng = np.random.default_rng(7)
  vecs = rng.random((64, 19))
  out["unit/moments/in"] = vecs
  out["unit/moments/out"] = np.array([ob.moments(v) for v in vecs])
  out["unit/populations/out"] = np.array([ob.populations(v) for v in vecs])
  for tag, p in UNIT_PARS:
      ref = ob.OracleLattice(4, 5, 6, ob.default_params(**p))
      f0 = 0.02 + 0.1 * rng.random(ref.f.shape); g0 = 0.02 + 0.1 * rng.random(ref.g.shape)
      out[f"unit/state/{tag}/f0"] = f0; out[f"unit/state/{tag}/g0"] = g0
      ref.init_from(f0, g0)
      out[f"unit/state/{tag}/hbar"] = ref.hbar[:9].copy(); out[f"unit/state/{tag}/h"] = ref.h.copy()
      ref.timestep()
      out[f"unit/state/{tag}/f1"] = ref.f.copy(); out[f"unit/state/{tag}/g1"] = ref.g.copy()
  return out


if __name__ == "__main__":
    out = build()
    np.savez_compressed(PATH, **{k.replace("/", "__"): v for k, v in out.items()})
    print("wrote", len(out), "entries")
