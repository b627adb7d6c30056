#!/usr/bin/env python3
"""The hand-over schedule inside `auto`'s bound on the most violent states the bound admits: the demixing mixture (rho = phi = 1, kBT = 1e-5) at
alpha0 = 2.5 (interaction strength 5) and 3.0 (strength 6, the bound itself).  Every 20 steps: the masked metric of tests/tolerances.py and the
unmasked maxima of GPU (schedule 3) against the oracle, next to the oracle's own response to a one-ulp perturbation, and the state's extremes."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import oracle_binding as ob
import tolerances
pkg = ge.load_package()
ob.lib().orc_set_threads(16)
shape = (128, 28, 26)
for a0 in (2.5, 3.0):
    par = dict(kBT=1e-5, alpha0=a0)
    ref = ob.OracleLattice(*shape, params=ob.default_params(**par)); ref.init_mixture()
    per = ob.OracleLattice(*shape, params=ob.default_params(**par)); per.init_mixture()
    rng = np.random.default_rng(1)
    per.f *= 1.0 + rng.integers(-1, 2, per.f.shape) * 2.0 ** -52
    per.g *= 1.0 + rng.integers(-1, 2, per.g.shape) * 2.0 ** -52
    per.refresh("absolute")
    gpu = pkg.BinaryLBM(*shape, params=pkg.default_params(**par), schedule="handover"); gpu.LBM_init_mixture()
    auto = pkg.BinaryLBM(*shape, params=pkg.default_params(**par)); auto.LBM_init_mixture()
    print(f"alpha0 {a0}: auto resolves to {auto.resolved_schedule()} (total density {auto.state_total_max})", flush=True)
    auto.close()
    done = 0
    for steps in range(20, 201, 20):
        for _ in range(steps - done):
            ref.timestep(); per.timestep()
        gpu.LBM_timestep(steps - done); done = steps
        h = gpu.LBM_hydrovars()
        e, k = tolerances.errors(h, ref.h), tolerances.errors(per.h, ref.h)
        m = lambda d: max(d[x] for x in tolerances.MASKED)
        print(f"  step {steps:3d}: rho in [{ref.h[0].min():.3f}, {ref.h[0].max():.3f}] |u|max {np.abs(ref.h[2:5]).max():.2e} | masked GPU {m(e):.1e} oracle-1ulp {m(k):.1e}"
              f" | unmasked dens GPU {e['dens_elem_all']:.1e} 1ulp {k['dens_elem_all']:.1e}  vel/cs GPU {e['vel_abs_all']:.1e} 1ulp {k['vel_abs_all']:.1e}", flush=True)
    gpu.close()
