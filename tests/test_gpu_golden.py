"""GPU: the HIP path against the committed fixture list of SURVEY.md 8c (tests/golden/oracle_golden_v2.npz,
written by tests/golden/make_golden_v2.py from the oracle; full arrays for 8^3, SHA-256 digests for 12^3 and
16^3).  Bit-exact; both schedules."""
import hashlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


def _digest(a):
    import make_golden_v2 as mg
    return mg.digest(a)


def _check(stored, arr, key):
    want = stored[key]
    got = _digest(arr) if want.dtype == np.uint8 else arr
    assert np.array_equal(got, want), key


@pytest.fixture(scope="module")
def golden():
    import make_golden_v2 as mg
    return mg, mg.load()


@pytest.mark.parametrize("schedule", ["two_pass", "fused"])
@pytest.mark.parametrize("n", [8, 12, 16])
def test_trajectories(pkg, golden, n, schedule):
    mg, g = golden
    for name, init in mg.INITS.items():
        lbm = pkg.BinaryLBM(n, n, n, schedule=schedule)
        getattr(lbm, "LBM_init_" + init[0])(*init[1:])
        done = 0
        for steps in (1, 3, 10, 100):
            lbm.LBM_timestep(steps - done); done = steps
            key = f"traj/{name}/{n}/{steps}/"
            f, gg = lbm.populations()
            _check(g, f, key + "f"); _check(g, gg, key + "g")
            _check(g, lbm.LBM_hydrovars_density(), key + "hbar")
            _check(g, lbm.LBM_hydrovars(), key + "h")
        lbm.close()


@pytest.mark.parametrize("schedule", ["two_pass", "fused"])
def test_injected_and_generated_noise(pkg, golden, schedule):
    mg, g = golden
    lbm = pkg.BinaryLBM(8, 8, 8, params=pkg.default_params(**mg.NOISE_PAR), schedule=schedule)
    lbm.LBM_init_droplet(0.3)
    for step in range(3):
        lbm.inject_noise(g[f"noise/injected/{step}/fn"], g[f"noise/injected/{step}/gn"])
        lbm.LBM_timestep(1)
        f, gg = lbm.populations()
        _check(g, f, f"noise/injected/{step}/f"); _check(g, gg, f"noise/injected/{step}/g")
        _check(g, lbm.LBM_hydrovars_density(), f"noise/injected/{step}/hbar")
    lbm.close()
    lbm = pkg.BinaryLBM(12, 12, 12, params=pkg.default_params(**mg.NOISE_PAR), schedule=schedule)
    lbm.LBM_init_droplet(0.3)
    for step in range(1, 4):
        lbm.LBM_timestep(1)
        f, gg = lbm.populations()
        fn, gn = lbm.thermal_noise()
        for nm, arr in (("f", f), ("g", gg), ("fn", fn), ("gn", gn), ("h", lbm.LBM_hydrovars())):
            _check(g, arr, f"noise/generated/{step}/{nm}")
    lbm.close()


@pytest.mark.parametrize("schedule", ["two_pass", "fused"])
def test_random_state_units(pkg, golden, schedule):
    """hydrovars (gradient + projection) and one collide+stream of random asymmetric states."""
    mg, g = golden
    for tag, par in mg.UNIT_PARS:
        lbm = pkg.BinaryLBM(4, 5, 6, params=pkg.default_params(**par), schedule=schedule)
        lbm.LBM_init(g[f"unit/state/{tag}/f0"], g[f"unit/state/{tag}/g0"])
        _check(g, lbm.LBM_hydrovars_density(), f"unit/state/{tag}/hbar")
        _check(g, lbm.LBM_hydrovars(), f"unit/state/{tag}/h")
        lbm.LBM_timestep(1)
        f, gg = lbm.populations()
        _check(g, f, f"unit/state/{tag}/f1"); _check(g, gg, f"unit/state/{tag}/g1")
        lbm.close()
