"""GPU: partial relaxation (tau != 1/2).  No number recorded by the reference's authors covers it (all their runs
use tau = 1/2, full relaxation), so beyond bit-equality with the oracle (test_gpu_parity.py::
test_other_parameter_sets) the GPU path is held to two closed-form consequences of LBM_binary.H:504-511
(tests/relaxation_cases.py)."""
import numpy as np
import pytest

import relaxation_cases as rc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("schedule", ["two_pass", "fused"])
@pytest.mark.parametrize("tau_f,tau_g", [(0.8, 0.6), (1.0, 1.0)])
def test_partial_relaxation_rate_of_every_mode(pkg, ob, schedule, tau_f, tau_g):
    n = 8
    for a in range(4, 19):
        lbm = pkg.BinaryLBM(n, n, n, params=pkg.default_params(tau_f=tau_f, tau_g=tau_g, alpha0=2.0), schedule=schedule)
        f0, g0 = rc.uniform_mode_state(ob, n, a)
        lbm.LBM_init(f0, g0)
        prev = (rc.moment(ob, f0[:, 1, 2, 3], a), rc.moment(ob, g0[:, 1, 2, 3], a))
        for _ in range(3):
            lbm.LBM_timestep(1)
            f, g = lbm.populations()
            cur = (rc.moment(ob, f[:, 1, 2, 3], a), rc.moment(ob, g[:, 1, 2, 3], a))
            assert abs(cur[0] / prev[0] - (1.0 - 1.0 / (tau_f + 0.5))) < 1e-10, (a, cur, prev)
            assert abs(cur[1] / prev[1] - (1.0 - 1.0 / (tau_g + 0.5))) < 1e-10, (a, cur, prev)
            assert np.array_equal(f, np.broadcast_to(f[:, :1, :1, :1], f.shape))      # stays uniform
            prev = cur
        lbm.close()


@pytest.mark.parametrize("tau", [0.5, 0.8, 1.0])
def test_shear_wave_decays_with_viscosity_cs2_tau(pkg, ob, tau):
    nx, ny, nz = 64, 8, 8
    lbm = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(tau_f=tau, tau_g=tau, alpha0=0.0))
    lbm.LBM_init(*rc.shear_wave_state(ob, nx, ny, nz))
    lbm.LBM_timestep(100)
    a1 = rc.shear_amplitude(ob, *lbm.populations())
    lbm.LBM_timestep(200)
    a2 = rc.shear_amplitude(ob, *lbm.populations())
    lbm.close()
    nu = -np.log(a2 / a1) / ((2 * np.pi / nx) ** 2 * 200)
    assert abs(nu / (tau / 3.0) - 1.0) < 5e-3, nu
