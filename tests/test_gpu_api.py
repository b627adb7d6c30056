"""GPU: behaviour of the C-ABI beyond the arithmetic -- error conventions, parameter edits between
steps (the reference edits source-level globals between runs), step protocol guards."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_parameter_edit_between_steps(pkg, ob):
    """ReadMe.ipynb workflow: relax at kBT=0, then switch noise on and keep going (same populations)."""
    n = (8, 8, 8)
    lbm = pkg.BinaryLBM(*n, params=pkg.default_params(alpha0=2.0))
    ref = ob.OracleLattice(*n, params=ob.default_params(alpha0=2.0))
    lbm.LBM_init_droplet(0.3)
    ref.init_droplet(0.3)
    lbm.LBM_timestep(3)
    for _ in range(3):
        ref.timestep()
    lbm.set_params(kBT=2e-5, alpha0=1.5, tau_f=0.7)
    ref.p.kBT, ref.p.alpha0, ref.p.tau_f = 2e-5, 1.5, 0.7
    ref.refresh()                      # noise/hydrovs of the current state under the new parameters
    assert np.array_equal(lbm.LBM_hydrovars(), ref.h)
    lbm.LBM_timestep(3)
    for _ in range(3):
        ref.timestep()
    f, g = lbm.populations()
    assert np.array_equal(f, ref.f) and np.array_equal(g, ref.g)
    assert lbm.steps_done == 6
    lbm.close()


def test_error_conventions(pkg):
    lib = pkg._lib.load()
    lbm = pkg.BinaryLBM(8, 8, 8)
    lbm.LBM_init_mixture()
    with pytest.raises(pkg.BflbmError):
        lbm.LBM_timestep(-1)
    lbm.step_boundary()
    with pytest.raises(pkg.BflbmError, match="already open"):
        lbm.step_boundary()
    with pytest.raises(pkg.BflbmError, match="open step"):
        lbm.populations()
    with pytest.raises(pkg.BflbmError, match="open step"):
        lbm.LBM_hydrovars()
    lbm.step_interior()
    lbm.step_finish()
    with pytest.raises(pkg.BflbmError, match="no open step"):
        lbm.step_finish()
    with pytest.raises(pkg.BflbmError, match="single slab"):
        lbm.halo_pack(0, 0, 1)
    # bad FAB: valid region outside the allocated box
    fab = pkg.make_fab((0, 0, 0), (7, 7, 7), (0, 0, 0), (8, 7, 7))
    out = np.zeros((9, 8, 8, 8))
    with pytest.raises(pkg.BflbmError, match="valid region"):
        lbm.LBM_hydrovars_density(out, fab)
    with pytest.raises(pkg.BflbmError):
        lbm.set_schedule(7)
    lbm.set_schedule("auto")
    with pytest.raises(TypeError):
        lbm.upload(np.zeros((19, 8, 8, 8), dtype=np.float32), np.zeros((19, 8, 8, 8)))
    assert lbm.steps_done == 1
    lbm.close()
    # multi-slab context: several steps per call are refused (a halo exchange is needed in between)
    slab = pkg.BinaryLBM(8, 8, 8, z0=0, z1=4, rank=0, nranks=2)
    slab.LBM_init_mixture()
    with pytest.raises(pkg.BflbmError, match="halo exchange"):
        slab.LBM_timestep(2)
    assert slab.halo_bytes() == 38 * 8 * 8 * 8
    slab.close()


def test_partial_box_download_only_touches_its_cells(pkg, ob):
    """A FAB covering only part of the lattice (and sticking out of it) gets only its overlap."""
    n = (8, 8, 8)
    lbm = pkg.BinaryLBM(*n)
    ref = ob.OracleLattice(*n)
    lbm.LBM_init_droplet(0.3)
    ref.init_droplet(0.3)
    lo, hi = (4, -2, 2), (9, 5, 6)                 # sticks out in x (hi) and y (lo)
    shp = (9, hi[2] - lo[2] + 1, hi[1] - lo[1] + 1, hi[0] - lo[0] + 1)
    out = np.full(shp, -1.0)
    lbm.LBM_hydrovars_density(out, pkg.make_fab(lo, hi))
    exp = np.full(shp, -1.0)
    exp[:, :, 2:, :4] = ref.hbar[:9, 2:7, 0:6, 4:8]
    assert np.array_equal(out, exp)
    lbm.close()


def test_device_bytes_and_timer(pkg):
    lbm = pkg.BinaryLBM(32, 32, 32)
    assert lbm.device_bytes() >= 2 * 38 * 32 ** 3 * 8
    lbm.LBM_init_stripe(0.5)
    lbm.timer_start()
    lbm.LBM_timestep(10)
    assert lbm.timer_stop() > 0.0
    assert lbm.debug_time_kernel(0, 3) > 0.0
    lbm.close()
