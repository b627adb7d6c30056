"""GPU: BASELINE.json's full sizes, through size-independent properties (the oracle cannot run
256^3/512^3 in seconds): schedule equivalence, translation invariance of the stripe, mass
conservation, golden fixtures, upload/download round trip."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_golden_fixtures(pkg):
    """tests/golden/oracle_trajectories.npz (written by tests/golden/make_golden.py)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_trajectories.npz"))
    for key in [k[:-2] for k in g.files if k.endswith("_f")]:
        name, n, steps = key.split("-")
        n = tuple(int(v) for v in n.split("x"))
        for schedule in ("two_pass", "fused"):
            lbm = pkg.BinaryLBM(*n, schedule=schedule)
            {"stripe": lambda: lbm.LBM_init_stripe(0.5), "droplet": lambda: lbm.LBM_init_droplet(0.3),
             "mixture": lbm.LBM_init_mixture}[name]()
            lbm.LBM_timestep(int(steps))
            f, gg = lbm.populations()
            assert np.array_equal(f, g[key + "_f"]) and np.array_equal(gg, g[key + "_g"]), (key, schedule)
            assert np.array_equal(lbm.LBM_hydrovars(), g[key + "_h"]), (key, schedule)
            lbm.close()


def test_256_cubed_schedules_agree_and_conserve_mass(pkg):
    """configs[1]: 256^3, zero noise.  The fused and the two-pass schedule are different kernels
    with different tilings; identical doubles after 20 steps is a checksum over all indexing."""
    n = 256
    res = {}
    for schedule in ("two_pass", "fused"):
        lbm = pkg.BinaryLBM(n, n, n, schedule=schedule)
        lbm.LBM_init_droplet(0.2)
        m0 = lbm.mass()
        lbm.LBM_timestep(20)
        m1 = lbm.mass()
        assert abs(m1[0] - m0[0]) <= 1e-12 * m0[0] and abs(m1[1] - m0[1]) <= 1e-12 * m0[1]
        res[schedule] = lbm.LBM_hydrovars_density()
        com = lbm.update_com()
        np.testing.assert_allclose(com, [n / 2, n / 2, n / 2], atol=1e-6)    # droplet centred at box/2
        lbm.close()
    assert np.array_equal(res["two_pass"], res["fused"])


def test_256_cubed_stripe_is_translation_invariant(pkg):
    """Flat interface: the state depends on z only, so after any number of steps every (x,y) column
    must hold bitwise the same doubles -- any mis-indexed tile, wrap or halo breaks this."""
    n = 256
    lbm = pkg.BinaryLBM(n, n, n, schedule="fused")     # bitwise statement: the exact schedule (auto = hand-over sums tile-edge rings in another order)
    lbm.LBM_init_stripe(0.5)
    lbm.LBM_timestep(25)
    hb = lbm.LBM_hydrovars_density()
    for comp in range(9):
        col = hb[comp, :, 0, 0]
        assert np.array_equal(hb[comp], np.broadcast_to(col[:, None, None], hb[comp].shape)), comp
    # no flow in x, y beyond the rounding residue of ((a-b)+b)-a in the gradient stencil
    assert np.abs(hb[2]).max() < 1e-14 and np.abs(hb[3]).max() < 1e-14
    # z-mirror symmetry of the stripe about the box centre (profile is even in pos = z - nz/2)
    rho = hb[0, :, 0, 0]
    np.testing.assert_allclose(rho[1:], rho[1:][::-1], rtol=0, atol=1e-13)
    lbm.close()


def test_512_cubed_step_runs_and_conserves_mass(pkg):
    """North-star size on one GPU (82 GB resident): droplet init, a few steps, mass and symmetry."""
    n = 512
    lbm = pkg.BinaryLBM(n, n, n)
    assert lbm.device_bytes() > 80e9
    lbm.LBM_init_stripe(0.5)
    m0 = lbm.mass()
    lbm.LBM_timestep(6)
    m1 = lbm.mass()
    assert abs(m1[0] - m0[0]) <= 1e-12 * m0[0] and abs(m1[1] - m0[1]) <= 1e-12 * m0[1]
    assert abs(m0[0] + m0[1] - n ** 3) <= 1e-9 * n ** 3
    lbm.close()


def test_upload_download_roundtrip_128(pkg):
    n = 128
    rng = np.random.default_rng(4)
    f0 = rng.random((19, n, n, n))
    g0 = rng.random((19, n, n, n))
    lbm = pkg.BinaryLBM(n, n, n)
    lbm.LBM_init(f0, g0)
    f, g = lbm.populations()
    assert np.array_equal(f, f0) and np.array_equal(g, g0)
    hb = lbm.LBM_hydrovars_density()
    acc = np.zeros((n, n, n))
    for i in range(19):
        acc += f0[i]
    assert np.array_equal(hb[0], acc)              # sequential sum in index order (LBM_binary.H:322-328)
    lbm.close()


def test_droplet_notebook_centre_of_mass_after_20000_steps(pkg):
    """Droplet_Fluctuation.ipynb cell 5 (run of the reference itself, 2025-12-29): 64^3 droplet, header
    defaults (alpha0=4, kBT=0, rho_hi=1, rho_lo=0, kappa=4, tau=1/2), r_init=0.2, frame 20000 ->
    'Center of Mass: [0.50470332 0.50470332 0.50470332]' with cell-centred coordinates (i+0.5)/n weighted
    by the 'rho' field.  An integrated observable of a 20000-step trajectory of the real reference; we
    require all printed digits."""
    n = 64
    lbm = pkg.BinaryLBM(n, n, n)
    lbm.LBM_init_droplet(0.2)
    lbm.LBM_timestep(20000)
    rho = lbm.LBM_hydrovars(ncomp=1)[0]
    c = (np.arange(n) + 0.5) / n
    m = rho.sum()
    com = np.array([(rho * c[None, None, :]).sum(), (rho * c[None, :, None]).sum(), (rho * c[:, None, None]).sum()]) / m
    assert ["%.8f" % v for v in com] == ["0.50470332"] * 3, com
    # same number from the device-side reduction behind update_com (cell indices, LBM_hydrovs.H:46-48)
    np.testing.assert_allclose((lbm.update_com() + 0.5) / n, com, rtol=1e-12)
    r0, p0 = lbm.mass()
    assert abs(r0 - 9313.703607) < 1e-5                      # rho mass is conserved from the initial profile
    # the other numbers of that notebook cell, with the notebook's definitions (analysis.py):
    #   Fitted Droplet Radius: 0.1669310108163054
    #   Eigenvalues: [0.03491747 0.03496081 0.03491747]   Principal Axes (a, b, c): [0.1668965  0.16700005 0.1668965 ]
    an = pkg.analysis
    rho_xyz = np.ascontiguousarray(rho.transpose(2, 1, 0))
    assert ["%.8f" % v for v in an.centre_of_mass(rho_xyz)] == ["0.50470332"] * 3
    hi, lo, R, W = an.fit_droplet(rho_xyz)
    assert abs(R - 0.1669310108163054) < 2e-7, R             # least-squares optimum; scipy versions differ in the last digits
    axes, ev, _ = an.principal_axes(rho_xyz, R)
    assert np.allclose(np.sort(ev), np.sort([0.03491747, 0.03496081, 0.03491747]), rtol=0, atol=6e-9), ev
    assert np.allclose(np.sort(axes), np.sort([0.1668965, 0.16700005, 0.1668965]), rtol=0, atol=3e-7), axes
    # ... and the same notebook numbers without any field leaving the GPU (csrc/bflbm_droplet.h)
    mom = lbm.droplet_moments()
    assert ["%.8f" % v for v in an.com_from_moments(mom, (n, n, n))] == ["0.50470332"] * 3
    hi_d, lo_d, R_d, W_d = lbm.fit_droplet()
    assert abs(R_d - 0.1669310108163054) < 2e-7, R_d
    axes_d, ev_d, _ = an.principal_axes_from_moments(mom, (n, n, n), R_d)
    assert np.allclose(np.sort(ev_d), np.sort([0.03491747, 0.03496081, 0.03491747]), rtol=0, atol=6e-9), ev_d
    assert np.allclose(np.sort(axes_d), np.sort([0.1668965, 0.16700005, 0.1668965]), rtol=0, atol=3e-7), axes_d
    lbm.close()


def test_config5_slab_shape_1024x1024(pkg):
    """configs[4]: 1024 x 1024 planes (the weak-scaling slab of the 8-GPU case, shortened in z).  2048 tile
    columns, 8 MB planes: the fused kernel, the two-pass kernels and a ring of two slabs with halo exchange
    must produce identical doubles; mass is conserved."""
    nx, ny, nz = 1024, 1024, 12
    res = {}
    for schedule in ("two_pass", "fused"):
        lbm = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(alpha0=2.0), schedule=schedule)
        lbm.LBM_init_droplet(0.01)                 # radius 10 cells around (512, 512, 512 -> wrapped z)
        m0 = lbm.mass()
        lbm.LBM_timestep(4)
        m1 = lbm.mass()
        assert abs(m1[0] - m0[0]) <= 1e-12 * m0[0] and abs(m1[1] - m0[1]) <= 1e-12 * m0[1]
        res[schedule] = lbm.LBM_hydrovars_density()
        lbm.close()
    assert np.array_equal(res["two_pass"], res["fused"])
    ring = pkg.RingLBM(nx, ny, nz, nslabs=2, params=pkg.default_params(alpha0=2.0), schedule="fused")
    ring.LBM_init_droplet(0.01)
    ring.LBM_timestep(4)
    assert np.array_equal(ring.LBM_hydrovars_density(), res["fused"])
    ring.close()


def test_flat_interface_notebook_height_full_box(pkg):
    """Flat_Interface.ipynb cell 4 (run of the reference, 2025-11-14): the reference's flat-interface box
    8 x 256 x 64 (main_run_job.cpp:128), alpha0 = 1.5, frame 2000 -> the rho = 1.05 contour sits at
    47.86628666 for every (x, y) (parameters not printed in the notebook: kappa = 0.1, rho_lo = 0.1,
    rho_hi = 3.0, see tests/test_oracle_pins.py::test_flat_interface_notebook_height)."""
    lbm = pkg.BinaryLBM(8, 256, 64, params=pkg.default_params(alpha0=1.5, kappa=0.1, rho_lo=0.1, rho_hi=3.0))
    lbm.LBM_init_stripe(0.5)
    lbm.LBM_timestep(2000)
    rho = lbm.LBM_hydrovars(ncomp=1)[0]                    # [z, y, x]
    a, b = rho[47], rho[48]
    height = 47 + (1.05 - a) / (b - a)
    assert np.all((a - 1.05) * (b - 1.05) < 0)
    assert {"%.8f" % v for v in height.ravel()} == {"47.86628666"}
    assert np.all(rho == rho[:, :1, :1])                   # uniform in x and y, bit for bit
    lbm.close()
