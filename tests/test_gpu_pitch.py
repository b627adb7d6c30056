"""GPU: lattices whose rows are padded on the device (nx not divisible by 16 -> row pitch = nx rounded up to
16).  Every path that translates between the padded resident layout and dense arrays (upload,
download, observables, injected noise, reference state, reductions, structure factors, halo exchange) against
the oracle / the host twins, at nx = 20 (pitch 32), 37 (48) and 70 (80)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [(20, 6, 9), (37, 5, 8), (70, 9, 4)])
@pytest.mark.parametrize("schedule", ["two_pass", "fused"])
def test_padded_rows_all_paths(pkg, ob, n, schedule):
    nx, ny, nz = n
    par = dict(kBT=1e-5, alpha0=1.5)
    rng = np.random.default_rng(nx)
    ref = ob.OracleLattice(nx, ny, nz, ob.default_params(**par))
    ref.init_stripe(0.5)       # both fluids present on any lattice (a "droplet" of radius 0.3 nx does not exist on
                               # a lattice thinner than nx/2 - 0.3 nx: LBM_binary.H:725 measures z from box[0]/2)
    f0 = ref.f * (1 + 0.03 * rng.standard_normal(ref.f.shape))
    g0 = ref.g * (1 + 0.03 * rng.standard_normal(ref.g.shape))
    ref.init_from(f0, g0)
    nslabs = 2 if nz >= 8 else 1
    runs = [pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(**par), schedule=schedule),
            pkg.RingLBM(nx, ny, nz, nslabs=nslabs, params=pkg.default_params(**par), schedule=schedule)]
    for lbm in runs:
        lbm.LBM_init(f0, g0)                                   # upload (box copies into padded rows)
    for step in range(3):
        ref.timestep()
        for lbm in runs:
            lbm.LBM_timestep(1)
    for lbm in runs:
        f, g = lbm.populations()                               # download
        assert np.array_equal(f, ref.f) and np.array_equal(g, ref.g)
        assert np.array_equal(lbm.LBM_hydrovars(), ref.h)
        assert np.array_equal(lbm.LBM_hydrovars_density(), ref.hbar[:9])
        fn, gn = lbm.thermal_noise()
        assert np.array_equal(fn, ref.fn) and np.array_equal(gn, ref.gn)
        np.testing.assert_allclose(lbm.update_com(), ref.com(), rtol=1e-12)
        np.testing.assert_allclose(lbm.mass(), [ref.hbar[0].sum(), ref.hbar[1].sum()], rtol=1e-13)
        m = lbm.droplet_moments()
        x = np.arange(nx)[None, None, :]
        np.testing.assert_allclose(m[1], (ref.hbar[0] * x).sum(), rtol=1e-12)
        np.testing.assert_allclose(m[4], (ref.hbar[0] * x * x).sum(), rtol=1e-12)
    one = runs[0]
    # injected noise (dense arrays into the step)
    fn = 1e-3 * rng.standard_normal(ref.fn.shape); gn = 1e-3 * rng.standard_normal(ref.gn.shape)
    fn[0] = 0; gn[0] = 0; gn[1:4] = -fn[1:4]
    one.inject_noise(fn, gn); one.LBM_timestep(1)
    ref.timestep_injected(fn, gn)
    f, g = one.populations()
    assert np.array_equal(f, ref.f) and np.array_equal(g, ref.g)
    # structure factor of the padded state against the host twin
    names = pkg.plotfile.variable_names(22)
    dev, host = pkg.structfact.DeviceStructFact(one, names), pkg.structfact.StructFact(names)
    dev.fort_structure(); host.fort_structure(one.LBM_hydrovars(), 0)
    d, h = dev.mean(1), host.mean(1)
    assert np.abs(d - h).max() <= 1e-11 * np.abs(h).max()
    # reference-state noise (dense global fields, shifted lookup)
    rho_eq = 0.3 + rng.random((nz, ny, nx)); phi_eq = 0.2 + rng.random((nz, ny, nx))
    com_ref = ref.com() - np.array([1.6, -1.4, 0.3])
    for lbm in runs:
        lbm.set_ref_state(rho_eq, phi_eq, rho_eq + phi_eq, com_ref)
    ref2 = ob.OracleLattice(nx, ny, nz, ob.default_params(**par))
    ref2.set_ref_state(rho_eq, phi_eq, rho_eq + phi_eq, com_ref)
    f1, g1 = runs[1].populations()
    ref2.init_from(f1, g1)
    runs[1].LBM_init(f1, g1)
    ref2.timestep(); runs[1].LBM_timestep(1)
    fr, gr = runs[1].populations()
    assert np.array_equal(fr, ref2.f) and np.array_equal(gr, ref2.g)
    dev.close()
    for lbm in runs:
        lbm.close()
