"""GPU: the reference's USE_REF_STATE noise branch (LBM_binary.H:12, :92-107, :585-590) as the run-time
switch bflbm_set_ref_state / bflbm_enable_ref_state.

The branch is compiled out in the shipped reference and none of its notebooks records a number produced
with it: parity is against the oracle's restatement of the branch only ("parity unpinned" beyond that).
Bit-exact comparisons; the shift static_cast<int>(COM - com_ref) is kept away from integer crossings by
the choice of com_ref (the COM sums of the two implementations differ in summation order, ~1e-15)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = (12, 10, 14)
PAR = dict(kBT=1e-5, alpha0=2.0)


def _ref_fields(n, seed=3):
    nx, ny, nz = n
    rng = np.random.default_rng(seed)
    rho = 0.2 + rng.random((nz, ny, nx))
    phi = 0.1 + rng.random((nz, ny, nx))
    rhot = rho + phi + 0.01 * rng.random((nz, ny, nx))      # an independent field in the reference
    return rho, phi, rhot


def _pair(pkg, ob, init, com_off, **kw):
    nx, ny, nz = N
    lbm = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(**PAR), **kw)
    orc = ob.OracleLattice(nx, ny, nz, ob.default_params(**PAR))
    # the centre of mass of the initial droplet, then a reference offset by com_off -> shift = trunc(com_off)
    tmp = ob.OracleLattice(nx, ny, nz, ob.default_params(**PAR))
    tmp.init_droplet(0.3)
    com_ref = tmp.com() - np.asarray(com_off)
    rho, phi, rhot = _ref_fields(N)
    lbm.set_ref_state(rho, phi, rhot, com_ref)
    orc.set_ref_state(rho, phi, rhot, com_ref)
    if init == "droplet":
        lbm.LBM_init_droplet(0.3); orc.init_droplet(0.3)
    elif init == "mixture":
        lbm.LBM_init_mixture(); orc.init_mixture()
    else:
        f0, g0 = tmp.f.copy(), tmp.g.copy()
        lbm.LBM_init(f0, g0); orc.init_from(f0, g0)
    return lbm, orc


def _same(lbm, orc):
    f, g = lbm.populations()
    assert np.array_equal(f, orc.f) and np.array_equal(g, orc.g)
    fn, gn = lbm.thermal_noise()
    assert np.array_equal(fn, orc.fn) and np.array_equal(gn, orc.gn)
    assert np.array_equal(lbm.LBM_hydrovars(), orc.h)


@pytest.mark.parametrize("init", ["droplet", "mixture", "upload"])
def test_ref_state_noise_matches_oracle(pkg, ob, init):
    """Zero shift after LBM_init_droplet (:739), absolute COM after LBM_init_mixture (:623-625), COM - com_ref
    after LBM_init (:651-654) and after every step (:588); shifts here are (2,-1,3) resp. the COM itself."""
    lbm, orc = _pair(pkg, ob, init, (2.6, -1.4, 3.3))
    _same(lbm, orc)
    for _ in range(4):
        lbm.LBM_timestep(1); orc.timestep()
        _same(lbm, orc)
    lbm.close()


def test_ref_state_really_changes_the_noise(pkg, ob):
    nx, ny, nz = N
    a = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(**PAR))
    a.LBM_init_droplet(0.3)
    fn0, _ = a.thermal_noise()
    a.set_ref_state(*_ref_fields(N), com_ref=(6.0, 5.0, 6.0))
    fn1, _ = a.thermal_noise()
    assert not np.array_equal(fn0, fn1)
    # amplitude of mode 4 follows rho_eq, not rho: fn/sqrt(|rho_eq|) is the same normal field as before
    rho_eq = _ref_fields(N)[0]
    h = a.LBM_hydrovars_density()
    m = h[0] > 1e-3
    np.testing.assert_allclose((fn1[4] / np.sqrt(rho_eq))[m], (fn0[4] / np.sqrt(h[0]))[m], rtol=1e-12)
    a.disable_ref_state()
    fn2, _ = a.thermal_noise()
    assert np.array_equal(fn0, fn2)
    # kBT = 0: the switch is inert
    a.set_params(kBT=0.0)
    a.set_ref_state(*_ref_fields(N), com_ref=(6.0, 5.0, 6.0))
    assert not a.ref_state_active
    a.close()


@pytest.mark.parametrize("nslabs", [2, 3])
def test_ref_state_is_independent_of_the_decomposition(pkg, ob, nslabs):
    """The reference fields cover the global lattice on every slab and the COM is global."""
    nx, ny, nz = N
    tmp = ob.OracleLattice(nx, ny, nz, ob.default_params(**PAR))
    tmp.init_droplet(0.3)
    com_ref = tmp.com() - np.array([2.6, -1.4, 3.3])
    rho, phi, rhot = _ref_fields(N)
    one = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(**PAR))
    one.set_ref_state(rho, phi, rhot, com_ref)
    one.LBM_init(tmp.f, tmp.g)
    rings = [pkg.RingLBM(nx, ny, nz, nslabs=nslabs, params=pkg.default_params(**PAR)),
             pkg.LocalSlabRing(nx, ny, nz, nslabs, params=pkg.default_params(**PAR))]
    for r in rings:
        r.set_ref_state(rho, phi, rhot, com_ref)
        r.LBM_init(tmp.f, tmp.g)
    for step in range(4):
        f, g = one.populations()
        fn, gn = one.thermal_noise()
        h = one.LBM_hydrovars()
        for r in rings:
            fr, gr = r.populations()
            assert np.array_equal(f, fr) and np.array_equal(g, gr), (type(r).__name__, step)
            fnr, gnr = r.thermal_noise()
            assert np.array_equal(fn, fnr) and np.array_equal(gn, gnr)
            assert np.array_equal(h, r.LBM_hydrovars())
            r.LBM_timestep(1)
        one.LBM_timestep(1)
    one.close()
    for r in rings:
        r.close()


def test_forced_fused_schedule_falls_back_with_a_reference_state(pkg, ob):
    lbm, orc = _pair(pkg, ob, "upload", (2.6, -1.4, 3.3), schedule="fused")
    for _ in range(3):
        lbm.LBM_timestep(1); orc.timestep()
    _same(lbm, orc)
    lbm.close()


def test_slab_without_global_com_fails_loudly(pkg):
    nx, ny, nz = N
    e = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(**PAR), z0=0, z1=7, rank=0, nranks=2)
    e.set_ref_state(*_ref_fields(N), com_ref=(6.0, 5.0, 6.0))
    e.upload(np.ones((19, 7, ny, nx)), np.ones((19, 7, ny, nx)))
    e.commit_upload()
    with pytest.raises(pkg.BflbmError, match="bflbm_set_com"):
        e.step_boundary()
    e.close()


def test_shift_larger_than_the_lattice_is_periodic(pkg, ob):
    """The reference wraps the shifted index once (LBM_binary.H:98-103), which is only in range for
    |COM - com_ref| < n.  Here the shift trunc(COM - com_ref) is reduced modulo the lattice first (identical
    below n), so the cell read is (x - trunc(rel)) mod n: a reference centre of mass several lattices away --
    on the same side, truncation being toward zero -- gives the same noise as the near one, on the GPU and in
    the oracle, instead of an out-of-range read (tests/test_oracle_pins.py has the sign-flipping cases)."""
    nx, ny, nz = N
    tmp = ob.OracleLattice(nx, ny, nz, ob.default_params(**PAR))
    tmp.init_droplet(0.3)
    rho, phi, rhot = _ref_fields(N)
    near = tmp.com() - np.array([2.6, -1.4, 3.3])
    far = near - np.array([3 * nx, -2 * ny, 5 * nz])           # shifts (38, -21, 73) on a 12 x 10 x 14 lattice
    res = []
    for com_ref in (near, far):
        lbm = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(**PAR))
        orc = ob.OracleLattice(nx, ny, nz, ob.default_params(**PAR))
        lbm.set_ref_state(rho, phi, rhot, com_ref); orc.set_ref_state(rho, phi, rhot, com_ref)
        lbm.LBM_init(tmp.f, tmp.g); orc.init_from(tmp.f, tmp.g)
        for _ in range(2):
            lbm.LBM_timestep(1); orc.timestep()
        _same(lbm, orc)
        res.append(lbm.populations()[0])
        lbm.close()
    assert np.array_equal(res[0], res[1])


def test_fluid_without_mass_fails_loudly(pkg):
    """No fluid f at all: its centre of mass is 0/0.  The lookup must not be attempted with that."""
    nx, ny, nz = N
    lbm = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(**PAR))
    lbm.set_ref_state(*_ref_fields(N), com_ref=(6.0, 5.0, 6.0))
    f0 = np.zeros((19, nz, ny, nx)); g0 = np.full((19, nz, ny, nx), 1.0 / 19)
    lbm.upload(f0, g0); lbm.commit_upload()
    with pytest.raises(pkg.BflbmError, match="not finite"):
        lbm.LBM_timestep(1)
    with pytest.raises(pkg.BflbmError, match="not finite"):
        lbm.set_com((float("nan"), 0.0, 0.0))
    lbm.close()
