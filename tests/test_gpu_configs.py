"""GPU: BASELINE.json's configs at their stated workloads.

  configs[0]  64^3 flat interface, zero noise: GPU vs the CPU oracle after 1, 10 and 100 steps -- bit-equal
              populations and hydrovars for the two exact schedules, and the north-star tolerances (SURVEY 8d:
              rho, phi, rho+phi relative 1e-12; velocities absolute 1e-12 cs) stated explicitly.
  configs[2]  256^3 with thermal noise (NoiseCovariance.ipynb parameters): per-mode variances at the real size.
  configs[3]  512^3 droplet r = 0.2 in 4 z-slabs of 512x512x128 (native ring on one GPU) == the single context for the exact
              schedules, and within the strict tolerances of it under the default schedule (auto -> hand-over).
  configs[4]  a pair of 1024x1024x64 slabs == the single 1024x1024x128 context.
plus the tolerance contract of the density hand-over schedule (csrc/bflbm_handover.h).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
CS = np.sqrt(1.0 / 3.0)


def _tolerances(h, href):
    """densities relative 1e-12, velocities absolute 1e-12 cs (their self-relative error is meaningless)."""
    for comp in (0, 1, 5):
        np.testing.assert_allclose(h[comp], href[comp], rtol=1e-12, atol=0)
    for comp in (2, 3, 4, 6, 7, 8, 15, 16, 17):
        np.testing.assert_allclose(h[comp], href[comp], rtol=0, atol=1e-12 * CS)


@pytest.mark.parametrize("schedule", ["two_pass", "fused"])
def test_config0_64_cubed_stripe_against_the_oracle(pkg, ob, schedule):
    n = 64
    ob.lib().orc_set_threads(16)
    try:
        ref = ob.OracleLattice(n, n, n)
        ref.init_stripe(0.5)
        lbm = pkg.BinaryLBM(n, n, n, schedule=schedule)
        lbm.LBM_init_stripe(0.5)
        done = 0
        for steps in (1, 10, 100):
            for _ in range(steps - done):
                ref.timestep()
            lbm.LBM_timestep(steps - done)
            done = steps
            f, g = lbm.populations()
            h = lbm.LBM_hydrovars()
            _tolerances(h, ref.h)
            assert np.array_equal(f, ref.f) and np.array_equal(g, ref.g), (schedule, steps)
            assert np.array_equal(h, ref.h), (schedule, steps)
        lbm.close()
    finally:
        ob.lib().orc_set_threads(1)


def test_config2_noise_mode_variances_at_256_cubed(pkg, ob):
    """NoiseCovariance.ipynb cell 3 (tau = 1, kBT = 1e-5, alpha0 = 0): per-mode variance / theory; the notebook's
    16^3 x 200 frames gave mean 1.00041.  One 256^3 frame has 1.7e7 samples per mode: standard error of a variance
    ratio sqrt(2/N) = 3.5e-4; assert 5 sigma = 1.8e-3 per mode, 6e-4 for the mean over the 18 modes."""
    n = 256
    par = dict(kBT=1e-5, alpha0=0.0, tau_f=1.0, tau_g=1.0)
    lbm = pkg.BinaryLBM(n, n, n, params=pkg.default_params(**par))
    lbm.LBM_init_mixture()
    fn, gn = lbm.thermal_noise()
    _, _, b = ob.lattice_tables()
    lam = 1.0 / 1.5
    base = 2.0 * (lam - 0.5 * lam * lam) * 1e-5
    theory = np.array([0.0] + [base * 0.5] * 3 + [base * 3.0 * b[a] for a in range(4, 19)])
    assert not fn[0].any() and not gn[0].any()
    for a in (1, 2, 3):
        assert np.array_equal(gn[a], -fn[a])                         # exactly anticorrelated (LBM_binary.H:118)
    vf = np.array([np.mean(fn[a] * fn[a]) for a in range(1, 19)]) / theory[1:]
    vg = np.array([np.mean(gn[a] * gn[a]) for a in range(1, 19)]) / theory[1:]
    se = np.sqrt(2.0 / n ** 3)
    assert np.all(np.abs(vf - 1) < 5 * se), vf
    assert np.all(np.abs(vg - 1) < 5 * se), vg
    assert abs(vf.mean() - 1) < 6e-4 and abs(vg[3:].mean() - 1) < 6e-4
    for a in (4, 9, 18):
        assert abs(fn[a].mean()) < 5 * np.sqrt(theory[a] / n ** 3)
    lbm.close()


@pytest.mark.parametrize("schedule", ["fused", "two_pass"])
def test_config3_512_cubed_droplet_in_four_slabs(pkg, schedule):
    """4 slabs of 512x512x128 on one GPU (peer copies are device copies) against the undecomposed 512^3 box,
    8 steps: identical doubles in all 9 hydrovsbar components (both schedules are bit-exact)."""
    n, steps = 512, 8
    a = pkg.BinaryLBM(n, n, n, schedule=schedule)
    a.LBM_init_droplet(0.2)
    a.LBM_timestep(steps)
    ha = a.LBM_hydrovars_density()
    a.close()
    r = pkg.RingLBM(n, n, n, nslabs=4, devices=(0,), schedule=schedule)
    r.LBM_init_droplet(0.2)
    r.LBM_timestep(steps)
    hr = r.LBM_hydrovars_density()
    r.close()
    for comp in range(9):
        assert np.array_equal(ha[comp], hr[comp]), comp


def test_config3_512_cubed_droplet_in_four_slabs_default_schedule(pkg):
    """configs[3] AS SHIPPED: `auto` resolves to the hand-over kernel in every slab's interior sweep (frames) with pulled
    rings in the boundary plane pairs -- what bench.py --gpus 4 and every user get.  4 slabs of 512x512x128 against the
    exact single-context run, 12 steps: first step bit-equal, then the strict SURVEY 8d tolerances at every site."""
    n = 512
    a = pkg.BinaryLBM(n, n, n, schedule="fused")
    a.LBM_init_droplet(0.2)
    r = pkg.RingLBM(n, n, n, nslabs=4, devices=(0,))                       # schedule: auto (the default)
    r.LBM_init_droplet(0.2)
    assert all(s.resolved_schedule() == "handover" for s in r.slabs)
    a.LBM_timestep(1); r.LBM_timestep(1)
    assert np.array_equal(a.LBM_hydrovars_density(), r.LBM_hydrovars_density())
    a.LBM_timestep(11); r.LBM_timestep(11)
    ha, hr = a.LBM_hydrovars(), r.LBM_hydrovars()
    a.close(); r.close()
    assert not np.array_equal(ha, hr)                                      # the frames were in use
    _tolerances(hr, ha)


def test_config4_pair_of_1024x1024x64_slabs(pkg):
    """configs[4]'s slab shape: two 1024x1024x64 slabs against the single 1024x1024x128 context (82 GB each way),
    droplet init (analytic, evaluated per slab; a homogeneous mixture would exercise no indexing), 6 steps."""
    nx, ny, nz, steps = 1024, 1024, 128, 6
    a = pkg.BinaryLBM(nx, ny, nz, schedule="fused")
    a.LBM_init_droplet(0.05)
    a.LBM_timestep(steps)
    ha = a.LBM_hydrovars_density()
    a.close()
    r = pkg.RingLBM(nx, ny, nz, nslabs=2, devices=(0,), schedule="fused")
    r.LBM_init_droplet(0.05)
    r.LBM_timestep(steps)
    hr = r.LBM_hydrovars_density()
    r.close()
    for comp in range(9):
        assert np.array_equal(ha[comp], hr[comp]), comp


@pytest.mark.parametrize("shape,init", [((128, 64, 48), ("droplet", 0.25)), ((256, 32, 24), ("stripe", 0.5)),
                                        ((128, 8, 2), ("droplet", 0.3)), ((128, 8, 3), ("stripe", 0.5)), ((192, 12, 5), ("droplet", 0.3)),
                                        ((128, 8, 9), ("droplet", 0.3))])
def test_handover_schedule_tolerance_contract(pkg, shape, init):
    """Schedule 3 hands the tile-ring densities over from the previous step: the first step after an init
    pulls them (bit-identical to schedule 1); afterwards the ring sums have another fixed order, so the
    results agree with the exact schedule to rounding -- asserted at the north-star tolerances after 50 steps
    -- and are reproducible run to run."""
    def run(schedule, steps):
        with pkg.BinaryLBM(*shape, schedule=schedule) as l:
            getattr(l, "LBM_init_" + init[0])(*init[1:])
            l.LBM_timestep(steps)
            return l.populations(), l.LBM_hydrovars()
    (fe, ge), _ = run("fused", 1)
    (fh, gh), _ = run("handover", 1)
    assert np.array_equal(fe, fh) and np.array_equal(ge, gh)
    (fe, ge), he = run("fused", 50)
    (fh, gh), hh = run("handover", 50)
    if shape[2] >= 8:
        assert not np.array_equal(fe, fh)            # the frames are in use (a chunk of fewer than 4 planes has none)
    assert max(np.abs(fe - fh).max(), np.abs(ge - gh).max()) < 1e-14
    _tolerances(hh, he)
    (fh2, gh2), _ = run("handover", 50)
    assert np.array_equal(fh, fh2) and np.array_equal(gh, gh2)


def test_auto_schedule_at_256_cubed_is_within_tolerance_of_the_exact_one(pkg):
    """configs[1] with the default (auto) schedule = the pipelined hand-over kernel: 30 steps of the 256^3 droplet agree
    with the bit-exact fused schedule at the north-star tolerances; a 256^3 box in 3 slabs (hand-over frames inside every
    slab's interior sweep, pulled rings at the slab faces) does too."""
    n, steps = 256, 30
    res = {}
    for name, make in (("exact", lambda: pkg.BinaryLBM(n, n, n, schedule="fused")), ("auto", lambda: pkg.BinaryLBM(n, n, n)),
                       ("auto_ring", lambda: pkg.RingLBM(n, n, n, nslabs=3, devices=(0,)))):
        l = make()
        l.LBM_init_droplet(0.2)
        l.LBM_timestep(steps)
        res[name] = l.LBM_hydrovars()
        l.close()
    assert not np.array_equal(res["exact"], res["auto"])        # auto really is the other kernel
    _tolerances(res["auto"], res["exact"])
    _tolerances(res["auto_ring"], res["exact"])


@pytest.mark.parametrize("shape", [(128, 16, 12), (256, 32, 40)])
def test_thermal_noise_in_the_pipelined_kernel(pkg, shape):
    """With kBT > 0 auto runs the pipelined hand-over kernel with the generator inside (one pass per step).  Same
    stream and amplitudes as the two-pass schedule: the first step is bit-identical, later steps differ only through the
    summation order of the tile-ring densities (rounding level after 30 steps)."""
    par = pkg.default_params(kBT=1e-5, alpha0=1.0)
    out = {}
    for steps in (1, 30):
        for sched in ("two_pass", "handover"):
            with pkg.BinaryLBM(*shape, params=par, schedule=sched) as l:
                l.LBM_init_droplet(0.25)
                l.LBM_timestep(steps)
                out[sched, steps] = l.populations()
    for k in (0, 1):
        assert np.array_equal(out["two_pass", 1][k], out["handover", 1][k])
        assert np.abs(out["two_pass", 30][k] - out["handover", 30][k]).max() < 1e-13
    f = out["handover", 30][0]
    assert np.abs(f - f.mean(axis=(1, 2, 3), keepdims=True)).max() > 1e-6          # the noise is there


def test_handover_random_cases(pkg):
    """Seeded random lattices of full 64 x 4 tiles, inits, parameters and step counts: the hand-over schedule stays within
    the north-star tolerances of the bit-exact fused schedule, single context and rings of 2-3 slabs alike."""
    rng = np.random.default_rng(20261004)
    for case in range(10):
        nx = int(rng.choice([128, 192, 256]))
        ny = int(rng.choice(np.arange(8, 44, 4)))
        nz = int(rng.integers(4, 28))
        steps = int(rng.integers(3, 30))
        par = pkg.default_params(alpha0=float(rng.choice([0.0, 1.5, 2.5, 4.0])), tau_f=float(rng.choice([0.5, 0.8, 1.0])),
                                 tau_g=float(rng.choice([0.5, 0.6, 1.0])), kappa=float(rng.choice([0.1, 1.0, 4.0])), rho_hi=float(rng.choice([1.0, 3.0])))
        init = ("droplet", float(rng.uniform(0.05, 0.3))) if rng.random() < 0.6 else ("stripe", float(rng.uniform(0.3, 0.7)))
        nslabs = int(rng.choice([1, 1, 2, 3])) if nz >= 12 else 1
        out = {}
        for sched in ("fused", "handover"):
            l = (pkg.BinaryLBM(nx, ny, nz, params=par, schedule=sched) if nslabs == 1 or sched == "fused"
                 else pkg.RingLBM(nx, ny, nz, nslabs=nslabs, params=par, schedule=sched))
            getattr(l, "LBM_init_" + init[0])(init[1])
            l.LBM_timestep(steps)
            out[sched] = l.LBM_hydrovars()
            l.close()
        assert np.isfinite(out["handover"]).all(), (case, nx, ny, nz)
        _tolerances(out["handover"], out["fused"])


def test_config4_spinodal_slabs_with_the_default_schedule(pkg):
    """configs[4] as a user runs it: LBM_init_mixture is homogeneous (LBM_binary.H:613-614), so the "spinodal mixture" is an
    alpha0 above the demixing threshold with kBT = 1e-5 seeding the decomposition.  SURVEY's example alpha0 = 4
    is NaN within 50 steps on the reference's own CPU path (rho = phi = 1: total density 2, interaction strength 8;
    tests/test_oracle_pins.py) and `auto` keeps such a run on an exact schedule; alpha0 = 2.5 demixes into stable domains
    (rho = 2.23 / 0) and is what runs here: two 1024 x 1024 x 64 slabs under `auto` (= the
    hand-over kernel with the generator inside, in every slab's interior sweep and boundary pairs).  The noise stream is
    a function of the GLOBAL site index, so the first step equals the undecomposed two-pass run bit for bit; after 8
    steps the two agree at the north-star tolerance (metric of tests/tolerances.py); mass is conserved to rounding."""
    import tolerances
    nx, ny, nz = 1024, 1024, 128
    par = pkg.default_params(kBT=1e-5, alpha0=2.5)
    hot = pkg.BinaryLBM(128, 128, 64, params=pkg.default_params(kBT=1e-5, alpha0=4.0))
    hot.LBM_init_mixture()
    assert hot.state_total_max == 2.0 and hot.resolved_schedule() == "two_pass"    # 4 x 2 > 6: never the hand-over kernel
    hot.close()
    out = {}
    for name, make in (("exact", lambda: pkg.BinaryLBM(nx, ny, nz, params=par, schedule="two_pass")),
                       ("slabs", lambda: pkg.RingLBM(nx, ny, nz, nslabs=2, devices=(0,), params=par))):
        l = make()
        if name == "slabs":
            assert l.slabs[0].resolved_schedule() == "handover"
        l.LBM_init_mixture()
        m0 = l.mass()
        l.LBM_timestep(1)
        h1 = l.LBM_hydrovars_density() if name == "exact" else l.LBM_hydrovars_density()
        l.LBM_timestep(7)
        out[name] = (h1, l.LBM_hydrovars(ncomp=9), m0, l.mass())
        l.close()
    assert np.array_equal(out["exact"][0], out["slabs"][0])                     # step 1: same normals at every site
    tolerances.check(out["slabs"][1], out["exact"][1], "1024x1024x128 spinodal, 8 steps")
    for name in out:
        m0, m1 = out[name][2], out[name][3]
        assert abs(m1[0] - m0[0]) <= 1e-12 * m0[0] and abs(m1[1] - m0[1]) <= 1e-12 * m0[1]
    assert out["slabs"][1][0].std() > 1e-4                                       # the noise is there
