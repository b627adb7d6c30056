"""GPU: the reference's job flow (main_run_job.cpp) end to end -- frames, checkpoints, equilibrium-state
extraction, restart from the last frame -- through run_job.py and the plotfile layer."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_job_outputs_and_restart_equals_uninterrupted(pkg, ob, tmp_path):
    pf = pkg.plotfile
    common = ["--system", "droplet", "--nx", "16", "--alpha0", "2.0", "--radius", "0.3", "--plot-int", "10", "--print-int", "0"]
    a_root, b_root = str(tmp_path / "A"), str(tmp_path / "B")
    assert pkg.run_job.main(common + ["--nsteps", "40", "--root", a_root]) == 0
    d = os.path.join(a_root, "data_droplet_density_1.00_alpha0_2.00_r0.300_size16-16-16")
    run = os.path.join(d, "lbm_data_shshan_alpha0_2.00_xi_0.0e+00_size16-16-16")
    assert sorted(os.listdir(run)) == ["plt%07d" % s for s in (0, 10, 20, 30, 40)]
    # frames hold hydrovs with the reference's names, 8 boxes (max_grid_size = nx/2)
    h40, hdr = pf.read_plotfile(os.path.join(run, "plt0000040"))
    assert hdr["names"] == pf.variable_names(22) and hdr["ngrids"] == 8 and hdr["time"] == 40.0
    ref = ob.OracleLattice(16, 16, 16, params=ob.default_params(alpha0=2.0))
    ref.init_droplet(0.3)
    frames = {0: ref.h.copy()}
    for s in range(1, 41):
        ref.timestep()
        if s % 10 == 0:
            frames[s] = ref.h.copy()
    assert np.array_equal(h40, frames[40])
    assert np.array_equal(pf.read_plotfile(os.path.join(run, "plt0000000"))[0], frames[0])
    # checkpoints: one name, 19 components, the post-stream populations
    chk = os.path.join(d, "f_checkpoint0000040_alpha0_2.00_xi_0.0e+00_size16-16-16")
    f40, hdr = pf.read_plotfile(chk)
    assert hdr["names"] == ["rho_chk"] and np.array_equal(f40, ref.f)
    # equilibrium state = ensemble mean of comps 0, 1, 5 over the last t_window/plot_int + 1 = 5 frames (0..40 here)
    rho_eq, hdr = pf.read_plotfile(os.path.join(d, "equilibrium_rho_alpha0_2.00_size16-16-16"))
    mean = np.zeros_like(frames[0][0])
    for s in (0, 10, 20, 30, 40):
        mean = mean + frames[s][0]
    assert np.array_equal(rho_eq[0], mean * (1.0 / 5)) and hdr["names"] == ["rho_eq"]
    rhot_eq, _ = pf.read_plotfile(os.path.join(d, "equilibrium_rhot_alpha0_2.00_size16-16-16"))
    np.testing.assert_allclose(rhot_eq[0], sum(frames[s][5] for s in (0, 10, 20, 30, 40)) / 5, rtol=1e-15)

    # restart from the step-40 checkpoint and run 20 more == 60 uninterrupted steps
    assert pkg.run_job.main(common + ["--nsteps", "20", "--root", a_root, "--restart", "--step-continue", "40"]) == 0
    assert pkg.run_job.main(common + ["--nsteps", "60", "--root", b_root]) == 0
    for fl in "fg":
        name = "%s_checkpoint0000060_alpha0_2.00_xi_0.0e+00_size16-16-16" % fl
        x, _ = pf.read_plotfile(os.path.join(d, name))
        y, _ = pf.read_plotfile(os.path.join(b_root, os.path.relpath(d, a_root), name))
        assert np.array_equal(x, y), fl
    for _ in range(20):
        ref.timestep()
    assert np.array_equal(pf.read_plotfile(os.path.join(run, "plt0000060"))[0], ref.h)


def test_noisy_job_writes_noise_frames(pkg, tmp_path):
    pf = pkg.plotfile
    root = str(tmp_path / "N")
    assert pkg.run_job.main(["--system", "mixture", "--nx", "8", "--alpha0", "0", "--kbt", "1e-5", "--tau", "1", "--nsteps", "20",
                             "--plot-int", "5", "--out-noise-step", "10", "--print-int", "0", "--root", root]) == 0
    run = os.path.join(root, "data_mixture_hydrovars", "lbm_data_shshan_alpha0_0.00_xi_1.0e-05_size8-8-8_continue")
    # frames only from out_step = 2*nsteps/10 on (main_run_job.cpp:89), plus the initial and the last one
    assert sorted(x for x in os.listdir(run) if x.startswith("plt")) == ["plt%07d" % s for s in (0, 5, 10, 15, 20)]
    fn, hdr = pf.read_plotfile(os.path.join(run, "data_fnoise", "fn0000010"))
    gn, _ = pf.read_plotfile(os.path.join(run, "data_gnoise", "gn0000010"))
    assert hdr["names"] == ["fa%d" % k for k in range(19)]          # NoiseCovariance.ipynb reads fa%d / ga%d
    assert np.all(fn[0] == 0) and np.array_equal(gn[1:4], -fn[1:4]) and fn[4:].std() > 0
    assert not os.path.exists(os.path.join(root, "data_mixture_hydrovars", "equilibrium_rho_alpha0_0.00_size8-8-8"))


def test_noisy_job_structure_factor_window(pkg, tmp_path):
    """main_run_job.cpp:99-103, :342-349: accumulate every out_SF_step inside the last plot_SF_window steps and
    write <plot root>_SF_mag / _real_imag with the last frame; S_rho/(kBT/cs2) averages to ~1 (Mixture.ipynb)."""
    pf = pkg.plotfile
    root = str(tmp_path / "S")
    kbt = 1e-5
    assert pkg.run_job.main(["--system", "mixture", "--nx", "16", "--alpha0", "0", "--kbt", str(kbt), "--tau", "1",
                             "--nsteps", "1500", "--plot-int", "500", "--plot-sf-window", "1000", "--out-sf-step", "20",
                             "--print-int", "0", "--root", root]) == 0
    run = os.path.join(root, "data_mixture_hydrovars", "lbm_data_shshan_alpha0_0.00_xi_1.0e-05_size16-16-16_continue")
    mag, hdr = pf.read_plotfile(os.path.join(run, "plt_SF_mag000001500"))
    assert hdr["names"][0] == "struct_fact_rho_rho" and mag.shape == (22, 16, 16, 16)
    s_rho = mag[0] / (kbt * 3.0)
    assert s_rho[8, 8, 8] == 0.0                                  # zero_avg: k = 0 sits at the box centre
    assert abs(s_rho.sum() / (16 ** 3 - 1) - 1.0) < 0.05, s_rho.sum() / (16 ** 3 - 1)
    # the lattice velocity ufbar = j_f/rho is the equipartitioned one (the real velocity ufx mixes in the
    # inter-species relaxation and half the momentum noise, cf. the markdown cell 1 of Mixture.ipynb)
    k = hdr["names"].index("struct_fact_ufbarx_ufbarx")
    assert abs(mag[k].sum() / (16 ** 3 - 1) / kbt - 1.0) < 0.08, mag[k].sum() / (16 ** 3 - 1) / kbt
    assert os.path.isdir(os.path.join(run, "plt_SF_real_imag000001500"))


def test_noisy_continuation_on_the_equilibrium_reference_state(pkg, tmp_path):
    """ReadMe workflow B: relax a droplet at kBT = 0 (writes equilibrium_{rho,phi,rhot}_* and the last-frame
    checkpoints), then continue from that frame with noise whose amplitudes come from the equilibrium
    state (--use-ref-state = a USE_REF_STATE build, main_run_job.cpp:216-235, :253-270)."""
    pf = pkg.plotfile
    root = str(tmp_path / "R")
    common = ["--system", "droplet", "--nx", "16", "--alpha0", "2.0", "--radius", "0.3", "--print-int", "0", "--root", root]
    assert pkg.run_job.main(common + ["--nsteps", "40", "--plot-int", "10"]) == 0
    d = os.path.join(root, "data_droplet_density_1.00_alpha0_2.00_r0.300_size16-16-16")
    noisy = common + ["--kbt", "1e-6", "--nsteps", "20", "--plot-int", "20", "--restart", "--step-continue", "40"]
    assert pkg.run_job.main(noisy + ["--use-ref-state"]) == 0
    f_ref, _ = pf.read_plotfile(os.path.join(d, "f_checkpoint0000060_alpha0_2.00_xi_1.0e-06_size16-16-16"))
    assert pkg.run_job.main(noisy) == 0
    f_cur, _ = pf.read_plotfile(os.path.join(d, "f_checkpoint0000060_alpha0_2.00_xi_1.0e-06_size16-16-16"))
    assert np.all(np.isfinite(f_ref)) and not np.array_equal(f_ref, f_cur)
    # same random stream, slightly different amplitudes: the two runs stay close, and both keep the mass
    assert np.abs(f_ref - f_cur).max() < 1e-3
    f40, _ = pf.read_plotfile(os.path.join(d, "f_checkpoint0000040_alpha0_2.00_xi_0.0e+00_size16-16-16"))
    assert abs(f_ref.sum() - f40.sum()) < 1e-9 and abs(f_cur.sum() - f40.sum()) < 1e-9
