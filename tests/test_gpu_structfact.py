"""GPU: the structure-factor accumulator on the device (csrc/bflbm_sf.h, hipFFT D2Z) against its host-side
twin (structfact.py, numpy FFT of downloaded frames).  FHDeX's StructFact is un-vendored: both follow the
reference's call sites (main_run_job.cpp:301-310, :342-349, :50-54) -- parity unpinned beyond that.
Two FFT libraries agree to rounding: tolerance 1e-11 of the largest |S| of each pair."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _agree(dev, host):
    scale = np.abs(host).max(axis=(1, 2, 3), keepdims=True)
    scale[scale == 0] = 1.0
    assert np.abs(dev - host).max() <= 1e-11 * scale.max() or np.all(np.abs(dev - host) <= 1e-11 * scale)


@pytest.mark.parametrize("n", [(16, 16, 16), (12, 10, 14), (9, 7, 5)])
def test_device_structure_factor_matches_host(pkg, n):
    nx, ny, nz = n
    lbm = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(kBT=1e-5, alpha0=1.0, tau_f=1.0, tau_g=1.0))
    lbm.LBM_init_mixture()
    names = pkg.plotfile.variable_names(22)
    host = pkg.structfact.StructFact(names)
    dev = pkg.structfact.DeviceStructFact(lbm, names)
    assert dev.pair_names() == host.pair_names() and len(dev.pairs) == 22
    lbm.LBM_timestep(20)
    for _ in range(3):
        lbm.LBM_timestep(5)
        host.fort_structure(lbm.LBM_hydrovars(), 0)
        dev.fort_structure()
    assert dev.nsamples == host.nsamples == 3
    for zero_avg in (1, 0):
        h = host.mean(zero_avg)
        d = dev.mean(zero_avg)
        _agree(d.real, h.real)
        _agree(d.imag, h.imag)
        _agree(dev.magnitude(zero_avg), np.abs(h))
    # auto-correlations are real and non-negative; the spectrum of real fields is Hermitian
    d = dev.mean(0)
    assert np.all(d[0].real >= 0) and np.abs(d[0].imag).max() <= 1e-25
    # reset restarts the average
    dev.fort_structure(reset=1)
    host.fort_structure(lbm.LBM_hydrovars(), 1)
    _agree(dev.mean(1).real, host.mean(1).real)
    # the step after an accumulation is unaffected by the scratch use
    ref = pkg.BinaryLBM(nx, ny, nz, params=pkg.default_params(kBT=1e-5, alpha0=1.0, tau_f=1.0, tau_g=1.0))
    ref.LBM_init_mixture(); ref.LBM_timestep(36)
    lbm.LBM_timestep(1)
    assert np.array_equal(lbm.populations()[0], ref.populations()[0])
    dev.close(); lbm.close(); ref.close()


def test_device_structure_factor_of_hydrovsbar_and_errors(pkg):
    lbm = pkg.BinaryLBM(8, 8, 8, params=pkg.default_params(kBT=1e-5, alpha0=0.0))
    lbm.LBM_init_mixture(); lbm.LBM_timestep(10)
    names = pkg.plotfile.variable_names(9)
    dev = pkg.structfact.DeviceStructFact(lbm, names, lb_hydrovars=True)       # pairs within the 9 names only
    host = pkg.structfact.StructFact(names)
    dev.fort_structure(); host.fort_structure(lbm.LBM_hydrovars_density(), 0)
    _agree(dev.mean(1).real, host.mean(1).real)
    dev.close()
    # closing the lattice first releases the accumulator that lives on it
    dev2 = pkg.structfact.DeviceStructFact(lbm, names, lb_hydrovars=True)
    lbm.close()
    assert dev2._h is None
    dev2.close()
    lbm = pkg.BinaryLBM(8, 8, 8)
    slab = pkg.BinaryLBM(8, 8, 8, z0=0, z1=4, rank=0, nranks=2)
    with pytest.raises(pkg.BflbmError, match="whole lattice"):
        pkg.structfact.DeviceStructFact(slab, pkg.plotfile.variable_names(22))
    slab.close(); lbm.close()


def test_job_with_device_structure_factors_writes_the_same_files(pkg, tmp_path):
    pf = pkg.plotfile
    args = ["--system", "mixture", "--nx", "16", "--alpha0", "0", "--kbt", "1e-5", "--tau", "1", "--nsteps", "60",
            "--plot-int", "0", "--print-int", "0", "--plot-sf-window", "40", "--out-sf-step", "10"]
    assert pkg.run_job.main(args + ["--root", str(tmp_path / "H")]) == 0
    assert pkg.run_job.main(args + ["--root", str(tmp_path / "D"), "--sf-device"]) == 0
    rel = os.path.join("data_mixture_hydrovars", "lbm_data_shshan_alpha0_0.00_xi_1.0e-05_size16-16-16_continue")
    for name in ("plt_SF_mag000000060", "plt_SF_real_imag000000060"):
        h, hh = pf.read_plotfile(os.path.join(str(tmp_path / "H"), rel, name))
        d, dh = pf.read_plotfile(os.path.join(str(tmp_path / "D"), rel, name))
        assert hh["names"] == dh["names"] and h.shape == d.shape
        _agree(d, h)


@pytest.mark.parametrize("nslabs,n", [(2, (16, 16, 16)), (3, (12, 10, 14)), (3, (20, 9, 25)), (4, (16, 12, 32))])
def test_structure_factor_on_a_slab_decomposed_lattice(pkg, nslabs, n):
    """main_run_job.cpp:301-310, :342-349 on a lattice split into z-slabs (configs[4]'s validation): the distributed
    accumulator (2-D transforms per plane, transpose over the slabs, z transforms of row blocks) equals the
    single-context one (3-D hipFFT) to 1e-11 of the pair's largest |S|, and the host (numpy) definition."""
    nx, ny, nz = n
    par = pkg.default_params(kBT=1e-5, alpha0=1.0, tau_f=1.0, tau_g=1.0)
    names = pkg.plotfile.variable_names(22)
    one = pkg.BinaryLBM(nx, ny, nz, params=par)
    ring = pkg.RingLBM(nx, ny, nz, nslabs=nslabs, params=par)
    one.LBM_init_mixture(); ring.LBM_init_mixture()
    d1 = pkg.structfact.DeviceStructFact(one, names)
    dr = pkg.structfact.DeviceStructFact(ring, names)
    host = pkg.structfact.StructFact(names)
    one.LBM_timestep(12); ring.LBM_timestep(12)
    for _ in range(3):
        one.LBM_timestep(4); ring.LBM_timestep(4)
        d1.fort_structure(); dr.fort_structure()
        host.fort_structure(ring.LBM_hydrovars(), 0)
    assert dr.nsamples == d1.nsamples == 3
    for zero_avg in (1, 0):
        a, b, h = d1.mean(zero_avg), dr.mean(zero_avg), host.mean(zero_avg)
        _agree(b.real, a.real); _agree(b.imag, a.imag)
        _agree(b.real, h.real); _agree(b.imag, h.imag)
        _agree(dr.magnitude(zero_avg), np.abs(h))
    dr.fort_structure(reset=1); d1.fort_structure(reset=1)
    _agree(dr.mean(1).real, d1.mean(1).real)
    # the accumulation used the slabs' scratch buffers only: the next step is unaffected
    one.LBM_timestep(1); ring.LBM_timestep(1)
    assert np.array_equal(one.populations()[0], ring.populations()[0])
    dr.close(); d1.close(); one.close(); ring.close()
