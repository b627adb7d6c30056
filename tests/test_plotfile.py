"""CPU: AMReX single-level plotfile writer/reader: round trip, on-disk layout,
the names the reference's notebooks look up, the one-name/19-component checkpoint quirk."""
import os

import numpy as np
import pytest


def test_roundtrip_multibox(pkg, tmp_path):
    pf = pkg.plotfile
    rng = np.random.default_rng(0)
    data = rng.standard_normal((22, 12, 10, 16))
    name = pf.concatenate(str(tmp_path / "plt"), 200)
    assert name.endswith("plt0000200")                      # amrex::Concatenate(root, step, 7)
    pf.write_plotfile(name, data, pf.variable_names(22), time=200.0, step=200, max_grid_size=8)
    back, hdr = pf.read_plotfile(name)
    assert np.array_equal(back, data)
    assert hdr["names"] == pf.variable_names(22) and hdr["step"] == 200 and hdr["time"] == 200.0
    assert hdr["n"] == (16, 10, 12) and hdr["ngrids"] == 2 * 2 * 2 == len(hdr["boxes"])
    assert hdr["boxes"][0] == ([0, 0, 0], [7, 7, 7]) and hdr["boxes"][1] == ([8, 0, 0], [15, 7, 7])


def test_on_disk_layout(pkg, tmp_path):
    pf = pkg.plotfile
    data = np.arange(2 * 4 * 4 * 4, dtype=float).reshape(2, 4, 4, 4)
    name = str(tmp_path / "plt0000001")
    pf.write_plotfile(name, data, ["rho", "phi"], time=1.0, step=1, max_grid_size=2)
    head = open(os.path.join(name, "Header")).read().split("\n")
    assert head[0] == "HyperCLaw-V1.1" and head[1] == "2" and head[2:4] == ["rho", "phi"]
    assert head[4] == "3" and head[5] == "1" and head[6] == "0"
    assert head[10].strip() == "((0,0,0) (3,3,3) (0,0,0))"
    assert head[12].split() == ["0.25", "0.25", "0.25"]
    assert head[15] == "0 8 1" and head[-2] == "Level_0/Cell"
    cell_h = open(os.path.join(name, "Level_0", "Cell_H")).read().split("\n")
    assert cell_h[:5] == ["1", "0", "2", "0", "(8 0"]
    assert cell_h[5] == "((0,0,0) (1,1,1) (0,0,0))"
    fod = [ln for ln in cell_h if ln.startswith("FabOnDisk: Cell_D_00000 ")]
    assert len(fod) == 8 and fod[0].endswith(" 0")
    raw = open(os.path.join(name, "Level_0", "Cell_D_00000"), "rb").read()
    rec = b"FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))((0,0,0) (1,1,1) (0,0,0)) 2\n"
    assert raw.startswith(rec)
    first = np.frombuffer(raw[len(rec):len(rec) + 16 * 8], dtype="<f8").reshape(2, 2, 2, 2)
    assert np.array_equal(first, data[:, :2, :2, :2])
    assert int(fod[1].split()[-1]) == len(rec) + 16 * 8


def test_names(pkg):
    pf = pkg.plotfile
    # field list printed by yt in Flat_Interface.ipynb cell 4 (sorted there)
    assert sorted(pf.variable_names(22)) == sorted(
        ["afx", "afy", "afz", "agx", "agy", "agz", "nfbarx", "ngbarx", "p_bulk", "phi", "rho", "ubx", "uby", "ubz",
         "ufbarx", "ufx", "ufy", "ufz", "ugbarx", "ugx", "ugy", "ugz"])
    assert pf.variable_names(22)[5] == "p_bulk" and pf.variable_names(15)[-1] == "agz"
    assert pf.noise_names("f")[:2] == ["fa0", "fa1"] and pf.noise_names("g")[18] == "ga18"


def test_checkpoint_quirk_one_name_many_components(pkg, tmp_path):
    """main_run_job.cpp:406-409 writes fold (19 comps) with the single name 'rho_chk'; the loader reads
    Level_0/Cell and never looks at the Header (AMReX_FileIO.H:30)."""
    pf = pkg.plotfile
    f = np.random.default_rng(1).random((19, 6, 6, 6))
    name = str(tmp_path / "f_checkpoint0000010_alpha0_4.00_xi_0.0e+00_size6-6-6")
    pf.write_plotfile(name, f, ["rho_chk"], max_grid_size=3)
    back, hdr = pf.read_plotfile(name)
    assert hdr["names"] == ["rho_chk"] and hdr["ncomp"] == 19
    assert np.array_equal(back, f)
