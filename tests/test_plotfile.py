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
    assert cell_h[:5] == ["1", "1", "2", "0", "(8 0"]     # version, how = VisMF::NFiles, ncomp, ngrow
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


def test_structfact_accumulator(pkg, tmp_path):
    """StructFact re-statement: pair naming and layout as read by Mixture.ipynb cell 2, normalisation
    S = a^ conj(b^)/N (Parseval: mean over k of S_aa = <a^2>), k = 0 at the box centre and removed."""
    sfm, pf = pkg.structfact, pkg.plotfile
    rng = np.random.default_rng(3)
    names = pf.variable_names(22)
    sf = sfm.StructFact(names)
    assert sf.pair_names()[0] == "struct_fact_rho_rho" and "struct_fact_ugx_ufx" in sf.pair_names() \
        and "struct_fact_phi_rho" in sf.pair_names() and len(sf.pair_names()) == 22
    n = 8
    frames = [rng.standard_normal((22, n, n, n)) for _ in range(5)]
    for fr in frames:
        sf.fort_structure(fr)
    s = sf.write_plotfile(600500, 600500.0, str(tmp_path / "plt_SF"), zero_avg=1, max_grid_size=4)
    mag, hdr = pf.read_plotfile(str(tmp_path / "plt_SF_mag000600500"))          # 9-digit step like the notebook path
    assert hdr["names"] == sf.pair_names() and np.array_equal(mag, np.abs(s))
    ri, hdr = pf.read_plotfile(str(tmp_path / "plt_SF_real_imag000600500"))
    assert hdr["names"][0] == "struct_fact_rho_rho_real" and hdr["names"][22] == "struct_fact_rho_rho_imag"
    assert np.array_equal(ri[:22], s.real) and np.array_equal(ri[22:], s.imag)
    assert np.all(mag[:, n // 2, n // 2, n // 2] == 0)                           # zero_avg
    head = open(str(tmp_path / "plt_SF_mag000600500" / "Header")).read().split("\n")
    assert head[2 + 22 + 3].split() == ["-4.5", "-4.5", "-4.5"] and head[2 + 22 + 4].split() == ["3.5", "3.5", "3.5"]
    # Parseval on the rho-rho pair (without removing k = 0)
    full = sf.mean(zero_avg=0)[0]
    want = np.mean([(fr[0] ** 2).mean() for fr in frames])
    assert abs(full.real.mean() - want) < 1e-12
    # cross pair ugx-ufx is the conjugate-symmetric product a^ conj(b^): A = ufx (2), B = ugx (6)
    k = sf.pair_names().index("struct_fact_ugx_ufx")
    a, b = np.fft.fftn(frames[0][2]), np.fft.fftn(frames[0][6])
    one = sfm.StructFact(names); one.fort_structure(frames[0])
    assert np.allclose(np.fft.ifftshift(one.mean(0)[k]), a * np.conj(b) / n ** 3)
