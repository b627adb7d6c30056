"""CPU tests of the boundary: the C-ABI library loads without a GPU, exports every symbol that
include/bflbm.h declares, mirrors the reference's globals, validates its arguments, and the
product package neither falls back to a CPU path nor touches the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "binary-fluctuating-lattice-boltzmann_amd")


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bflbm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bflbm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    names = _declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/bflbm.h but not exported"
        assert n in pkg._lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(pkg._lib.SIGNATURES) == names


def test_default_params_mirror_reference_globals(pkg):
    p = pkg.default_params()
    # LBM_binary.H:17-30, LBM_d3q19.H:6-10
    assert (p.tau_f, p.tau_g, p.alpha0, p.alpha1, p.kappa, p.kBT) == (0.5, 0.5, 4.0, 0.0, 4.0, 0.0)
    assert p.cs2 == 1.0 / 3.0 and (p.rho_lo, p.rho_hi) == (0.0, 1.0) and p.seed == 12345
    with pytest.raises(AttributeError):
        pkg.default_params(not_a_parameter=1.0)
    assert pkg._lib.load().bflbm_abi_version() == 1


def test_host_rng_equals_oracle_stream(pkg, ob):
    """The product's counter-based Gaussian stream (csrc/bflbm_rng.h, host build) against the oracle's
    independent restatement, bit for bit."""
    rng = np.random.default_rng(3)
    for _ in range(300):
        seed = int(rng.integers(0, 2 ** 63))
        site = int(rng.integers(0, 2 ** 40))
        idx = int(rng.integers(0, 2 ** 31))
        assert np.array_equal(pkg.rng_site_normals(seed, site, idx), ob.site_normals(seed, site, idx))


def test_create_rejects_bad_domains(pkg):
    lib = pkg._lib.load()
    p = pkg.default_params()

    def create(n, z0, z1, rank, nranks):
        d = pkg.Domain()
        d.n[0], d.n[1], d.n[2] = n
        d.z0, d.z1, d.rank, d.nranks, d.device = z0, z1, rank, nranks, 0
        h = ctypes.c_void_p()
        rc = lib.bflbm_create(ctypes.byref(p), ctypes.byref(d), ctypes.byref(h))
        msg = lib.bflbm_last_error().decode()
        if rc == 0:
            lib.bflbm_destroy(h)
        return rc, msg

    for args in [((0, 8, 8), 0, 8, 0, 1), ((8, 8, 8), 0, 4, 0, 1), ((8, 8, 8), 4, 2, 0, 2),
                 ((8, 8, 8), 0, 2, 0, 4), ((8, 8, 8), 0, 8, 3, 2), ((8, 8, 8), 0, 9, 0, 1)]:
        rc, msg = create(*args)
        assert rc != 0 and msg, args
    assert lib.bflbm_create(None, None, None) != 0
    assert lib.bflbm_step(None, 1) != 0
    assert b"null" in lib.bflbm_last_error()


def test_missing_extension_fails_loudly():
    """No silent CPU fallback: with the HIP library absent the product import path raises."""
    code = ("import __graft_entry__ as g; p = g.load_package(); "
            "p._lib.load()")
    env = dict(os.environ, BFLBM_LIB="/nonexistent/libbflbm.so", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr


def test_product_never_references_the_oracle():
    """oracle/ is test infrastructure: nothing under the package may import, link or name it."""
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".h", ".H", ".hip", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for m in re.finditer(r"liboracle|oracle_binding|orc_[a-z_]+\(|oracle/", text):
                    bad.append((f, m.group(0)))
    assert not bad, bad


def test_fab_descriptor(pkg):
    f = pkg.make_fab((-2, -2, -2), (5, 5, 5), (0, 0, 0), (3, 3, 3))
    assert list(f.lo) == [-2, -2, -2] and list(f.vhi) == [3, 3, 3]
    f = pkg.make_fab((0, 0, 4), (7, 7, 7))
    assert list(f.vlo) == [0, 0, 4] and list(f.vhi) == [7, 7, 7]


def test_slab_bounds(pkg):
    assert [pkg.slab_bounds(256, 8, r) for r in range(8)] == [(32 * r, 32 * (r + 1)) for r in range(8)]
    b = [pkg.slab_bounds(10, 3, r) for r in range(3)]
    assert b[0][0] == 0 and b[-1][1] == 10 and all(b[i][1] == b[i + 1][0] for i in range(2))


def test_bench_contract_fields():
    """bench.py must print the contract's keys; checked statically here (no GPU)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "workload"]:
        assert f'"{key}"' in src, key


def test_contract_sentence_of_the_default_schedule_is_the_same_everywhere():
    """the contract of the default (hand-over) schedule is stated once, in the same words, in the header, DESIGN.md,
    INTEGRATION.md and README.md (tests/test_gpu_handover_oracle.py has one test per clause)."""
    import re
    sentence = ("the first step after an init or upload equals the CPU reference path bit for bit; after that the results differ from it by what a one-ulp change of the reference's own state does: in a well-conditioned run rho, phi, rho + phi agree to 1e-12 relative and the velocities to 1e-12 max(cs, |u|) absolute at every site, except at near-vacuum sites (a density below 1e-3 of its field's maximum), where the bound holds for the density relative to the field maximum and for the momentum; in a run through a violent transient (spinodal demixing: velocities of thousands of lattice units at near-vacuum sites) the difference is bounded by that run's own one-ulp response and by nothing smaller")
    norm = lambda t: re.sub(r"\s+", " ", re.sub(r"\n \* ", " ", t.replace("*", "")))
    for rel in ("include/bflbm.h", "DESIGN.md", "INTEGRATION.md", "README.md"):
        assert norm(sentence) in norm(open(os.path.join(ROOT, rel)).read()), rel
