"""ctypes binding of the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")
Q = 19


class OParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in
                "tau_f tau_g alpha0 alpha1 kappa kBT cs2 cs4 rho_lo rho_hi".split()] + [("seed", ctypes.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "bflbm_oracle.c")
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
            subprocess.run(["make", "-C", ORACLE_DIR], check=True)
        _lib = ctypes.CDLL(LIB)
        _lib.orc_bench.restype = ctypes.c_double
    return _lib


def default_params(**kw):
    p = OParams()
    lib().orc_default_params(ctypes.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    if "cs2" in kw and "cs4" not in kw:
        p.cs4 = p.cs2 * p.cs2
    return p


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def moments(f):
    f = np.ascontiguousarray(f, dtype=np.float64)
    m = np.empty(Q)
    lib().orc_moments(_p(f), _p(m))
    return m


def populations(m):
    m = np.ascontiguousarray(m, dtype=np.float64)
    f = np.empty(Q)
    lib().orc_populations(_p(m), _p(f))
    return f


def lattice_tables():
    c = np.empty((Q, 3), dtype=np.int32)
    w = np.empty(Q)
    b = np.empty(Q)
    lib().orc_lattice(_p(c), _p(w), _p(b))
    return c, w, b


def site_normals(seed, site, noise_index):
    out = np.empty(36)
    lib().orc_site_normals(ctypes.c_uint64(seed), ctypes.c_uint64(site), ctypes.c_uint32(noise_index), _p(out))
    return out[:33].copy()


class OracleLattice:
    """The reference's MultiFabs for one periodic box and its LBM_* operators (oracle side)."""

    def __init__(self, nx, ny, nz, params=None):
        self.n = (nx, ny, nz)
        self.p = params if params is not None else default_params()
        shp = (nz, ny, nx)
        self.f = np.zeros((Q,) + shp)
        self.g = np.zeros((Q,) + shp)
        self._ft = np.zeros((Q,) + shp)
        self._gt = np.zeros((Q,) + shp)
        self.fn = np.zeros((Q,) + shp)
        self.gn = np.zeros((Q,) + shp)
        self.hbar = np.zeros((15,) + shp)
        self.h = np.zeros((22,) + shp)
        self.steps = 0
        self.ref = None            # USE_REF_STATE: (rho_eq, phi_eq, rhot_eq, com_ref)

    def set_ref_state(self, rho_eq, phi_eq, rhot_eq, com_ref):
        """Switch to the reference's USE_REF_STATE build (LBM_binary.H:12, :92-107)."""
        shp = self.f.shape[1:]
        self.ref = tuple(np.ascontiguousarray(a, dtype=np.float64).reshape(shp) for a in (rho_eq, phi_eq, rhot_eq)) + \
                   (np.ascontiguousarray(com_ref, dtype=np.float64),)

    def _refresh_ref(self, rel):
        nx, ny, nz = self.n
        rel = np.ascontiguousarray(rel, dtype=np.float64)
        lib().orc_refresh_ref(ctypes.byref(self.p), nx, ny, nz, ctypes.c_uint32(self.steps),
                              _p(self.f), _p(self.g), _p(self.hbar), _p(self.fn), _p(self.gn), _p(self.h),
                              _p(self.ref[0]), _p(self.ref[1]), _p(self.ref[2]), _p(rel))

    def _dims(self):
        return self.n

    def refresh(self, rel_kind="relative"):
        """densities -> noise -> hydrovars.  With a reference state the position handed to
        thermal_noise depends on the caller: LBM_init -> COM relative to com_ref (:651-654),
        LBM_init_mixture -> absolute COM (:623-625), LBM_init_stripe/_droplet -> zero (:690, :739)."""
        nx, ny, nz = self.n
        if self.ref is not None:
            if rel_kind == "zero":
                rel = np.zeros(3)
            else:
                lib().orc_hydrovars_density(ctypes.byref(self.p), nx, ny, nz, _p(self.f), _p(self.g), _p(self.hbar))
                rel = self.com() - (self.ref[3] if rel_kind == "relative" else 0.0)
            return self._refresh_ref(rel)
        lib().orc_refresh(ctypes.byref(self.p), nx, ny, nz, ctypes.c_uint32(self.steps),
                          _p(self.f), _p(self.g), _p(self.hbar), _p(self.fn), _p(self.gn), _p(self.h))

    def init_mixture(self):
        nx, ny, nz = self.n
        lib().orc_init_mixture(ctypes.byref(self.p), nx, ny, nz, _p(self.f), _p(self.g))
        self.steps = 0
        self.refresh("absolute")

    def init_stripe(self, frac):
        nx, ny, nz = self.n
        lib().orc_init_stripe(ctypes.byref(self.p), nx, ny, nz, ctypes.c_double(frac), _p(self.f), _p(self.g))
        self.steps = 0
        self.refresh("zero")

    def init_droplet(self, r):
        nx, ny, nz = self.n
        lib().orc_init_droplet(ctypes.byref(self.p), nx, ny, nz, ctypes.c_double(r), _p(self.f), _p(self.g))
        self.steps = 0
        self.refresh("zero")

    def init_from(self, f0, g0):
        self.f[...] = f0
        self.g[...] = g0
        self.steps = 0
        self.refresh()

    def timestep(self):
        nx, ny, nz = self.n
        if self.ref is not None:
            lib().orc_timestep_ref(ctypes.byref(self.p), nx, ny, nz, ctypes.c_uint32(self.steps),
                                   _p(self.f), _p(self.g), _p(self._ft), _p(self._gt),
                                   _p(self.hbar), _p(self.fn), _p(self.gn), _p(self.h),
                                   _p(self.ref[0]), _p(self.ref[1]), _p(self.ref[2]), _p(self.ref[3]))
            self.steps += 1
            return
        lib().orc_timestep(ctypes.byref(self.p), nx, ny, nz, ctypes.c_uint32(self.steps),
                           _p(self.f), _p(self.g), _p(self._ft), _p(self._gt),
                           _p(self.hbar), _p(self.fn), _p(self.gn), _p(self.h))
        self.steps += 1

    def timestep_injected(self, fn, gn, fn_next=None, gn_next=None):
        """Collide with the given noise moments, stream, then refresh with fn_next/gn_next
        (or zeros) as the new noise so that hydrovs stays consistent."""
        nx, ny, nz = self.n
        # hydrovs must be consistent with the injected noise (hydrovars reads modes 1..3)
        self.fn[...] = fn
        self.gn[...] = gn
        lib().orc_hydrovars(ctypes.byref(self.p), nx, ny, nz, _p(self.f), _p(self.g), _p(self.hbar),
                            _p(self.fn), _p(self.gn), _p(self.h))
        lib().orc_collide_stream(ctypes.byref(self.p), nx, ny, nz, _p(self.f), _p(self.g),
                                 _p(self._ft), _p(self._gt), _p(self.h), _p(self.fn), _p(self.gn))
        self.steps += 1
        lib().orc_hydrovars_density(ctypes.byref(self.p), nx, ny, nz, _p(self.f), _p(self.g), _p(self.hbar))
        self.fn[...] = 0.0 if fn_next is None else fn_next
        self.gn[...] = 0.0 if gn_next is None else gn_next
        lib().orc_hydrovars(ctypes.byref(self.p), nx, ny, nz, _p(self.f), _p(self.g), _p(self.hbar),
                            _p(self.fn), _p(self.gn), _p(self.h))

    def com(self):
        nx, ny, nz = self.n
        out = np.empty(3)
        lib().orc_update_com(nx, ny, nz, _p(self.hbar), _p(out))
        return out


def bench(nx, ny, nz, nsteps, params=None):
    p = params if params is not None else default_params()
    chk = ctypes.c_double()
    secs = lib().orc_bench(ctypes.byref(p), nx, ny, nz, nsteps, ctypes.byref(chk))
    return secs, chk.value
