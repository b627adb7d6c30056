"""Stand-in slab engine for the CPU test-suite (test infrastructure only).

Re-states, in numpy on top of the CPU oracle, the semantics of one HIP slab context
(csrc/bflbm.hip): post-collision storage S[c][p][y][x] with 2 halo planes per side, the
(component, plane) halo tables, the split step, upload/commit.  It lets the multi-rank protocol
of slab.SlabLattice run without a GPU (gloo, world_size 2 and 3) and be compared bit for bit with
the single-box oracle.  The real tables live in C++ and are checked on the GPU by
tests/test_gpu_slabs.py.
"""
import ctypes

import numpy as np

import oracle_binding as ob

Q = 19
HALO_STATE, HALO_NEXT, HALO_UPLOAD = 0, 1, 2


def _buf(ptr, n):
    return np.ctypeslib.as_array((ctypes.c_double * n).from_address(ptr))


class StandinEngine:
    def __init__(self, nx, ny, nz, z0, z1, rank, nranks, params=None):
        self.n = (nx, ny, nz)
        self.z0, self.z1, self.rank, self.nranks = z0, z1, rank, nranks
        self.nzl = z1 - z0
        self.H = 0 if nranks == 1 else 2
        self.nzs = self.nzl + 2 * self.H
        self.p = params if params is not None else ob.default_params()
        self.S = [np.zeros((2 * Q, self.nzs, ny, nx)), np.zeros((2 * Q, self.nzs, ny, nx))]
        self.cur = 0
        self.steps = 0
        self.c, self.w, self.b = ob.lattice_tables()
        self.open = False

    # ---- geometry helpers
    def _gz(self, p):
        return (self.z0 - self.H + p) % self.n[2]

    def _pulled(self, S, planes):
        """natural populations f_i(x) = S_i(x - c_i) on the given storage planes."""
        nx, ny, nz = self.n
        out = np.empty((2 * Q, len(planes), ny, nx))
        for k, p in enumerate(planes):
            for i in range(Q):
                cx, cy, cz = (int(v) for v in self.c[i])
                src = (p - cz) % self.nzs if self.H == 0 else p - cz
                out[i, k] = np.roll(S[i, src], (cy, cx), axis=(0, 1))
                out[i + Q, k] = np.roll(S[i + Q, src], (cy, cx), axis=(0, 1))
        return out

    # ---- halo tables (mirror of halo_table in csrc/bflbm.hip)
    def _table(self, kind, side, pack):
        H, nzl = self.H, self.nzl
        if kind == HALO_UPLOAD:
            plane = (H if side == 0 else H + nzl - 1) if pack else (H - 1 if side == 0 else H + nzl)
            return [(k, plane) for k in range(2 * Q)]
        if pack:
            d = -1 if side == 0 else +1
            near, far = (H, H + 1) if side == 0 else (H + nzl - 1, H + nzl - 2)
        else:
            d = +1 if side == 0 else -1
            near, far = (H - 1, H - 2) if side == 0 else (H + nzl, H + nzl + 1)
        t = []
        for fl in range(2):
            t += [(fl * Q + i, near) for i in range(Q) if self.c[i, 2] in (0, d)]
            t += [(fl * Q + i, far) for i in range(Q) if self.c[i, 2] == d]
        assert len(t) == 2 * Q
        return t

    def halo_bytes(self, kind=0):
        return 2 * Q * self.n[0] * self.n[1] * 8

    def _halo_array(self, kind):
        return self.S[self.cur] if kind == HALO_STATE else self.S[1 - self.cur]

    def halo_plane_tensors(self, kind, side, pack, device=None):
        """Staging-free exchange: the 38 component planes of a face as torch tensors that alias the state arrays."""
        import torch
        if getattr(self, "no_direct", False):
            raise AttributeError("halo_plane_tensors")
        A = self._halo_array(kind)
        return [torch.from_numpy(A[comp, p]).reshape(-1) for comp, p in self._table(kind, side, pack)]

    def halo_pack(self, kind, side, ptr):
        plane = self.n[0] * self.n[1]
        buf = _buf(ptr, 2 * Q * plane).reshape(2 * Q, self.n[1], self.n[0])
        A = self._halo_array(kind)
        for e, (comp, p) in enumerate(self._table(kind, side, True)):
            buf[e] = A[comp, p]

    def halo_unpack(self, kind, side, ptr):
        plane = self.n[0] * self.n[1]
        buf = _buf(ptr, 2 * Q * plane).reshape(2 * Q, self.n[1], self.n[0])
        A = self._halo_array(kind)
        for e, (comp, p) in enumerate(self._table(kind, side, False)):
            A[comp, p] = buf[e]

    # ---- oracle-backed field evaluation on planes [H-1, H+nzl+1) (or the whole box when H == 0)
    def _fields(self):
        nx, ny, nz = self.n
        if self.H == 0:
            planes = list(range(self.nzs))
            gz0 = 0
        else:
            planes = list(range(self.H - 1, self.H + self.nzl + 1))
            gz0 = self.z0 - 1
        nat = self._pulled(self.S[self.cur], planes)
        f = np.ascontiguousarray(nat[:Q])
        g = np.ascontiguousarray(nat[Q:])
        nzb = len(planes)
        hbar = np.zeros((15, nzb, ny, nx))
        fn = np.zeros((Q, nzb, ny, nx))
        gn = np.zeros((Q, nzb, ny, nx))
        h = np.zeros((22, nzb, ny, nx))
        L = ob.lib()
        pr = ctypes.byref(self.p)
        L.orc_hydrovars_density(pr, nx, ny, nzb, ob._p(f), ob._p(g), ob._p(hbar))
        L.orc_thermal_noise_slab(pr, nx, ny, nzb, gz0, nz, ob._p(hbar), ctypes.c_uint32(self.steps), ob._p(fn), ob._p(gn))
        L.orc_hydrovars(pr, nx, ny, nzb, ob._p(f), ob._p(g), ob._p(hbar), ob._p(fn), ob._p(gn), ob._p(h))
        return f, g, hbar, fn, gn, h

    def _own(self, a):
        return a if self.H == 0 else a[:, 1:-1]

    # ---- stepping
    def step_boundary(self):
        assert not self.open
        self.open = True
        nx, ny, nz = self.n
        f, g, hbar, fn, gn, h = self._fields()
        ob.lib().orc_collide(ctypes.byref(self.p), nx, ny, f.shape[1], ob._p(f), ob._p(g), ob._p(h), ob._p(fn), ob._p(gn))
        D = self.S[1 - self.cur]
        D[:Q, self.H:self.H + self.nzl] = self._own(f)
        D[Q:, self.H:self.H + self.nzl] = self._own(g)

    def step_interior(self):
        assert self.open

    def step_finish(self):
        assert self.open
        self.cur = 1 - self.cur
        self.steps += 1
        self.open = False

    # ---- initial conditions
    def _init_from_global(self, fG, gG):
        for p in range(self.nzs):
            for i in range(Q):
                cx, cy, cz = (int(v) for v in self.c[i])
                gz = (self._gz(p) + cz) % self.n[2]
                self.S[self.cur][i, p] = np.roll(fG[i, gz], (-cy, -cx), axis=(0, 1))
                self.S[self.cur][i + Q, p] = np.roll(gG[i, gz], (-cy, -cx), axis=(0, 1))
        self.steps = 0

    def _init(self, name, *args):
        ref = ob.OracleLattice(*self.n, params=self.p)
        nx, ny, nz = self.n
        fn = getattr(ob.lib(), "orc_init_" + name)
        cargs = [ctypes.c_double(a) for a in args]
        fn(ctypes.byref(self.p), nx, ny, nz, *cargs, ob._p(ref.f), ob._p(ref.g))
        self._init_from_global(ref.f, ref.g)

    def LBM_init_mixture(self):
        self._init("mixture")

    def LBM_init_stripe(self, frac):
        self._init("stripe", frac)

    def LBM_init_droplet(self, r):
        self._init("droplet", r)

    def upload(self, f0, g0, fab=None):
        N = self.S[1 - self.cur]
        N[:Q, self.H:self.H + self.nzl] = f0
        N[Q:, self.H:self.H + self.nzl] = g0

    def commit_upload(self, reset=True):
        N, S = self.S[1 - self.cur], self.S[self.cur]
        for p in range(self.H, self.H + self.nzl):
            for i in range(Q):
                cx, cy, cz = (int(v) for v in self.c[i])
                src = (p + cz) % self.nzs if self.H == 0 else p + cz
                S[i, p] = np.roll(N[i, src], (-cy, -cx), axis=(0, 1))
                S[i + Q, p] = np.roll(N[i + Q, src], (-cy, -cx), axis=(0, 1))
        if reset:
            self.steps = 0

    # ---- outputs (own planes)
    def populations(self):
        f, g, *_ = self._fields()
        return self._own(f).copy(), self._own(g).copy()

    def LBM_hydrovars_density(self):
        return self._own(self._fields()[2])[:9].copy()

    def LBM_hydrovars(self):
        return self._own(self._fields()[5]).copy()

    def thermal_noise(self):
        _, _, _, fn, gn, _ = self._fields()
        return self._own(fn).copy(), self._own(gn).copy()

    def com_sums(self):
        hb = self.LBM_hydrovars_density()
        nx, ny, nz = self.n
        z = np.arange(self.z0, self.z1)[:, None, None]
        y = np.arange(ny)[None, :, None]
        x = np.arange(nx)[None, None, :]
        r = hb[0]
        return np.array([r.sum(), (r * x).sum(), (r * y).sum(), (r * z).sum()])

    def mass(self):
        hb = self.LBM_hydrovars_density()
        return float(hb[0].sum()), float(hb[1].sum())

    def sync(self):
        pass

    def close(self):
        pass
