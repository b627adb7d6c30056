"""Host-side mirror of the reference's LBM operator surface (LBM_binary.H) for one z-slab.

The method names, argument meaning and state semantics follow the reference's free
functions so that tests read like the reference's own driver (main_run_job.cpp:269-339):

    LBM_init_mixture / LBM_init_stripe(frac) / LBM_init_droplet(r) / LBM_init(f0, g0)
    LBM_timestep()                       LBM_binary.H:545-594
    LBM_hydrovars_density() -> hydrovsbar  :343-354
    thermal_noise() -> fnoisevs, gnoisevs  :73-132
    LBM_hydrovars() -> hydrovs             :298-313
    update_com()                           LBM_hydrovs.H:26-60

All arithmetic happens in the HIP library behind the C-ABI (include/bflbm.h); this file
only moves numpy arrays in AMReX FAB layout (component slowest, x fastest) across it.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import Domain, Fab, Params, NVEL, NHYDRO, NHYDROBAR, check


def default_params(**overrides):
    """The reference's shipped globals (LBM_binary.H:17-30, LBM_d3q19.H:6-10)."""
    p = Params()
    _lib.load().bflbm_default_params(ctypes.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(f"unknown model parameter {k!r}")
        setattr(p, k, v)
    return p


def make_fab(lo, hi, vlo=None, vhi=None):
    """bflbm_fab for a host array allocated over cells lo..hi with valid region vlo..vhi."""
    f = Fab()
    vlo = lo if vlo is None else vlo
    vhi = hi if vhi is None else vhi
    for d in range(3):
        f.lo[d], f.hi[d], f.vlo[d], f.vhi[d] = int(lo[d]), int(hi[d]), int(vlo[d]), int(vhi[d])
    return f


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check_array(a, ncomp, shape3, name):
    if not isinstance(a, np.ndarray) or a.dtype != np.float64 or not a.flags.c_contiguous:
        raise TypeError(f"{name}: need a C-contiguous float64 numpy array")
    if a.shape != (ncomp,) + tuple(shape3):
        raise ValueError(f"{name}: shape {a.shape}, expected {(ncomp,) + tuple(shape3)}")


class _DevicePlane:
    """A component plane of the resident state as seen by torch.as_tensor (no copy)."""

    def __init__(self, ptr, ndoubles):
        self.__cuda_array_interface__ = {"shape": (int(ndoubles),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class _DropletMixin:
    """Droplet observables reduced on the device (csrc/bflbm_droplet.h); `_droplet_fn` names the C-ABI pair."""

    def droplet_moments(self):
        """20 raw mass moments of rho (see include/bflbm.h); feed analysis.*_from_moments."""
        m = (ctypes.c_double * 20)()
        check(getattr(self.lib, self._droplet_fn[0])(self._h, m))
        return np.array(list(m))

    def fit_droplet(self, r0=None, p0=None, max_iter=200, tol=1e-12):
        """(hi, lo, R, W) of the tanh profile about r0 (default: the centre of mass), unit-box coordinates.
        Start values default to the notebook's (max rho, min rho, 0.5, 0.5) with max/min replaced by the
        model's natural bounds rho_hi, rho_lo of the parameters."""
        from . import analysis
        if r0 is None:
            r0 = analysis.com_from_moments(self.droplet_moments(), self.n)
        if p0 is None:
            p0 = (float(self.params.rho_hi), float(self.params.rho_lo), 0.5, 0.5)
        r = (ctypes.c_double * 3)(*[float(v) for v in r0])
        p = (ctypes.c_double * 4)(*[float(v) for v in p0])
        cost, it = ctypes.c_double(), ctypes.c_int()
        check(getattr(self.lib, self._droplet_fn[1])(self._h, r, p, int(max_iter), float(tol), ctypes.byref(cost), ctypes.byref(it)))
        self.last_fit = dict(cost=cost.value, iterations=it.value)
        return tuple(p)

    def fit_droplet_flow(self, **opts):
        """(W, R, undulation) of the reference's own gradient-flow fit (fittingDropletParams, LBM_hydrovs.H:160-213):
        rho ~ 1/2 (1 + tanh((R - |r - r0|) / sqrt(2 W))) in unit-box coordinates.  Keyword options are the fields of
        bflbm_flowfit_opts (W0, R0, eta_W, eta_R, dt, undul_ratio, nstep, step_window, max_retry); the driver's call is
        fit_droplet_flow(step_window=20, undul_ratio=0.01, nstep=400, W0=kappa, R0=radius) (main_run_job.cpp:365).
        Raises BflbmError where the reference throws (undulation out of bounds after the retries)."""
        o = _lib.FlowFitOpts()
        self.lib.bflbm_flowfit_default_opts(ctypes.byref(o))
        for k, v in opts.items():
            if not hasattr(o, k):
                raise AttributeError(f"unknown fit option {k!r}")
            setattr(o, k, v)
        res = (ctypes.c_double * 3)()
        retries = ctypes.c_int()
        check(getattr(self.lib, self._droplet_fn[2])(self._h, ctypes.byref(o), res, ctypes.byref(retries)))
        self.last_fit = dict(retries=retries.value)
        return tuple(res)


# kernel schedules of include/bflbm.h (bflbm_set_schedule): "fused" = plane march with the ring densities pulled
# (bit-exact), "handover" = plane march with the ring densities handed over from the previous step (tolerance)
SCHEDULES = {"two_pass": 0, "fused": 1, "fused_exact": 1, "auto": 2, "handover": 3}


class BinaryLBM(_DropletMixin):
    _droplet_fn = ("bflbm_droplet_moments", "bflbm_fit_droplet", "bflbm_fit_droplet_flow")

    """One z-slab [z0, z1) of a periodic nx*ny*nz D3Q19 binary-fluid lattice on one GPU."""

    def __init__(self, nx, ny=None, nz=None, params=None, z0=0, z1=None, rank=0, nranks=1,
                 device=0, schedule=None, stream=None):
        self.lib = _lib.load()
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        z1 = nz if z1 is None else z1
        self.n = (int(nx), int(ny), int(nz))
        self.z0, self.z1 = int(z0), int(z1)
        self.nzl = self.z1 - self.z0
        self.rank, self.nranks = int(rank), int(nranks)
        self.params = params if params is not None else default_params()
        dom = Domain()
        dom.n[0], dom.n[1], dom.n[2] = self.n
        dom.z0, dom.z1, dom.rank, dom.nranks, dom.device = self.z0, self.z1, self.rank, self.nranks, int(device)
        h = ctypes.c_void_p()
        check(self.lib.bflbm_create(ctypes.byref(self.params), ctypes.byref(dom), ctypes.byref(h)))
        self._h = h
        if stream is not None:      # an external hipStream_t handle; 0 is the legacy default stream
            check(self.lib.bflbm_set_stream(self._h, ctypes.c_void_p(stream), 1))
        if schedule is not None:
            self.set_schedule(schedule)

    @classmethod
    def _borrow(cls, handle, n, z0, z1, rank, nranks, params):
        """View on a context owned by someone else (a slab of a RingLBM); close() does not destroy it."""
        self = cls.__new__(cls)
        self.lib = _lib.load()
        self.n, self.z0, self.z1, self.nzl = tuple(n), int(z0), int(z1), int(z1) - int(z0)
        self.rank, self.nranks, self.params = int(rank), int(nranks), params
        self._h, self._borrowed = handle, True
        return self

    # -- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            for d in list(getattr(self, "_dependents", [])):   # e.g. structure-factor accumulators living on this context
                d.close()
            if not getattr(self, "_borrowed", False):
                self.lib.bflbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- configuration (the reference edits source-level globals) ---------------------------
    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(f"unknown model parameter {k!r}")
            setattr(self.params, k, v)
        check(self.lib.bflbm_set_params(self._h, ctypes.byref(self.params)))

    def set_schedule(self, schedule):
        code = SCHEDULES.get(schedule, schedule)
        check(self.lib.bflbm_set_schedule(self._h, int(code)))

    # -- shapes ----------------------------------------------------------------------------
    def resolved_schedule(self):
        """Name of the schedule the next step runs (auto resolved for the current parameters and lattice)."""
        v = ctypes.c_int()
        check(self.lib.bflbm_resolved_schedule(self._h, ctypes.byref(v)))
        return {0: "two_pass", 1: "fused", 3: "handover"}[v.value]

    @property
    def slab_shape(self):
        return (self.nzl, self.n[1], self.n[0])

    def slab_fab(self):
        """FAB descriptor of a ghost-free array covering exactly this slab."""
        lo = (0, 0, self.z0)
        hi = (self.n[0] - 1, self.n[1] - 1, self.z1 - 1)
        return make_fab(lo, hi)

    def _new(self, ncomp):
        return np.empty((ncomp,) + self.slab_shape, dtype=np.float64)

    # -- initial conditions ------------------------------------------------------------------
    def LBM_init_mixture(self):
        check(self.lib.bflbm_init_mixture(self._h))

    def LBM_init_stripe(self, frac):
        check(self.lib.bflbm_init_stripe(self._h, float(frac)))

    def LBM_init_droplet(self, r):
        check(self.lib.bflbm_init_droplet(self._h, float(r)))

    def upload(self, f0, g0, fab=None):
        """Stage populations of one box (mf.ParallelCopy(mf0), LBM_binary.H:643)."""
        if fab is None:
            _check_array(f0, NVEL, self.slab_shape, "f0")
            _check_array(g0, NVEL, self.slab_shape, "g0")
            fab = self.slab_fab()
        check(self.lib.bflbm_upload_fg(self._h, _ptr(f0), _ptr(g0), ctypes.byref(fab)))

    def commit_upload(self, reset_step_counter=True):
        check(self.lib.bflbm_commit_upload(self._h, int(bool(reset_step_counter))))

    def LBM_init(self, f0, g0, fab=None):
        """Continue from given populations (LBM_binary.H:632-661); single slab only --
        multi-slab callers use SlabLattice.LBM_init which exchanges the z faces."""
        if self.nranks != 1:
            raise _lib.BflbmError("BinaryLBM.LBM_init needs the slab driver when nranks > 1")
        self.upload(f0, g0, fab)
        self.commit_upload(True)

    # -- time stepping ---------------------------------------------------------------------
    def LBM_timestep(self, nsteps=1):
        check(self.lib.bflbm_step(self._h, int(nsteps)))

    def step_boundary(self):
        check(self.lib.bflbm_step_boundary(self._h))

    def step_interior(self):
        check(self.lib.bflbm_step_interior(self._h))

    def step_finish(self):
        check(self.lib.bflbm_step_finish(self._h))

    @property
    def steps_done(self):
        n = ctypes.c_longlong()
        check(self.lib.bflbm_step_count(self._h, ctypes.byref(n)))
        return n.value

    def set_steps_done(self, n):
        """Absolute step of the resident state = noise index of the next step (restart from a kBT > 0 checkpoint)."""
        check(self.lib.bflbm_set_step_count(self._h, int(n)))

    def placement_report(self):
        """What the placement tuning at creation measured (bflbm_tune_placement): ms per step of every candidate allocation
        tried and the index of the one kept; None when the context was never tuned (small lattices, BFLBM_PLACEMENT_CANDIDATES=1)."""
        ms = (ctypes.c_float * 8)()
        n, k = ctypes.c_int(), ctypes.c_int()
        check(self.lib.bflbm_placement_report(self._h, ms, ctypes.byref(n), ctypes.byref(k)))
        return None if n.value == 0 else {"candidates_ms_per_step": [round(float(v), 4) for v in list(ms)[:n.value]], "kept": k.value}

    def tune_placement(self, max_candidates=3):
        """Draw the physical placement of the state again (leaves the context as freshly created: call before an init)."""
        check(self.lib.bflbm_tune_placement(self._h, int(max_candidates), None, None))
        return self.placement_report()

    @property
    def state_total_max(self):
        """Largest |rho + phi| of the state the last LBM_init(f0, g0) made resident (what `auto` keys its stability bound
        on); negative after an analytic init, where rho_hi + rho_lo of the parameters is that number."""
        v = ctypes.c_double()
        check(self.lib.bflbm_state_total_max(self._h, ctypes.byref(v)))
        return v.value

    def set_state_total_max(self, v):
        check(self.lib.bflbm_set_state_total_max(self._h, float(v)))

    # -- state and per-step fields -------------------------------------------------------------
    def populations(self, f=None, g=None, fab=None):
        """fold, gold valid cells (post-stream state of the last completed step)."""
        if fab is None:
            f = self._new(NVEL) if f is None else f
            g = self._new(NVEL) if g is None else g
            fab = self.slab_fab()
        check(self.lib.bflbm_download_fg(self._h, _ptr(f), _ptr(g), ctypes.byref(fab)))
        return f, g

    def LBM_hydrovars_density(self, out=None, fab=None, ncomp=NHYDROBAR):
        if fab is None:
            out = self._new(ncomp) if out is None else out
            fab = self.slab_fab()
        check(self.lib.bflbm_get_hydrovsbar(self._h, _ptr(out), int(ncomp), ctypes.byref(fab)))
        return out

    def LBM_hydrovars(self, out=None, fab=None, ncomp=NHYDRO):
        if fab is None:
            out = self._new(ncomp) if out is None else out
            fab = self.slab_fab()
        check(self.lib.bflbm_get_hydrovs(self._h, _ptr(out), int(ncomp), ctypes.byref(fab)))
        return out

    def thermal_noise(self, fn=None, gn=None, fab=None):
        if fab is None:
            fn = self._new(NVEL) if fn is None else fn
            gn = self._new(NVEL) if gn is None else gn
            fab = self.slab_fab()
        check(self.lib.bflbm_get_noise(self._h, _ptr(fn), _ptr(gn), ctypes.byref(fab)))
        return fn, gn

    def inject_noise(self, fn, gn, fab=None):
        """Test hook: the next step's collision uses these noise moments."""
        if fn is None:
            check(self.lib.bflbm_inject_noise(self._h, None, None, None))
            return
        if fab is None:
            _check_array(fn, NVEL, self.slab_shape, "fn")
            _check_array(gn, NVEL, self.slab_shape, "gn")
            fab = self.slab_fab()
        check(self.lib.bflbm_inject_noise(self._h, _ptr(fn), _ptr(gn), ctypes.byref(fab)))

    # -- reductions -------------------------------------------------------------------------
    def com_sums(self):
        s = (ctypes.c_double * 4)()
        check(self.lib.bflbm_com_sums(self._h, s))
        return np.array(list(s))

    def update_com(self):
        """Centre of mass of rho (LBM_hydrovs.H:26-60), single slab."""
        s = self.com_sums()
        return s[1:] / s[0]

    def mass(self):
        r, p = ctypes.c_double(), ctypes.c_double()
        check(self.lib.bflbm_mass(self._h, ctypes.byref(r), ctypes.byref(p)))
        return r.value, p.value

    # -- reference-state noise: the reference's USE_REF_STATE build (LBM_binary.H:12, :92-107) ---------
    def set_ref_state(self, rho_eq, phi_eq, rhot_eq, com_ref, fab=None):
        """Equilibrium fields over the GLOBAL lattice (nz, ny, nx) (main_run_job.cpp:216-235) and the
        reference centre of mass com_ref[0]; noise amplitudes then come from them."""
        n = self.n
        fab = fab if fab is not None else make_fab((0, 0, 0), (n[0] - 1, n[1] - 1, n[2] - 1))
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (rho_eq, phi_eq, rhot_eq)]
        check(self.lib.bflbm_set_ref_state(self._h, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), ctypes.byref(fab)))
        c = (ctypes.c_double * 3)(*[float(v) for v in com_ref])
        check(self.lib.bflbm_enable_ref_state(self._h, 1, c))

    def disable_ref_state(self):
        check(self.lib.bflbm_enable_ref_state(self._h, 0, None))

    @property
    def ref_state_active(self):
        a = ctypes.c_int()
        check(self.lib.bflbm_ref_state_active(self._h, ctypes.byref(a)))
        return bool(a.value)

    def set_com(self, com):
        """Global centre of mass of the resident state, for a slab of a decomposed lattice."""
        c = (ctypes.c_double * 3)(*[float(v) for v in com])
        check(self.lib.bflbm_set_com(self._h, c))

    # -- halo exchange plumbing (used by slab.SlabLattice) ---------------------------------------
    def halo_bytes(self, kind=_lib.HALO_STATE):
        n = ctypes.c_size_t()
        check(self.lib.bflbm_halo_bytes(self._h, int(kind), ctypes.byref(n)))
        return n.value

    def halo_pack(self, kind, side, device_ptr):
        check(self.lib.bflbm_halo_pack(self._h, int(kind), int(side), ctypes.c_void_p(device_ptr)))

    def halo_unpack(self, kind, side, device_ptr):
        check(self.lib.bflbm_halo_unpack(self._h, int(kind), int(side), ctypes.c_void_p(device_ptr)))

    def halo_planes(self, kind, side, pack):
        """(device addresses of the 38 component planes of a face, bytes per plane): the staging-free exchange."""
        ptrs = (ctypes.c_void_p * 64)()
        nb, cnt = ctypes.c_size_t(), ctypes.c_int()
        check(self.lib.bflbm_halo_planes(self._h, int(kind), int(side), int(bool(pack)), ptrs, ctypes.byref(nb), ctypes.byref(cnt)))
        return [int(ptrs[k]) for k in range(cnt.value)], nb.value

    def halo_plane_tensors(self, kind, side, pack, device):
        """The same as torch tensors that alias the state buffer (zero-copy, through __cuda_array_interface__)."""
        import torch
        cache = self.__dict__.setdefault("_plane_views", {})
        ptrs, nb = self.halo_planes(kind, side, pack)
        out = []
        for p in ptrs:
            t = cache.get(p)
            if t is None:
                t = cache[p] = torch.as_tensor(_DevicePlane(p, nb // 8), device=device)
            out.append(t)
        return out

    # -- misc ------------------------------------------------------------------------------
    def sync(self):
        check(self.lib.bflbm_sync(self._h))

    def timer_start(self):
        check(self.lib.bflbm_timer_start(self._h))

    def timer_stop(self):
        ms = ctypes.c_float()
        check(self.lib.bflbm_timer_stop(self._h, ctypes.byref(ms)))
        return ms.value

    def debug_time_kernel(self, which, reps=10):
        """ms per launch of a calibration kernel (0 pull-copy, 1 density pass, 2 D2D memcpy)."""
        ms = ctypes.c_float()
        check(self.lib.bflbm_debug_time_kernel(self._h, int(which), int(reps), ctypes.byref(ms)))
        return ms.value

    def device_bytes(self):
        n = ctypes.c_size_t()
        check(self.lib.bflbm_device_bytes(self._h, ctypes.byref(n)))
        return n.value


class RingLBM(_DropletMixin):
    _droplet_fn = ("bflbm_ring_droplet_moments", "bflbm_ring_fit_droplet", "bflbm_ring_fit_droplet_flow")

    """The whole lattice as a ring of z-slabs driven by this one process through the native ring of the
    C-ABI (bflbm_ring_*): slab r on GPU devices[r % len(devices)], halo exchange by peer copies overlapped
    with the interior planes.  Same operator surface as BinaryLBM; fields are full-lattice arrays."""

    def __init__(self, nx, ny=None, nz=None, nslabs=1, devices=(0,), params=None, schedule=None):
        self.lib = _lib.load()
        ny = nx if ny is None else ny
        nz = nx if nz is None else nz
        self.n = (int(nx), int(ny), int(nz))
        self.params = params if params is not None else default_params()
        n3 = (ctypes.c_int * 3)(*self.n)
        dev = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        check(self.lib.bflbm_ring_create(ctypes.byref(self.params), n3, int(nslabs), dev, len(devices), ctypes.byref(h)))
        self._h = h
        self.slabs = []
        for r in range(int(nslabs)):
            c = ctypes.c_void_p()
            check(self.lib.bflbm_ring_slab(self._h, r, ctypes.byref(c)))
            z0, z1 = (self.n[2] * r) // nslabs, (self.n[2] * (r + 1)) // nslabs
            self.slabs.append(BinaryLBM._borrow(c, self.n, z0, z1, r, nslabs, self.params))
        if schedule is not None:
            code = SCHEDULES.get(schedule, schedule)
            check(self.lib.bflbm_ring_set_schedule(self._h, int(code)))

    def close(self):
        if getattr(self, "_h", None):
            for d in list(getattr(self, "_dependents", [])):      # ring-level accumulators (structure factors)
                d.close()
            for s in self.slabs:
                for d in list(getattr(s, "_dependents", [])):
                    d.close()
            self.lib.bflbm_ring_destroy(self._h)
            self._h = None
            for s in self.slabs:
                s._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(f"unknown model parameter {k!r}")
            setattr(self.params, k, v)
        check(self.lib.bflbm_ring_set_params(self._h, ctypes.byref(self.params)))

    def LBM_init_mixture(self):
        check(self.lib.bflbm_ring_init_mixture(self._h))

    def LBM_init_stripe(self, frac):
        check(self.lib.bflbm_ring_init_stripe(self._h, float(frac)))

    def LBM_init_droplet(self, r):
        check(self.lib.bflbm_ring_init_droplet(self._h, float(r)))

    def LBM_init(self, f0, g0):
        """f0, g0: full-lattice arrays (19, nz, ny, nx); every slab takes its planes."""
        _check_array(f0, NVEL, (self.n[2], self.n[1], self.n[0]), "f0")
        _check_array(g0, NVEL, (self.n[2], self.n[1], self.n[0]), "g0")
        fab = make_fab((0, 0, 0), (self.n[0] - 1, self.n[1] - 1, self.n[2] - 1))
        for s in self.slabs:
            s.upload(f0, g0, fab)
        check(self.lib.bflbm_ring_commit_upload(self._h, 1))

    def LBM_timestep(self, nsteps=1):
        check(self.lib.bflbm_ring_step(self._h, int(nsteps)))

    def _gather(self, ncomp, call):
        out = np.empty((ncomp, self.n[2], self.n[1], self.n[0]))
        fab = make_fab((0, 0, 0), (self.n[0] - 1, self.n[1] - 1, self.n[2] - 1))
        for s in self.slabs:
            call(s, out, fab)
        return out

    def populations(self):
        f = np.empty((NVEL, self.n[2], self.n[1], self.n[0])); g = np.empty_like(f)
        fab = make_fab((0, 0, 0), (self.n[0] - 1, self.n[1] - 1, self.n[2] - 1))
        for s in self.slabs:
            s.populations(f, g, fab)
        return f, g

    def LBM_hydrovars_density(self):
        return self._gather(NHYDROBAR, lambda s, o, fab: s.LBM_hydrovars_density(o, fab))

    def LBM_hydrovars(self, ncomp=NHYDRO):
        check(self.lib.bflbm_ring_prepare_ref(self._h))
        return self._gather(ncomp, lambda s, o, fab: s.LBM_hydrovars(o, fab, ncomp))

    def set_ref_state(self, rho_eq, phi_eq, rhot_eq, com_ref):
        n = self.n
        fab = make_fab((0, 0, 0), (n[0] - 1, n[1] - 1, n[2] - 1))
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (rho_eq, phi_eq, rhot_eq)]
        check(self.lib.bflbm_ring_set_ref_state(self._h, _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]), ctypes.byref(fab)))
        c = (ctypes.c_double * 3)(*[float(v) for v in com_ref])
        check(self.lib.bflbm_ring_enable_ref_state(self._h, 1, c))

    def thermal_noise(self):
        check(self.lib.bflbm_ring_prepare_ref(self._h))
        fn = np.empty((NVEL, self.n[2], self.n[1], self.n[0])); gn = np.empty_like(fn)
        fab = make_fab((0, 0, 0), (self.n[0] - 1, self.n[1] - 1, self.n[2] - 1))
        for s in self.slabs:
            s.thermal_noise(fn, gn, fab)
        return fn, gn

    def update_com(self):
        s = (ctypes.c_double * 4)()
        check(self.lib.bflbm_ring_com_sums(self._h, s))
        s = np.array(list(s))
        return s[1:] / s[0]

    def mass(self):
        r, p = ctypes.c_double(), ctypes.c_double()
        check(self.lib.bflbm_ring_mass(self._h, ctypes.byref(r), ctypes.byref(p)))
        return r.value, p.value

    def sync(self):
        check(self.lib.bflbm_ring_sync(self._h))

    @property
    def steps_done(self):
        return self.slabs[0].steps_done

    def set_steps_done(self, n):
        check(self.lib.bflbm_ring_set_step_count(self._h, int(n)))

    TRANSPORTS = {"kernel": 0, "copy": 1}

    def set_transport(self, transport):
        """How the faces move: "kernel" (default) = one gather kernel per face reading the neighbour's planes in place
        (peer memory over xGMI); "copy" = the copy engine, 38 hipMemcpyPeerAsync per face, no compute units."""
        check(self.lib.bflbm_ring_set_transport(self._h, int(self.TRANSPORTS.get(transport, transport))))

    def set_overlap(self, on):
        """False: the faces move after the interior sweep instead of behind it (measures what the overlap buys)."""
        check(self.lib.bflbm_ring_set_overlap(self._h, 1 if on else 0))

    def last_transport(self):
        """(faces moved by the gather kernel, faces moved by the copy engine) in the last exchange."""
        a, b = ctypes.c_int(), ctypes.c_int()
        check(self.lib.bflbm_ring_last_transport(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value


def rng_site_normals(seed, site, noise_index):
    """Host evaluation of the project's Gaussian stream: the 33 normals of one site and noise index."""
    out = (ctypes.c_double * 36)()
    check(_lib.load().bflbm_rng_site_normals(int(seed), int(site), int(noise_index), out))
    return np.array(list(out))[:33]
