"""z-slab decomposition of the lattice over the GPUs of one node.

The reference distributes boxes with AMReX (BoxArray::maxSize + DistributionMapping,
main_run_job.cpp:140-143) and fills 113 components x 2 ghost layers on all six faces every
step (LBM_binary.H:130-131, :312, :353, :553-555).  Here each GPU owns a contiguous z-slab;
x and y wrap inside the slab, and one exchange per step moves exactly the populations that
cross a z face: 38 component-planes per face (see csrc/bflbm.hip halo_table).  The exchange
of step t's results is posted as soon as the slab's two outermost plane pairs are done and
overlaps the interior planes of the same step:

    step_boundary -> 38 isend + 38 irecv per face, one batch (RCCL stream) || step_interior -> wait -> finish

Every (component, plane) entry of a face is one contiguous plane of the state buffer, so the sends are posted straight
from the planes the boundary kernels wrote and the receives straight into the halo planes the next step pulls from
(`halo_plane_tensors`): no pack/unpack kernels, no buffers.  With two ranks both faces go to the same peer in that one
batch.  An engine without `halo_plane_tensors`, BFLBM_SLAB_TRANSPORT=staged, or -- until the direct path has been seen to
work on several GPUs -- the RCCL backend by default, takes the staged path (pack -> one message per face -> unpack).

torch.distributed is plumbing only (process group, P2P); all lattice work is in the HIP library.
`engine_factory` lets the CPU test-suite drive the same protocol with a stand-in engine.
"""
import os

import numpy as np

from . import _lib
from .lattice import BinaryLBM


def slab_bounds(nz, nranks, rank):
    """Balanced contiguous split of nz planes."""
    return (nz * rank) // nranks, (nz * (rank + 1)) // nranks


class _Protocol:
    """The per-step and upload protocols, independent of how buffers travel."""

    overlap = True      # False: exchange after the whole sweep (measurement of what the overlap buys)

    def __init__(self, engine):
        self.engine = engine

    # subclasses implement _exchange(kind, pack=True)
    def _prepare_ref(self):
        """Reference-state noise (USE_REF_STATE): the slab needs the GLOBAL centre of mass of the
        resident state (LBM_binary.H:585-588) before its noise is drawn."""
        e = self.engine
        if getattr(e, "ref_state_active", False):
            e.set_com(self.update_com())

    def LBM_timestep(self, nsteps=1):
        e = self.engine
        for _ in range(int(nsteps)):
            self._prepare_ref()
            e.step_boundary()
            if self.overlap:
                self._post(_lib.HALO_NEXT)
                e.step_interior()
            else:
                e.step_interior()
                e.sync()                       # nothing of the sweep is left to hide the exchange behind
                self._post(_lib.HALO_NEXT)
            self._complete(_lib.HALO_NEXT)
            e.step_finish()

    def exchange(self, kind):
        self._post(kind)
        self._complete(kind)


class SlabLattice(_Protocol):
    """One rank's slab plus the ring exchange with its +-z neighbours via torch.distributed."""

    def __init__(self, nx, ny, nz, params=None, group=None, device=None, schedule=None,
                 engine_factory=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.n = (nx, ny, nz)
        z0, z1 = slab_bounds(nz, self.world, self.rank)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        if engine_factory is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            engine = BinaryLBM(nx, ny, nz, params=params, z0=z0, z1=z1, rank=self.rank, nranks=self.world,
                               device=self.device.index or 0, schedule=schedule, stream=stream)
        else:
            engine = engine_factory(nx, ny, nz, z0, z1, self.rank, self.world)
        super().__init__(engine)
        self.z0, self.z1 = z0, z1
        self.lower = (self.rank - 1) % self.world
        self.upper = (self.rank + 1) % self.world
        self._bufs = None
        self._work = None
        # what crosses one face per step: 38 component planes (14 + 14 of the nearest plane, 5 + 5 of the second)
        self.halo_bytes_per_face = int(engine.halo_bytes(_lib.HALO_STATE)) if self.world > 1 else 0
        # RCCL orders its work after the current stream by itself; host-driven backends (gloo) read
        # the send buffers from the host side, so the pack kernels must have completed first.
        self._host_sync = dist.get_backend(group) != "nccl"
        # Which exchange: "direct" (38 plane-sized P2P operations per face between the state buffers, no staging kernels)
        # or "staged" (pack -> one message per face -> unpack).  Over RCCL the default is STAGED until the direct path has
        # one verified multi-GPU run (152 P2P operations in one group, receives landing in live state planes: ADVICE r3);
        # the staging kernels move 6 x 38 planes of the 38 x nz the step moves anyway.  Host-driven backends (gloo) take the
        # direct path.  BFLBM_SLAB_TRANSPORT=direct|staged (or the older BFLBM_SLAB_STAGED=1) decides explicitly.
        want = os.environ.get("BFLBM_SLAB_TRANSPORT", "").strip().lower()
        if os.environ.get("BFLBM_SLAB_STAGED", "0") == "1":
            want = "staged"
        if want not in ("", "direct", "staged"):
            raise ValueError(f"BFLBM_SLAB_TRANSPORT={want!r}: expected direct or staged")
        if want == "":
            want = "staged" if dist.get_backend(group) == "nccl" else "direct"
        self.direct = hasattr(engine, "halo_plane_tensors") and want == "direct"
        if self.world > 1 and not self.direct:
            n = engine.halo_bytes(_lib.HALO_STATE) // 8
            mk = lambda: torch.empty(n, dtype=torch.float64, device=self.device)
            # send_lo goes to the lower neighbour (its high halo); recv_hi comes from the upper one
            self._bufs = dict(send_lo=mk(), send_hi=mk(), recv_lo=mk(), recv_hi=mk())

    def _ranks(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def _post(self, kind):
        if self.world == 1:
            return
        if self.direct:
            return self._post_direct(kind)
        e, b, dist = self.engine, self._bufs, self.dist
        e.halo_pack(kind, 0, b["send_lo"].data_ptr())
        e.halo_pack(kind, 1, b["send_hi"].data_ptr())
        if self._host_sync:
            e.sync()
        lo, up = self._ranks(self.lower), self._ranks(self.upper)
        # posting order matters when lower == upper (world == 2): the peer's first recv (recv_hi,
        # from ITS upper = me) must meet my first send (send_lo).
        ops = [dist.P2POp(dist.isend, b["send_lo"], lo, self.group, tag=0),
               dist.P2POp(dist.isend, b["send_hi"], up, self.group, tag=1),
               dist.P2POp(dist.irecv, b["recv_hi"], up, self.group, tag=0),
               dist.P2POp(dist.irecv, b["recv_lo"], lo, self.group, tag=1)]
        self._work = dist.batch_isend_irecv(ops)

    def _post_direct(self, kind):
        """One batch of plane-sized P2P operations between the state buffers themselves.  Entry k of my low face pairs
        with entry k of the lower neighbour's high halo (same table order on both sides); the tag carries face and
        entry so that host-driven backends match them too.  Posting order as in the staged path: with two ranks the
        peer's first receives (its high halo, from ITS upper neighbour = me) meet my first sends (my low face)."""
        e, dist = self.engine, self.dist
        dev = self.device
        s_lo, s_hi = e.halo_plane_tensors(kind, 0, True, dev), e.halo_plane_tensors(kind, 1, True, dev)
        r_lo, r_hi = e.halo_plane_tensors(kind, 0, False, dev), e.halo_plane_tensors(kind, 1, False, dev)
        if self._host_sync:
            e.sync()
        lo, up = self._ranks(self.lower), self._ranks(self.upper)
        ops = [dist.P2POp(dist.isend, t, lo, self.group, tag=k) for k, t in enumerate(s_lo)]
        ops += [dist.P2POp(dist.isend, t, up, self.group, tag=64 + k) for k, t in enumerate(s_hi)]
        ops += [dist.P2POp(dist.irecv, t, up, self.group, tag=k) for k, t in enumerate(r_hi)]
        ops += [dist.P2POp(dist.irecv, t, lo, self.group, tag=64 + k) for k, t in enumerate(r_lo)]
        self._work = dist.batch_isend_irecv(ops)

    def _complete(self, kind):
        if self.world == 1:
            return
        for w in self._work:
            w.wait()
        self._work = None
        if self.direct:
            return
        e, b = self.engine, self._bufs
        e.halo_unpack(kind, 0, b["recv_lo"].data_ptr())
        e.halo_unpack(kind, 1, b["recv_hi"].data_ptr())

    # -- reference operator surface -------------------------------------------------------------
    def LBM_init_mixture(self):
        self.engine.LBM_init_mixture()          # analytic: halo planes are filled by the kernel

    def LBM_init_stripe(self, frac):
        self.engine.LBM_init_stripe(frac)

    def LBM_init_droplet(self, r):
        self.engine.LBM_init_droplet(r)

    def LBM_init(self, f0, g0, fab=None):
        """LBM_binary.H:632-661 with mf.ParallelCopy + FillBoundary replaced by upload + z exchange."""
        self.engine.upload(f0, g0, fab)
        if self.world > 1:
            self.exchange(_lib.HALO_UPLOAD)
        self.engine.commit_upload(True)
        if self.world > 1:
            # every rank resolves `auto` on the whole lattice's total density, not on its own slab's (csrc/bflbm.hip
            # handover_contract_params): one all-reduce of a scalar
            if hasattr(self.engine, "set_state_total_max"):
                m = self.torch.tensor([self.engine.state_total_max], dtype=self.torch.float64, device=self.device)
                self.dist.all_reduce(m, op=self.dist.ReduceOp.MAX, group=self.group)
                self.engine.set_state_total_max(float(m.item()))
            self.exchange(_lib.HALO_STATE)

    def populations(self, *a, **k):
        return self.engine.populations(*a, **k)

    def LBM_hydrovars_density(self, *a, **k):
        return self.engine.LBM_hydrovars_density(*a, **k)

    def LBM_hydrovars(self, *a, **k):
        self._prepare_ref()
        return self.engine.LBM_hydrovars(*a, **k)

    def thermal_noise(self, *a, **k):
        self._prepare_ref()
        return self.engine.thermal_noise(*a, **k)

    def set_ref_state(self, rho_eq, phi_eq, rhot_eq, com_ref):
        """Global equilibrium fields (nz, ny, nx) on every rank."""
        self.engine.set_ref_state(rho_eq, phi_eq, rhot_eq, com_ref)

    def update_com(self):
        """update_com (LBM_hydrovs.H:26-60) with the four .sum() reductions as one all-reduce."""
        s = self.torch.as_tensor(self.engine.com_sums(), dtype=self.torch.float64, device=self.device)
        if self.world > 1:
            self.dist.all_reduce(s, group=self.group)
        s = s.cpu().numpy()
        return s[1:] / s[0]

    def mass(self):
        r, p = self.engine.mass()
        s = self.torch.tensor([r, p], dtype=self.torch.float64, device=self.device)
        if self.world > 1:
            self.dist.all_reduce(s, group=self.group)
        return tuple(s.cpu().tolist())

    def sync(self):
        self.engine.sync()

    @property
    def steps_done(self):
        return self.engine.steps_done

    def set_steps_done(self, n):
        """Absolute step of the resident state = noise index of the next step: a restart from a kBT > 0 checkpoint
        continues the stream instead of replaying the first segment's normals (main_run_job.cpp:80 step_continue)."""
        self.engine.set_steps_done(n)

    def close(self):
        self.engine.close()


class LocalSlabRing:
    """Several slabs of one lattice driven by ONE process -- the way the reference itself runs
    (single process, USE_MPI=FALSE).  `devices` places slab r on GPU devices[r % len(devices)]: one
    GPU (validation on a 1-GPU box, lattices larger than one allocation) or all GPUs of the node,
    with the halo buffers moved by device-to-device (peer, xGMI) copies instead of RCCL."""

    def __init__(self, nx, ny, nz, nslabs, params=None, device=0, schedule=None, engine_factory=None, devices=None):
        self.n = (nx, ny, nz)
        self.nslabs = int(nslabs)
        self.devices = list(devices) if devices is not None else [device]
        self.engines = []
        for r in range(self.nslabs):
            z0, z1 = slab_bounds(nz, self.nslabs, r)
            if engine_factory is None:
                e = BinaryLBM(nx, ny, nz, params=params, z0=z0, z1=z1, rank=r, nranks=self.nslabs,
                              device=self.devices[r % len(self.devices)], schedule=schedule)
            else:
                e = engine_factory(nx, ny, nz, z0, z1, r, self.nslabs)
            self.engines.append(e)
        self._bufs = None

    def _buffers(self):
        """Per slab: send_lo, send_hi, recv_lo, recv_hi on that slab's device."""
        if self._bufs is None:
            import torch
            e0 = self.engines[0]
            n = e0.halo_bytes(_lib.HALO_STATE) // 8
            self._bufs = []
            for r, e in enumerate(self.engines):
                dev = ("cuda:%d" % self.devices[r % len(self.devices)]) if isinstance(e, BinaryLBM) else "cpu"
                self._bufs.append([torch.empty(n, dtype=torch.float64, device=dev) for _ in range(4)])
        return self._bufs

    def _sync_all(self):
        for e in self.engines:
            e.sync()

    def exchange(self, kind):
        if self.nslabs == 1:
            return
        import torch
        bufs = self._buffers()
        for r, e in enumerate(self.engines):
            e.halo_pack(kind, 0, bufs[r][0].data_ptr())
            e.halo_pack(kind, 1, bufs[r][1].data_ptr())
        self._sync_all()                       # contexts use independent streams
        for r in range(self.nslabs):
            lower, upper = (r - 1) % self.nslabs, (r + 1) % self.nslabs
            bufs[r][2].copy_(bufs[lower][1])   # recv_lo <- lower neighbour's high face (peer copy across GPUs)
            bufs[r][3].copy_(bufs[upper][0])   # recv_hi <- upper neighbour's low face
        if bufs[0][0].is_cuda:
            for d in set(self.devices):
                torch.cuda.synchronize(d)
        for r, e in enumerate(self.engines):
            e.halo_unpack(kind, 0, bufs[r][2].data_ptr())
            e.halo_unpack(kind, 1, bufs[r][3].data_ptr())
        self._sync_all()

    def _prepare_ref(self):
        if getattr(self.engines[0], "ref_state_active", False):
            com = self.update_com()
            for e in self.engines:
                e.set_com(com)

    def set_ref_state(self, rho_eq, phi_eq, rhot_eq, com_ref):
        for e in self.engines:
            e.set_ref_state(rho_eq, phi_eq, rhot_eq, com_ref)

    def LBM_timestep(self, nsteps=1):
        for _ in range(int(nsteps)):
            self._prepare_ref()
            for e in self.engines:
                e.step_boundary()
            for e in self.engines:
                e.step_interior()
            self.exchange(_lib.HALO_NEXT)
            for e in self.engines:
                e.step_finish()

    def LBM_init_mixture(self):
        for e in self.engines:
            e.LBM_init_mixture()

    def LBM_init_stripe(self, frac):
        for e in self.engines:
            e.LBM_init_stripe(frac)

    def LBM_init_droplet(self, r):
        for e in self.engines:
            e.LBM_init_droplet(r)

    def LBM_init(self, f0, g0):
        """f0,g0: full-lattice arrays (19, nz, ny, nx)."""
        for e in self.engines:
            e.upload(np.ascontiguousarray(f0[:, e.z0:e.z1]), np.ascontiguousarray(g0[:, e.z0:e.z1]))
        self.exchange(_lib.HALO_UPLOAD)
        for e in self.engines:
            e.commit_upload(True)
        self.exchange(_lib.HALO_STATE)

    def _gather(self, getter, ncomp):
        nx, ny, nz = self.n
        out = np.empty((ncomp, nz, ny, nx))
        for e in self.engines:
            out[:, e.z0:e.z1] = getter(e)
        return out

    def populations(self):
        nx, ny, nz = self.n
        f = np.empty((19, nz, ny, nx)); g = np.empty_like(f)
        for e in self.engines:
            a, b = e.populations()
            f[:, e.z0:e.z1] = a; g[:, e.z0:e.z1] = b
        return f, g

    def LBM_hydrovars_density(self):
        return self._gather(lambda e: e.LBM_hydrovars_density(), 9)

    def LBM_hydrovars(self):
        self._prepare_ref()
        return self._gather(lambda e: e.LBM_hydrovars(), 22)

    def thermal_noise(self):
        self._prepare_ref()
        nx, ny, nz = self.n
        f = np.empty((19, nz, ny, nx)); g = np.empty_like(f)
        for e in self.engines:
            a, b = e.thermal_noise()
            f[:, e.z0:e.z1] = a; g[:, e.z0:e.z1] = b
        return f, g

    def update_com(self):
        s = sum(e.com_sums() for e in self.engines)
        return s[1:] / s[0]

    def close(self):
        for e in self.engines:
            e.close()
