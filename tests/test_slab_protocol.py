"""CPU tests of the multi-GPU path: slab.SlabLattice's per-step exchange protocol over
torch.distributed (gloo, world_size 2 and 3) and slab.LocalSlabRing, driven by the stand-in
engine, must reproduce the single-box oracle bit for bit."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, outdir, case, staged=False):
    if staged:
        os.environ["BFLBM_SLAB_STAGED"] = "1"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import oracle_binding as ob
    from slab_standin import StandinEngine
    pkg = ge.load_package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n, init, steps, par, upload = case
    params = ob.default_params(**par)
    fac = lambda nx, ny, nz, z0, z1, r, w: StandinEngine(nx, ny, nz, z0, z1, r, w, params)
    lat = pkg.SlabLattice(*n, engine_factory=fac, device="cpu")
    assert lat.direct == (not staged)          # default: sends and receives between the state arrays themselves
    if upload:
        full = ob.OracleLattice(*n, params=params)
        full.init_droplet(0.3)
        rng = np.random.default_rng(5)
        f0 = full.f * (1 + 0.01 * rng.standard_normal(full.f.shape))
        g0 = full.g * (1 + 0.01 * rng.standard_normal(full.g.shape))
        lat.LBM_init(np.ascontiguousarray(f0[:, lat.z0:lat.z1]), np.ascontiguousarray(g0[:, lat.z0:lat.z1]))
    else:
        getattr(lat, "LBM_init_" + init[0])(*init[1:])
    if steps > 2:                      # the last steps with the exchange after the sweep (bench's informational
        lat.LBM_timestep(steps - 2)    # leg): same results, only the order of posting differs
        lat.overlap = False
        lat.LBM_timestep(2)
        lat.overlap = True
    else:
        lat.LBM_timestep(steps)
    f, g = lat.populations()
    h = lat.LBM_hydrovars()
    fn, gn = lat.thermal_noise()
    com = lat.update_com()
    mass = lat.mass()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), f=f, g=g, h=h, fn=fn, z0=lat.z0, z1=lat.z1, com=com, mass=np.array(mass))
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    ((6, 5, 8), ("stripe", 0.5), 4, {}, False),
    ((8, 6, 12), ("droplet", 0.3), 3, dict(kBT=1e-5, alpha0=1.0, seed=99), False),
    ((6, 6, 9), None, 2, {}, True),
]


@pytest.mark.parametrize("staged", [False, True], ids=["direct", "staged"])
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", CASES)
def test_slab_lattice_over_gloo_matches_single_box(ob, world, case, staged):
    """Both transports of slab.SlabLattice: 38 plane-sized sends per face straight between the state arrays (default),
    and pack -> one message per face -> unpack."""
    import torch.multiprocessing as mp
    n, init, steps, par, upload = case
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d, case, staged), nprocs=world, join=True)
        params = ob.default_params(**par)
        ref = ob.OracleLattice(*n, params=params)
        if upload:
            ref.init_droplet(0.3)
            rng = np.random.default_rng(5)
            f0 = ref.f * (1 + 0.01 * rng.standard_normal(ref.f.shape))
            g0 = ref.g * (1 + 0.01 * rng.standard_normal(ref.g.shape))
            ref.init_from(f0, g0)
        else:
            getattr(ref, "init_" + init[0])(*init[1:])
        for _ in range(steps):
            ref.timestep()
        covered = 0
        for r in range(world):
            o = np.load(os.path.join(d, f"r{r}.npz"))
            z0, z1 = int(o["z0"]), int(o["z1"])
            covered += z1 - z0
            assert np.array_equal(o["f"], ref.f[:, z0:z1]), f"rank {r} f"
            assert np.array_equal(o["g"], ref.g[:, z0:z1]), f"rank {r} g"
            assert np.array_equal(o["h"], ref.h[:, z0:z1]), f"rank {r} hydrovs"
            assert np.array_equal(o["fn"], ref.fn[:, z0:z1]), f"rank {r} noise"
            np.testing.assert_allclose(o["com"], ref.com(), rtol=1e-12)
            np.testing.assert_allclose(o["mass"], [ref.hbar[0].sum(), ref.hbar[1].sum()], rtol=1e-12)
        assert covered == n[2]


@pytest.mark.parametrize("nslabs", [1, 2, 4])
def test_local_slab_ring_with_standin(pkg, ob, nslabs):
    from slab_standin import StandinEngine
    n = (6, 5, 16)
    params = ob.default_params(kBT=2e-5, alpha0=2.0)
    fac = lambda nx, ny, nz, z0, z1, r, w: StandinEngine(nx, ny, nz, z0, z1, r, w, params)
    ring = pkg.LocalSlabRing(*n, nslabs, engine_factory=fac)
    ring.LBM_init_droplet(0.3)
    ring.LBM_timestep(3)
    ref = ob.OracleLattice(*n, params=params)
    ref.init_droplet(0.3)
    for _ in range(3):
        ref.timestep()
    f, g = ring.populations()
    assert np.array_equal(f, ref.f) and np.array_equal(g, ref.g)
    assert np.array_equal(ring.LBM_hydrovars(), ref.h)
    np.testing.assert_allclose(ring.update_com(), ref.com(), rtol=1e-12)
