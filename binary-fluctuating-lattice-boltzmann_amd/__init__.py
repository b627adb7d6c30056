"""MI355X-native D3Q19 binary fluctuating-LBM collide-and-stream path.

Import name: the directory name contains hyphens, so load it through
``__graft_entry__.load_package()`` (registers it as ``bflbm_amd``) or importlib.
The product path needs the HIP library csrc/libbflbm.so; there is no CPU fallback.
"""
from . import _lib
from ._lib import BflbmError, Params, Domain, Fab, NVEL, NHYDRO, NHYDROBAR, HALO_STATE, HALO_NEXT, HALO_UPLOAD
from .lattice import BinaryLBM, RingLBM, default_params, make_fab, rng_site_normals
from .slab import SlabLattice, LocalSlabRing, slab_bounds
from . import plotfile
from . import analysis
from . import structfact
from . import run_job

__all__ = ["RingLBM", "SlabLattice", "LocalSlabRing", "slab_bounds", "BinaryLBM", "default_params", "make_fab", "rng_site_normals", "BflbmError",
           "Params", "Domain", "Fab", "NVEL", "NHYDRO", "NHYDROBAR"]
