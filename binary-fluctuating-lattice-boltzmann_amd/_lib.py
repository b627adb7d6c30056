"""ctypes binding of the C-ABI in include/bflbm.h (libbflbm.so, built by csrc/Makefile).

There is deliberately no CPU fallback: if the HIP library is missing or fails to load,
importing the product path raises.  The CPU checker used by the tests is test infrastructure
and is never imported from here.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BFLBM_LIB", os.path.join(_HERE, "csrc", "libbflbm.so"))   # override only for A/B kernel builds

NVEL = 19
NHYDRO = 22
NHYDROBAR = 9
HALO_STATE, HALO_NEXT, HALO_UPLOAD = 0, 1, 2


class Params(ctypes.Structure):
    """bflbm_params: the reference's model globals (LBM_binary.H:17-30, LBM_d3q19.H:6-10)."""
    _fields_ = [
        ("tau_f", ctypes.c_double), ("tau_g", ctypes.c_double),
        ("alpha0", ctypes.c_double), ("alpha1", ctypes.c_double),
        ("kappa", ctypes.c_double), ("kBT", ctypes.c_double), ("cs2", ctypes.c_double),
        ("rho_lo", ctypes.c_double), ("rho_hi", ctypes.c_double),
        ("seed", ctypes.c_uint64),
    ]


class Domain(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_int * 3), ("z0", ctypes.c_int), ("z1", ctypes.c_int),
        ("rank", ctypes.c_int), ("nranks", ctypes.c_int), ("device", ctypes.c_int),
    ]


class Fab(ctypes.Structure):
    _fields_ = [("lo", ctypes.c_int * 3), ("hi", ctypes.c_int * 3),
                ("vlo", ctypes.c_int * 3), ("vhi", ctypes.c_int * 3)]


class FlowFitOpts(ctypes.Structure):
    """bflbm_flowfit_opts (include/bflbm.h): options of the reference's gradient-flow radius fit."""
    _fields_ = [(n, ctypes.c_double) for n in "W0 R0 eta_W eta_R dt undul_ratio".split()] + \
               [(n, ctypes.c_int) for n in "nstep step_window max_retry".split()]


# name -> (restype, argtypes); every symbol declared in include/bflbm.h
_P = ctypes.POINTER
_vp = ctypes.c_void_p
_dp = _P(ctypes.c_double)
SIGNATURES = {
    "bflbm_default_params": (None, [_P(Params)]),
    "bflbm_abi_version": (ctypes.c_int, []),
    "bflbm_last_error": (ctypes.c_char_p, []),
    "bflbm_device_count": (ctypes.c_int, [_P(ctypes.c_int)]),
    "bflbm_create": (ctypes.c_int, [_P(Params), _P(Domain), _P(_vp)]),
    "bflbm_destroy": (ctypes.c_int, [_vp]),
    "bflbm_set_params": (ctypes.c_int, [_vp, _P(Params)]),
    "bflbm_get_params": (ctypes.c_int, [_vp, _P(Params)]),
    "bflbm_set_stream": (ctypes.c_int, [_vp, _vp, ctypes.c_int]),
    "bflbm_set_schedule": (ctypes.c_int, [_vp, ctypes.c_int]),
    "bflbm_resolved_schedule": (ctypes.c_int, [_vp, _P(ctypes.c_int)]),
    "bflbm_init_mixture": (ctypes.c_int, [_vp]),
    "bflbm_init_stripe": (ctypes.c_int, [_vp, ctypes.c_double]),
    "bflbm_init_droplet": (ctypes.c_int, [_vp, ctypes.c_double]),
    "bflbm_upload_fg": (ctypes.c_int, [_vp, _vp, _vp, _P(Fab)]),
    "bflbm_commit_upload": (ctypes.c_int, [_vp, ctypes.c_int]),
    "bflbm_download_fg": (ctypes.c_int, [_vp, _vp, _vp, _P(Fab)]),
    "bflbm_step": (ctypes.c_int, [_vp, ctypes.c_int]),
    "bflbm_step_count": (ctypes.c_int, [_vp, _P(ctypes.c_longlong)]),
    "bflbm_set_step_count": (ctypes.c_int, [_vp, ctypes.c_longlong]),
    "bflbm_ring_set_step_count": (ctypes.c_int, [_vp, ctypes.c_longlong]),
    "bflbm_ring_set_overlap": (ctypes.c_int, [_vp, ctypes.c_int]),
    "bflbm_ring_set_transport": (ctypes.c_int, [_vp, ctypes.c_int]),
    "bflbm_ring_last_transport": (ctypes.c_int, [_vp, _P(ctypes.c_int), _P(ctypes.c_int)]),
    "bflbm_debug_addresses": (ctypes.c_int, [_vp, _P(ctypes.c_ulonglong)]),
    "bflbm_tune_placement": (ctypes.c_int, [_vp, ctypes.c_int, _P(ctypes.c_float), _P(ctypes.c_int)]),
    "bflbm_placement_report": (ctypes.c_int, [_vp, _P(ctypes.c_float), _P(ctypes.c_int), _P(ctypes.c_int)]),
    "bflbm_state_total_max": (ctypes.c_int, [_vp, _dp]),
    "bflbm_set_state_total_max": (ctypes.c_int, [_vp, ctypes.c_double]),
    "bflbm_step_boundary": (ctypes.c_int, [_vp]),
    "bflbm_step_interior": (ctypes.c_int, [_vp]),
    "bflbm_step_finish": (ctypes.c_int, [_vp]),
    "bflbm_halo_bytes": (ctypes.c_int, [_vp, ctypes.c_int, _P(ctypes.c_size_t)]),
    "bflbm_halo_pack": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    "bflbm_halo_unpack": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    "bflbm_halo_planes": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P(_vp), _P(ctypes.c_size_t), _P(ctypes.c_int)]),
    "bflbm_ring_create": (ctypes.c_int, [_P(Params), _P(ctypes.c_int), ctypes.c_int, _P(ctypes.c_int), ctypes.c_int, _P(_vp)]),
    "bflbm_ring_destroy": (ctypes.c_int, [_vp]),
    "bflbm_ring_size": (ctypes.c_int, [_vp, _P(ctypes.c_int)]),
    "bflbm_ring_slab": (ctypes.c_int, [_vp, ctypes.c_int, _P(_vp)]),
    "bflbm_ring_set_params": (ctypes.c_int, [_vp, _P(Params)]),
    "bflbm_ring_set_schedule": (ctypes.c_int, [_vp, ctypes.c_int]),
    "bflbm_ring_init_mixture": (ctypes.c_int, [_vp]),
    "bflbm_ring_init_stripe": (ctypes.c_int, [_vp, ctypes.c_double]),
    "bflbm_ring_init_droplet": (ctypes.c_int, [_vp, ctypes.c_double]),
    "bflbm_ring_commit_upload": (ctypes.c_int, [_vp, ctypes.c_int]),
    "bflbm_ring_step": (ctypes.c_int, [_vp, ctypes.c_int]),
    "bflbm_ring_com_sums": (ctypes.c_int, [_vp, _dp]),
    "bflbm_ring_mass": (ctypes.c_int, [_vp, _dp, _dp]),
    "bflbm_ring_sync": (ctypes.c_int, [_vp]),
    "bflbm_ring_set_ref_state": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Fab)]),
    "bflbm_ring_enable_ref_state": (ctypes.c_int, [_vp, ctypes.c_int, _dp]),
    "bflbm_ring_prepare_ref": (ctypes.c_int, [_vp]),
    "bflbm_set_ref_state": (ctypes.c_int, [_vp, _vp, _vp, _vp, _P(Fab)]),
    "bflbm_enable_ref_state": (ctypes.c_int, [_vp, ctypes.c_int, _dp]),
    "bflbm_ref_state_active": (ctypes.c_int, [_vp, _P(ctypes.c_int)]),
    "bflbm_set_com": (ctypes.c_int, [_vp, _dp]),
    "bflbm_get_hydrovsbar": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _P(Fab)]),
    "bflbm_get_hydrovs": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _P(Fab)]),
    "bflbm_get_noise": (ctypes.c_int, [_vp, _vp, _vp, _P(Fab)]),
    "bflbm_inject_noise": (ctypes.c_int, [_vp, _vp, _vp, _P(Fab)]),
    "bflbm_com_sums": (ctypes.c_int, [_vp, _dp]),
    "bflbm_mass": (ctypes.c_int, [_vp, _dp, _dp]),
    "bflbm_sync": (ctypes.c_int, [_vp]),
    "bflbm_droplet_moments": (ctypes.c_int, [_vp, _dp]),
    "bflbm_ring_droplet_moments": (ctypes.c_int, [_vp, _dp]),
    "bflbm_fit_droplet": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, ctypes.c_double, _dp, _P(ctypes.c_int)]),
    "bflbm_ring_fit_droplet": (ctypes.c_int, [_vp, _dp, _dp, ctypes.c_int, ctypes.c_double, _dp, _P(ctypes.c_int)]),
    "bflbm_flowfit_default_opts": (None, [_P(FlowFitOpts)]),
    "bflbm_fit_droplet_flow": (ctypes.c_int, [_vp, _P(FlowFitOpts), _dp, _P(ctypes.c_int)]),
    "bflbm_ring_fit_droplet_flow": (ctypes.c_int, [_vp, _P(FlowFitOpts), _dp, _P(ctypes.c_int)]),
    "bflbm_flowfit_coefficients": (ctypes.c_int, [ctypes.c_double] * 6 + [_dp]),
    "bflbm_sf_create": (ctypes.c_int, [_vp, ctypes.c_int, _P(ctypes.c_int), _P(ctypes.c_int), _dp, _P(_vp)]),
    "bflbm_sf_destroy": (ctypes.c_int, [_vp]),
    "bflbm_sf_reset": (ctypes.c_int, [_vp]),
    "bflbm_sf_accumulate": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "bflbm_sf_nsamples": (ctypes.c_int, [_vp, _P(ctypes.c_longlong)]),
    "bflbm_sf_get": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    "bflbm_ring_sf_create": (ctypes.c_int, [_vp, ctypes.c_int, _P(ctypes.c_int), _P(ctypes.c_int), _dp, _P(_vp)]),
    "bflbm_ring_sf_destroy": (ctypes.c_int, [_vp]),
    "bflbm_ring_sf_reset": (ctypes.c_int, [_vp]),
    "bflbm_ring_sf_accumulate": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "bflbm_ring_sf_nsamples": (ctypes.c_int, [_vp, _P(ctypes.c_longlong)]),
    "bflbm_ring_sf_get": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _vp]),
    "bflbm_timer_start": (ctypes.c_int, [_vp]),
    "bflbm_timer_stop": (ctypes.c_int, [_vp, _P(ctypes.c_float)]),
    "bflbm_rng_site_normals": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, _dp]),
    "bflbm_debug_time_kernel": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, _P(ctypes.c_float)]),
    "bflbm_device_bytes": (ctypes.c_int, [_vp, _P(ctypes.c_size_t)]),
}


class BflbmError(RuntimeError):
    pass


_lib = None


def load():
    """Load libbflbm.so; raises (never falls back) when the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BflbmError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C "
            f"{os.path.dirname(LIB_PATH)}). There is no CPU fallback.")
    # torch bundles its own ROCm runtime (libamdhip64.so.7, same soname as /opt/rocm's).  Whichever
    # copy is loaded first serves the process; if ours came first, torch would load a second copy
    # that cannot see the GPU ("No HIP GPUs are available").  Importing torch first makes both
    # sides share torch's runtime (measured on MI355X: tools/order_probe.py).
    if os.environ.get("BFLBM_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.bflbm_abi_version() != 1:
        raise BflbmError("libbflbm.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().bflbm_last_error()
        raise BflbmError(msg.decode() if msg else f"bflbm call failed ({rc})")
