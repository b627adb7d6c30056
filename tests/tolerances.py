"""The north-star tolerance ("velocity/density moments within 1e-12 relative", BASELINE.json) as an explicit metric for a
schedule that is NOT bit-exact (schedule 3).  Test infrastructure.

Read element by element a relative error is undefined where the moment itself passes through zero, and both fluids of
this model do: with rho_lo = 0 the minority fluid of a bath is a sum of 19 populations of both signs (rho ~ 1e-9 moving
at |u| ~ 100 lattice units next to a fresh interface), its density changes sign, and u = j / rho is guarded by
`abs(rho) > FLT_EPSILON` (LBM_binary.H:246-247).  The ORACLE itself answers a change of one ulp in its initial
populations with element-wise relative errors of 6e-12 in such densities and 1e-11 in such velocities within 10 steps,
while everything below stays at 1e-15 (tools/ho_stress.py, NOTES.md section 3.1c).  The metric therefore is:

  densities rho, phi, rho+phi   max |d| / max |field|                                       (everywhere)
                                |d| / |value| at sites where |value| >= 1e-3 max |field|    (where there is fluid)
  velocities uf, ug             |d| / max(cs, |u|) at sites where that fluid's |density| >= 1e-3 of its maximum
  momentum  rho uf, phi ug      |density| |d u| / (cs max |density|)                        (everywhere)

`errors()` returns the largest of each; `check()` asserts all four <= tol (1e-12).

The STRICT form of SURVEY 8d / BASELINE.json north_star -- rho, phi, rho+phi relative 1e-12 at EVERY site, velocities
absolute 1e-12 cs at EVERY site -- is reported beside it, unmasked (`dens_elem_all`, `vel_abs_all`; `strict()`): the
tests assert it wherever it holds and fall back to the metric above only for an explicit, committed list of cases
(tests/golden/handover_strict_exceptions.json), where the unmasked figure is in turn held to a multiple of the oracle's
own response to a one-ulp perturbation of its initial state (`one_ulp_response`).
"""
import numpy as np

CS = float(np.sqrt(1.0 / 3.0))
FLUID = 1e-3      # "where there is fluid": the density is at least this fraction of the field's maximum


def errors(h, ref):
    """h, ref: hydrovs arrays with at least comps 0..8 (rho, phi, uf, rho+phi, ug).  -> dict of the four error figures."""
    out = dict(dens_norm=0.0, dens_elem=0.0, vel=0.0, mom=0.0, dens_elem_all=0.0, vel_abs_all=0.0)
    with np.errstate(all="ignore"):
        for c in (0, 1, 5):
            scale = float(np.abs(ref[c]).max())
            d = np.abs(h[c] - ref[c])
            nz = (ref[c] != 0) & (d != 0)
            if nz.any():                                 # unmasked: every site, whatever its density
                out["dens_elem_all"] = max(out["dens_elem_all"], float(np.nanmax(d[nz] / np.abs(ref[c][nz]))))
            if scale > 0:
                out["dens_norm"] = max(out["dens_norm"], float(np.nanmax(d)) / scale)
                m = np.abs(ref[c]) >= FLUID * scale
                if m.any():
                    out["dens_elem"] = max(out["dens_elem"], float(np.nanmax(d[m] / np.abs(ref[c][m]))))
        for c, uc in ((0, (2, 3, 4)), (1, (6, 7, 8))):
            scale = float(np.abs(ref[c]).max())
            m = np.abs(ref[c]) >= FLUID * scale
            for k in uc:
                d = np.abs(h[k] - ref[k])
                out["vel_abs_all"] = max(out["vel_abs_all"], float(np.nanmax(d)) / CS)      # unmasked, absolute, in units of cs
                if m.any():
                    out["vel"] = max(out["vel"], float(np.nanmax(d[m] / np.maximum(CS, np.abs(ref[k][m])))))
                if scale > 0:
                    out["mom"] = max(out["mom"], float(np.nanmax(np.abs(ref[c]) * d)) / (CS * scale))
        if not (np.isfinite(h[:9]).all()):
            out = {k: float("nan") for k in out}
    return out


MASKED = ("dens_norm", "dens_elem", "vel", "mom")


def check(h, ref, what="", tol=1e-12):
    e = errors(h, ref)
    assert all(e[k] <= tol for k in MASKED), f"{what}: {e}"
    return e


def strict(e, tol=1e-12):
    """SURVEY 8d as written: densities relative `tol` and velocities absolute `tol` cs at every site."""
    return e["dens_elem_all"] <= tol and e["vel_abs_all"] <= tol


def one_ulp_response(ob, shape, init, par, checkpoints):
    """The oracle against ITSELF with every initial population changed by -1, 0 or +1 ulp (what schedule 3's re-ordered
    ring sums amount to): the unmasked error figures at each checkpoint.  -> {steps: errors()}"""
    runs = []
    for perturb in (False, True):
        o = ob.OracleLattice(*shape, params=ob.default_params(**par))
        getattr(o, "init_" + init[0])(*init[1:])
        if perturb:
            rng = np.random.default_rng(1)
            o.f *= 1.0 + rng.integers(-1, 2, o.f.shape) * 2.0 ** -52
            o.g *= 1.0 + rng.integers(-1, 2, o.g.shape) * 2.0 ** -52
            o.refresh("zero")
        done, hs = 0, {}
        for steps in checkpoints:
            for _ in range(steps - done):
                o.timestep()
            done = steps
            hs[steps] = o.h.copy()
        runs.append(hs)
    return {steps: errors(runs[1][steps], runs[0][steps]) for steps in checkpoints}
