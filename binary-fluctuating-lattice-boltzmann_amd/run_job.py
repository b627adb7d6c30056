"""The reference's job flow (main_run_job.cpp) on the MI355X path, with the run constants as arguments.

    python -m ... run_job  (or  python binary-fluctuating-lattice-boltzmann_amd/run_job.py)
        --system droplet --nx 64 --nsteps 40000 --plot-int 200 --alpha0 4 --kbt 0 --root ./out

What it mirrors (main_run_job.cpp line numbers):
  * systems and domains :108-134 (mixture nx^3, droplet nx^3, flat interface nx x ny x nz), init dispatch :274-286
  * directory / file names :150-202, :400-405 -- data_mixture_lb_hydrovars | data_interface_alpha0_X |
    data_droplet_density_X_alpha0_X_rX_sizeN-N-N ; .../lbm_data_shshan_alpha0_X_xi_X_sizeN-N-N[_continue]/pltNNNNNNN
  * the loop :335-387: LBM_timestep every step, a frame every plot_int steps from out_step on, noise frames
    every out_noise_step (Debug.H:380-409), the last frame always
  * last-frame checkpoints of fold/gold as one-name plotfiles :400-409, restart from them through LBM_init
    :253-270 (if_continue_from_last_frame)
  * equilibrium-state extraction at kBT == 0 :423-438: PrintConvergence (Debug.H:275-358) = ensemble mean of
    components 0, 1, 5 over the last t_window/plot_int + 1 frames -> equilibrium_{rho,phi,rhot}_alpha0_X_sizeN-N-N,
    and its printed convergence measure (mean absolute deviation from the ensemble mean)
Frames hold `hydrovs` (22 components, VariableNames) -- the reference's STRUCT_HYDROVARS variant; with
--lb-hydrovars the 15-component `hydrovsbar` is written under the same names like the shipped
STRUCT_LB_HYDROVARS build does (main_run_job.cpp:19, :321).
  * --use-ref-state: the noiseSwitch branch :216-235 for a build with USE_REF_STATE (LBM_binary.H:12) -- load the
    three equilibrium files written by a previous kBT = 0 run, com_ref = update_com(rho_eq), and let
    thermal_noise take its amplitudes from them
  * structure factors :99-103, :301-310, :342-349: accumulated every out_SF_step steps inside the last
    plot_SF_window steps of a noisy run and written with the last frame (structfact.py; shipped window = 0)
"""
import argparse
import os
import sys
import time

import numpy as np

NDIGITS = 7                                    # main_run_job.cpp:29


def _fmt_size(n):
    return "size%d-%d-%d" % n


def job_paths(a, n):
    """Directory and file roots exactly as main_run_job.cpp builds them."""
    if a.system == "interface":
        plot_dir = "data_interface_alpha0_%.2f" % a.alpha0
    elif a.system == "droplet":
        plot_dir = "data_droplet_density_%.2f_alpha0_%.2f_r%.3f_%s" % (a.rho_hi, a.alpha0, a.radius, _fmt_size(n))
    else:
        plot_dir = "data_mixture" + ("_lb_hydrovars" if a.lb_hydrovars else "_hydrovars")
    base = os.path.join(a.root, plot_dir)
    tag = "alpha0_%.2f_xi_%.1e_%s" % (a.alpha0, a.kbt, _fmt_size(n))
    plot_root = os.path.join(base, "lbm_data_shshan_" + tag + ("_continue" if a.kbt != 0 else ""), "plt")
    return dict(base=base, plot_root=plot_root,
                chk_f=os.path.join(base, "f_checkpoint"), chk_g=os.path.join(base, "g_checkpoint"),
                eq={k: os.path.join(base, "equilibrium_%s_alpha0_%.2f_%s" % (k, a.alpha0, _fmt_size(n))) for k in ("rho", "phi", "rhot")})


def checkpoint_name(root, step, alpha0, kbt, n):
    return "%s%0*d_alpha0_%.2f_xi_%.1e_%s" % (root, NDIGITS, step, alpha0, kbt, _fmt_size(n))


def print_convergence(pf, plot_root, step1, step2, plot_int, comp):
    """PrintConvergence (Debug.H:275-358): ensemble mean of one component over the saved frames and the
    lattice-averaged mean absolute deviation from it (the p = 1 measure the reference prints)."""
    frames = [pf.read_plotfile(pf.concatenate(plot_root, s, NDIGITS))[0][comp] for s in range(step1, step2 + 1, plot_int)]
    mean = np.zeros_like(frames[0])
    for fr in frames:
        mean = mean + fr
    mean = mean * (1.0 / len(frames))
    dev = np.zeros_like(mean)
    for fr in frames:
        dev = dev + np.abs(fr - mean)
    dev = dev * (1.0 / len(frames))
    return mean, dev.sum() / mean.size


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--system", choices=["mixture", "droplet", "interface"], default="mixture")   # SYS_* :24-26
    ap.add_argument("--nx", type=int, default=32)
    ap.add_argument("--ny", type=int, default=0)
    ap.add_argument("--nz", type=int, default=0)
    ap.add_argument("--nsteps", type=int, default=40000)                   # :87
    ap.add_argument("--plot-int", type=int, default=200)                   # :90
    ap.add_argument("--print-int", type=int, default=20)                   # :91
    ap.add_argument("--out-noise-step", type=int, default=0, help="0 = never (reference: nsteps+1)")   # :96
    ap.add_argument("--plot-sf-window", type=int, default=0)               # :99
    ap.add_argument("--out-sf-step", type=int, default=100)                # :100
    ap.add_argument("--step-continue", type=int, default=0)                # :80
    ap.add_argument("--continue-from-nonfluct", action=argparse.BooleanOptionalAction, default=True)   # :84 (--no-continue-from-nonfluct: restart from a kBT > 0 checkpoint, :259)
    ap.add_argument("--restart", action="store_true", help="if_continue_from_last_frame (:248)")
    ap.add_argument("--radius", type=float, default=0.2)                   # :110
    ap.add_argument("--init-frac", type=float, default=0.5)                # :33
    ap.add_argument("--alpha0", type=float, default=4.0)
    ap.add_argument("--kbt", type=float, default=0.0)
    ap.add_argument("--tau", type=float, default=0.5)
    ap.add_argument("--kappa", type=float, default=4.0)
    ap.add_argument("--rho-hi", type=float, default=1.0)
    ap.add_argument("--rho-lo", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--max-grid-size", type=int, default=0, help="boxes per frame file; default nx/2 (:73)")
    ap.add_argument("--use-ref-state", action="store_true", help="noise amplitudes from the equilibrium_* files (USE_REF_STATE)")
    ap.add_argument("--sf-device", action="store_true", help="accumulate the structure factors on the GPU (hipFFT)")
    ap.add_argument("--lb-hydrovars", action="store_true", help="write hydrovsbar (15 comps) like STRUCT_LB_HYDROVARS")
    ap.add_argument("--root", default=".")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as ge
    pkg = ge.load_package()
    pf = pkg.plotfile

    if a.system == "interface" and not a.ny:
        a.nx, a.ny, a.nz = 8, 256, 64                                       # :128
    n = (a.nx, a.ny or a.nx, a.nz or a.nx)
    mgs = a.max_grid_size or max(1, a.nx // 2)
    noise = a.kbt != 0.0
    paths = job_paths(a, n)
    os.makedirs(os.path.dirname(paths["plot_root"]), exist_ok=True)
    t0 = time.time()

    params = pkg.default_params(alpha0=a.alpha0, kBT=a.kbt, tau_f=a.tau, tau_g=a.tau, kappa=a.kappa,
                                rho_hi=a.rho_hi, rho_lo=a.rho_lo, seed=a.seed)
    lbm = pkg.BinaryLBM(*n, params=params, device=a.device)
    names = pf.variable_names(22)

    def frame():
        if a.lb_hydrovars:
            out = np.zeros((15,) + lbm.slab_shape)
            out[:9] = lbm.LBM_hydrovars_density()
            return out
        return lbm.LBM_hydrovars()

    def write_output(step):                                                # WriteOutput :35-55
        pf.write_plotfile(pf.concatenate(paths["plot_root"], step, NDIGITS), frame(), names,
                          time=float(step), step=step, max_grid_size=mgs)

    if noise and a.use_ref_state:                                          # :216-235
        print("Noise switch on")
        eq = [pf.read_plotfile(paths["eq"][k])[0][0] for k in ("rho", "phi", "rhot")]
        z, y, x = np.meshgrid(np.arange(n[2]), np.arange(n[1]), np.arange(n[0]), indexing="ij")
        m = eq[0].sum()
        com_ref = np.array([(eq[0] * x).sum(), (eq[0] * y).sum(), (eq[0] * z).sum()]) / m     # update_com(rho_eq), :230
        print("Mass rho_eq = %.17g" % m)
        print("Center of Mass: (%g,%g,%g)" % tuple(com_ref))
        lbm.set_ref_state(eq[0], eq[1], eq[2], com_ref)

    if a.restart:                                                          # :253-270
        chk_temp = 0.0 if a.continue_from_nonfluct else a.kbt
        f0, _ = pf.read_plotfile(checkpoint_name(paths["chk_f"], a.step_continue, a.alpha0, chk_temp, n))
        g0, _ = pf.read_plotfile(checkpoint_name(paths["chk_g"], a.step_continue, a.alpha0, chk_temp, n))
        print("Loading in last frame checkpoint files....")
        lbm.LBM_init(np.ascontiguousarray(f0), np.ascontiguousarray(g0))
        if noise and not a.continue_from_nonfluct:
            # the checkpoint already carries a fluctuating run: continue the noise stream at its absolute step instead of
            # replaying the normals of steps 1..nsteps (the step counter is the generator's noise index)
            lbm.set_steps_done(a.step_continue)
            print("noise index continues at step %d (this segment draws indices %d..%d)" % (a.step_continue, a.step_continue, a.step_continue + a.nsteps - 1))
    elif a.system == "mixture":
        print("Init mixture system ...")
        lbm.LBM_init_mixture()
    elif a.system == "interface":
        print("Init flate interface system ...")
        lbm.LBM_init_stripe(a.init_frac)
    else:
        print("Init droplet system ...")
        lbm.LBM_init_droplet(a.radius)

    h0 = lbm.LBM_hydrovars()
    if not np.all(np.isfinite(h0)):                                        # MultiFabNANCheck :294-297
        print("NaN in the initial hydrodynamic quantities")
        return 1
    if a.plot_int > 0 and a.step_continue == 0:
        write_output(0)                                                    # :314-323
    print("LB initialized with alpha0 = %g and T = %g" % (a.alpha0, a.kbt))

    last = a.step_continue + a.nsteps
    plot_sf = a.plot_sf_window if noise else 0                              # :102
    sf = None                                                               # :310
    if plot_sf > 0:
        # STRUCT_LB_HYDROVARS accumulates hydrovsbar, whose components 9..14 are never written (LBM_binary.H:333-339)
        # and which has no components 15..21: only the pairs inside its 9 defined components are formed
        sf_names = names[:9] if a.lb_hydrovars else names
        sf = (pkg.structfact.DeviceStructFact(lbm, sf_names, lb_hydrovars=a.lb_hydrovars) if a.sf_device
              else pkg.structfact.StructFact(sf_names))
    sf_start = last - a.plot_sf_window                                      # :330
    out_step = a.step_continue + 2 * a.nsteps // 10 if noise else a.step_continue    # :89
    for step in range(a.step_continue + 1, last + 1):                       # :335-387
        lbm.LBM_timestep(1)
        if a.print_int and step % a.print_int == 0 and step % (a.print_int * 50) == 0:
            print("LB step %d" % step)
        if sf is not None and step >= sf_start and step % a.out_sf_step == 0:   # :342-349
            sf.fort_structure(None if a.sf_device else frame(), 0)
        if noise and a.out_noise_step and step % a.out_noise_step == 0:     # WriteOutNoise, Debug.H:380-409
            fn, gn = lbm.thermal_noise()
            base = paths["plot_root"][:-3]
            pf.write_plotfile(pf.concatenate(base + "data_fnoise/fn", step, NDIGITS), fn, pf.noise_names("f"), float(step), step, mgs)
            pf.write_plotfile(pf.concatenate(base + "data_gnoise/gn", step, NDIGITS), gn, pf.noise_names("g"), float(step), step, mgs)
        if a.plot_int > 0 and step % a.plot_int == 0 and step >= out_step and step != last:
            write_output(step)
        if step == last:
            write_output(step)
            if sf is not None and sf.nsamples:                              # WriteOutput(..., plot_SF) :50-54
                sf.write_plotfile(step, float(step), paths["plot_root"] + "_SF", 1, mgs)

    f, g = lbm.populations()                                                # :400-409
    pf.write_plotfile(checkpoint_name(paths["chk_f"], last, a.alpha0, a.kbt, n), f, ["rho_chk"], 0.0, 0, mgs)
    pf.write_plotfile(checkpoint_name(paths["chk_g"], last, a.alpha0, a.kbt, n), g, ["phi_chk"], 0.0, 0, mgs)
    rho_mass, phi_mass = lbm.mass()
    print("mass rho = %.17g phi = %.17g" % (rho_mass, phi_mass))
    print("Run time = %g" % (time.time() - t0))

    if not noise and a.plot_int > 0:                                        # :423-438
        t_window = 5 * a.plot_int                                           # :95
        step1, step2 = max(last - t_window, out_step + (-out_step) % a.plot_int), last
        for key, comp in (("rho", 0), ("phi", 1), ("rhot", 5)):
            mean, dev = print_convergence(pf, paths["plot_root"], step1, step2, a.plot_int, comp)
            print("convergence(%s) = %.6e" % (key, dev))
            pf.write_plotfile(paths["eq"][key], mean[None], [key + "_eq"], 0.0, 0, mgs)
    lbm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
