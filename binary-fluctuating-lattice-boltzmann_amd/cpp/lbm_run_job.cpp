// lbm_run_job.cpp -- a driver with the call sequence of the reference's live driver
// (main_run_job.cpp: domain/BoxArray/nghost/nhydro :136-147, MultiFab set :205-212, init dispatch
// :274-286, time loop :335-339, mass print :414), written against include/bflbm_amrex.H and this
// project's host container.  It demonstrates that the operator surface is drop-in: the LBM_* calls
// below are textually the reference's.
//
// usage: lbm_run_job <nx> [<ny> <nz>] <nsteps> <mixture|stripe|droplet> [kBT] [alpha0] [sync 0|1|2] [plot_root]
// With plot_root the final hydrovs frame, the noise and the f/g checkpoints are written as AMReX
// plotfiles named like the reference's (main_run_job.cpp:42, :400-409; Debug.H:390-408).
//
// LBM_SF_WINDOW=w [LBM_SF_STEP=s]: structure factors like main_run_job.cpp:301-310, :342-349, :50-54 --
// FortStructure(hydrovs, 0) every s steps inside the last w steps, WritePlotFile with the last frame.
//
// LBM_RESTART_FROM=<plot_root> LBM_RESTART_STEP=<n>: the `if_continue_from_last_frame` branch of the live driver
// (main_run_job.cpp:253-270): LoadSingleMultiFab of <plot_root>/f_checkpoint<n>, g_checkpoint<n> from DISK, then
// LBM_init instead of the init dispatch.  LBM_EDIT_CHECK=1: after the run the driver scales fold on the host and calls
// LBM_hydrovars_density(geom, fold, gold, hydrovsbar) -- the operator must evaluate the populations it is handed.
//
// Built a second time with -DUSE_REF_STATE (lbm_run_job_ref), like the reference's compile-time switch
// (LBM_binary.H:12): the run then mirrors the noiseSwitch flow of main_run_job.cpp:216-235, :253-270 --
// equilibrium fields into rho_eq/phi_eq/rhot_eq (the initial state's hydrovs 0, 1, 5, or, with
// LBM_EQ_FROM=<plot_root>, LoadSingleMultiFab of <plot_root>/equilibrium_rho, _phi, _rhot written by a kBT = 0 run of
// this driver, and the populations of its checkpoints), com_ref from update_com(rho_eq) (moved by (2.6,-1.4,3.3) so
// that the lookup is shifted), continuation through LBM_init, then the time loop.
#include <algorithm>
#include <array>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "host_multifab.H"
using namespace bflbm::host;
// #define USE_REF_STATE  -- from the command line (-DUSE_REF_STATE), before the adapter like before LBM_binary.H
#include "../../include/bflbm_amrex.H"
#include "../../include/bflbm_plotfile.H"
#include "../../include/bflbm_structfact.H"
using StructFact = bflbm::StructFact;                       // FHDeX's class in the reference's driver

int main(int argc, char* argv[]) {
  if (argc < 4) { std::fprintf(stderr, "usage: %s nx [ny nz] nsteps system [kBT] [alpha0] [sync]\n", argv[0]); return 2; }
  int a = 1;
  int nx = std::atoi(argv[a++]), ny = nx, nz = nx;
  if (argc - a >= 4 && std::atoi(argv[a]) > 0 && std::atoi(argv[a + 1]) > 0 && std::atoi(argv[a + 2]) > 0 && !std::isalpha(argv[a + 2][0])) {
    ny = std::atoi(argv[a++]); nz = std::atoi(argv[a++]);
  }
  const int nsteps = std::atoi(argv[a++]);
  const std::string system = argv[a++];
  if (a < argc) kBT = std::atof(argv[a++]);
  if (a < argc) alpha0 = std::atof(argv[a++]);
  if (a < argc) bflbm::set_sync(std::atoi(argv[a++]));
  const std::string plot_root = (a < argc) ? argv[a++] : "";

  const int max_grid_size = std::max(1, nx / 2);            // main_run_job.cpp:73
  Box domain(IntVect3{{0, 0, 0}}, IntVect3{{nx - 1, ny - 1, nz - 1}});
  Geometry geom(domain, {1, 1, 1});                         // :136-139
  BoxArray ba(domain);
  ba.maxSize(max_grid_size);                                // :142
  const int nghost = 2;                                     // :145
  // LBM_LEGACY=1: the stale drivers' settings -- nhydro = 15 (main_driver.cpp:165) and the 12-argument
  // LBM_timestep without com_ref (main_driver.cpp:336); the adapter must never write past nComp().
  const bool legacy = std::getenv("LBM_LEGACY") && std::atoi(std::getenv("LBM_LEGACY")) != 0;
  const int nhydro = legacy ? 15 : 22;                      // :147

  MultiFab rho_eq(ba, 1, nghost, domain), phi_eq(ba, 1, nghost, domain), rhot_eq(ba, 1, nghost, domain);
  rhot_eq.setVal(1.);
  MultiFab fold(ba, nvel, nghost, domain), fnew(ba, nvel, nghost, domain);    // :205-212
  MultiFab gold(ba, nvel, nghost, domain), gnew(ba, nvel, nghost, domain);
  MultiFab hydrovs(ba, nhydro, nghost, domain);
  MultiFab hydrovsbar(ba, 15, nghost, domain);
  MultiFab fnoisevs(ba, nvel, nghost, domain), gnoisevs(ba, nvel, nghost, domain);
  std::vector<std::array<double, 3>> com_ref(3, {nx / 2., nx / 2., nx / 2.});   // :117-119

  const char* restart_from = std::getenv("LBM_RESTART_FROM");
  if (restart_from) {                                       // if_continue_from_last_frame, :253-270
    const int rstep = std::getenv("LBM_RESTART_STEP") ? std::atoi(std::getenv("LBM_RESTART_STEP")) : 0;
    MultiFab f_last_frame(ba, nvel, nghost, domain), g_last_frame(ba, nvel, nghost, domain);
    std::printf("Loading in last frame checkpoint files....\n");
    if (!bflbm::LoadSingleMultiFab(bflbm::Concatenate(std::string(restart_from) + "/f_checkpoint", rstep), f_last_frame) ||
        !bflbm::LoadSingleMultiFab(bflbm::Concatenate(std::string(restart_from) + "/g_checkpoint", rstep), g_last_frame)) return 3;
    if (kBT != 0. && !std::getenv("LBM_RESTART_NOISE_FROM_ZERO")) { // continue the noise stream at the checkpoint's step (restart == uninterrupted run)
      bflbm::set_restart_step(rstep);
      std::printf("noise index continues at step %d\n", rstep);
    }
    LBM_init(geom, fold, gold, hydrovs, hydrovsbar, fnoisevs, gnoisevs, f_last_frame, g_last_frame, rho_eq, phi_eq, rhot_eq, com_ref);
  } else
  if (system == "mixture")      LBM_init_mixture(geom, fold, gold, hydrovs, hydrovsbar, fnoisevs, gnoisevs, rho_eq, phi_eq, rhot_eq);
  else if (system == "stripe")  LBM_init_stripe(0.5, geom, fold, gold, hydrovs, hydrovsbar, fnoisevs, gnoisevs, rho_eq, phi_eq, rhot_eq);
  else if (system == "droplet") LBM_init_droplet(0.2, geom, fold, gold, hydrovs, hydrovsbar, fnoisevs, gnoisevs, rho_eq, phi_eq, rhot_eq);
  else { std::fprintf(stderr, "unknown system %s\n", system.c_str()); return 2; }
  std::printf("LB initialized with alpha0 = %g and T = %g, %zu boxes of max size %d\n", (double)alpha0, (double)kBT, ba.size(), max_grid_size);
#ifdef USE_REF_STATE
  {
    bflbm::materialize(geom, fold, gold, hydrovs, hydrovsbar, fnoisevs, gnoisevs);
    MultiFab f0(ba, nvel, nghost, domain), g0(ba, nvel, nghost, domain);
    const char* eq_from = std::getenv("LBM_EQ_FROM");
    if (eq_from) {                                          // noiseSwitch: the equilibrium state from DISK, :216-220
      const int estep = std::getenv("LBM_EQ_STEP") ? std::atoi(std::getenv("LBM_EQ_STEP")) : 0;
      const std::string r(eq_from);
      if (!bflbm::LoadSingleMultiFab(r + "/equilibrium_rho", rho_eq) || !bflbm::LoadSingleMultiFab(r + "/equilibrium_phi", phi_eq) ||
          !bflbm::LoadSingleMultiFab(r + "/equilibrium_rhot", rhot_eq)) return 3;
      if (!bflbm::LoadSingleMultiFab(bflbm::Concatenate(r + "/f_checkpoint", estep), f0) ||
          !bflbm::LoadSingleMultiFab(bflbm::Concatenate(r + "/g_checkpoint", estep), g0)) return 3;
      std::printf("Mass rho_eq = %.17g\n", rho_eq.sum(0));   // :224
    } else {
      for (MFIter mfi(hydrovs); mfi.isValid(); ++mfi) {
        const Box& v = mfi.validbox();
        for (int z = v.smallEnd(2); z <= v.bigEnd(2); ++z) for (int y = v.smallEnd(1); y <= v.bigEnd(1); ++y) for (int x = v.smallEnd(0); x <= v.bigEnd(0); ++x) {
          rho_eq[mfi](x, y, z, 0) = hydrovs[mfi](x, y, z, 0);
          phi_eq[mfi](x, y, z, 0) = hydrovs[mfi](x, y, z, 1);
          rhot_eq[mfi](x, y, z, 0) = hydrovs[mfi](x, y, z, 5);
        }
      }
      bflbm::pull_field(geom, bflbm::F_POPS, f0, &g0);
    }
    bflbm::invalidate_ref_state();                          // the fields changed after the first LBM_* call
    update_com(geom, com_ref[0], rho_eq, true);             // main_run_job.cpp:230-233
    const double off[3] = {2.6, -1.4, 3.3};
    for (int d = 0; d < 3; ++d) com_ref[0][d] -= off[d];
    LBM_init(geom, fold, gold, hydrovs, hydrovsbar, fnoisevs, gnoisevs, f0, g0, rho_eq, phi_eq, rhot_eq, com_ref);   // :269-270
  }
#endif

  // set up StructFact (:299-310)
  const std::vector<int> pairA = { 0,  1,  0,  2,  3,  4,  6,  7,  8,  2,  9,  15, 16, 17, 15, 18, 19, 20, 21, 20, 20, 21};
  const std::vector<int> pairB = { 0,  1,  1,  2,  3,  4,  6,  7,  8,  6,  9,  15, 16, 17, 16, 18, 19, 20, 21, 21, 18, 18};
  const std::vector<double> var_scaling(22, 1.0);
  const std::vector<std::string> var_names = bflbm::VariableNames(nhydro);
  const int dm = 0;                                         // no DistributionMapping in a single process
  StructFact structFact(ba, dm, var_names, var_scaling, pairA, pairB);
  const int plot_SF_window = std::getenv("LBM_SF_WINDOW") ? std::atoi(std::getenv("LBM_SF_WINDOW")) : 0;   // :99
  const int out_SF_step = std::getenv("LBM_SF_STEP") ? std::atoi(std::getenv("LBM_SF_STEP")) : 100;        // :100
  const int SF_start = nsteps - plot_SF_window;             // :330

  for (int step = 1; step <= nsteps; ++step) {              // :335-339
    if (legacy) LBM_timestep(geom, fold, gold, fnew, gnew, hydrovs, hydrovsbar, fnoisevs, gnoisevs, rho_eq, phi_eq, rhot_eq);
    else        LBM_timestep(geom, fold, gold, fnew, gnew, hydrovs, hydrovsbar, fnoisevs, gnoisevs, rho_eq, phi_eq, rhot_eq, com_ref);
    if (plot_SF_window > 0 && step >= SF_start && step % out_SF_step == 0) structFact.FortStructure(hydrovs, 0);   // :342-349
  }
  if (plot_SF_window > 0 && !plot_root.empty()) {
    structFact.WritePlotFile(nsteps, (double)nsteps, plot_root + "/plt_SF", 1);                           // :50-54
    std::printf("sf_samples %lld\n", structFact.nsamples());
  }
  // if_print_radius (:111, :364-367): the reference's own radius fit of rho, here on the resident state
  if (std::getenv("LBM_PRINT_RADIUS") && system == "droplet") {
    const double W0 = std::getenv("LBM_RADIUS_W0") ? std::atof(std::getenv("LBM_RADIUS_W0")) : (double)kappa;   // the driver passes kappa, :365
    const std::array<double, 3> param_arr = bflbm::fittingDropletParams(geom, 20, 0.01, 400, W0, 0.2);
    std::printf("fitting parameters for equilibrium density rho: (W=%f, R=%f)\n", param_arr[0], param_arr[1]);
    std::printf("radius_fit %.17g %.17g %.3e\n", param_arr[0], param_arr[1], param_arr[2]);
  }
  // restart path of the live driver (:253-270): continue from the populations just produced
  if (std::getenv("LBM_RESTART_CHECK") && std::atoi(std::getenv("LBM_RESTART_CHECK")) != 0) {
    MultiFab f_last(ba, nvel, nghost, domain), g_last(ba, nvel, nghost, domain);
    bflbm::pull_field(geom, bflbm::F_POPS, f_last, &g_last);
    LBM_init(geom, fold, gold, hydrovs, hydrovsbar, fnoisevs, gnoisevs, f_last, g_last, rho_eq, phi_eq, rhot_eq, com_ref);
  }
  bflbm::materialize(geom, fold, gold, hydrovs, hydrovsbar, fnoisevs, gnoisevs);   // no-op cost when sync == 2

  // sequential sum over cells in x,y,z order (MultiFabValidSum-like, Debug.H:35-46)
  double tot = 0.0;
  for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) tot += hydrovs.at(x, y, z, 5);
  std::array<double, 3> com;
  update_com(geom, com, hydrovsbar);
  std::printf("step %d\n", nsteps);
  std::printf("total_mass %.17g\n", tot);
  std::printf("rho_mass %.17g phi_mass %.17g\n", hydrovsbar.sum(0), hydrovsbar.sum(1));
  if (nz > 4) std::printf("rho(0,0,4) %.17g\n", hydrovsbar.at(0, 0, 4, 0));
  if (nz > 2) std::printf("ufz(0,0,2) %.17g\n", hydrovs.at(0, 0, 2, 4));
  std::printf("fold(1,1,1,3) %.17g gold(1,1,1,7) %.17g\n", fold.at(1 % nx, 1 % ny, 1 % nz, 3), gold.at(1 % nx, 1 % ny, 1 % nz, 7));
  std::printf("com %.12g %.12g %.12g\n", com[0], com[1], com[2]);
  std::printf("fnoise(1,2,3,4) %.17g gnoise(3,2,1,18) %.17g\n", fnoisevs.at(1 % nx, 2 % ny, 3 % nz, 4), gnoisevs.at(3 % nx, 2 % ny, 1 % nz, 18));
  // ghost cells must hold the periodic image after the adapter's FillBoundary
  MFIter it(hydrovsbar);
  std::printf("ghost_check %d\n", (int)(hydrovsbar[it](-1, 0, 0, 0) == hydrovsbar.at(nx - 1, 0, 0, 0)));
  if (!plot_root.empty()) {
    bool ok = bflbm::WriteSingleLevelPlotfile(bflbm::Concatenate(plot_root + "/plt", nsteps), hydrovs, bflbm::VariableNames(nhydro), geom, (double)nsteps, nsteps);
    std::vector<std::string> nf, ng;
    for (int k = 0; k < nvel; ++k) { nf.push_back("fa" + std::to_string(k)); ng.push_back("ga" + std::to_string(k)); }
    ok = ok && bflbm::WriteSingleLevelPlotfile(bflbm::Concatenate(plot_root + "/fn", nsteps), fnoisevs, nf, geom, (double)nsteps, nsteps);
    ok = ok && bflbm::WriteSingleLevelPlotfile(bflbm::Concatenate(plot_root + "/gn", nsteps), gnoisevs, ng, geom, (double)nsteps, nsteps);
    ok = ok && bflbm::WriteSingleLevelPlotfile(bflbm::Concatenate(plot_root + "/f_checkpoint", nsteps), fold, {"rho_chk"}, geom, 0., 0);   // 1 name, 19 comps (:406-407)
    ok = ok && bflbm::WriteSingleLevelPlotfile(bflbm::Concatenate(plot_root + "/g_checkpoint", nsteps), gold, {"phi_chk"}, geom, 0., 0);
    if (kBT == 0.) {
      // equilibrium state for a later noise run (:423-438; the reference averages its last frames, PrintConvergence,
      // Debug.H:275-358 -- a converged kBT = 0 run is stationary, here the final frame is written)
      const char* nm[3] = {"rho", "phi", "rhot"};
      const int cmp[3] = {0, 1, 5};
      for (int q = 0; q < 3; ++q) {
        MultiFab one(ba, 1, nghost, domain);
        for (MFIter mfi(one); mfi.isValid(); ++mfi) {
          const Box& v = mfi.validbox();
          for (int z = v.smallEnd(2); z <= v.bigEnd(2); ++z) for (int y = v.smallEnd(1); y <= v.bigEnd(1); ++y) for (int x = v.smallEnd(0); x <= v.bigEnd(0); ++x)
            one[mfi](x, y, z, 0) = hydrovs[mfi](x, y, z, cmp[q]);
        }
        ok = ok && bflbm::WriteSingleLevelPlotfile(plot_root + "/equilibrium_" + nm[q], one, {std::string(nm[q]) + "_eq"}, geom, (double)nsteps, nsteps);
      }
    }
    std::printf("plotfiles %d\n", (int)ok);
  }
  if (std::getenv("LBM_EDIT_CHECK") && std::atoi(std::getenv("LBM_EDIT_CHECK")) != 0) {
    // a driver that edits the populations on the host and asks for the densities of WHAT IT HOLDS (LBM_binary.H:343-354)
    for (MFIter mfi(fold); mfi.isValid(); ++mfi) {
      const Box& v = mfi.validbox();
      for (int c = 0; c < nvel; ++c) for (int z = v.smallEnd(2); z <= v.bigEnd(2); ++z) for (int y = v.smallEnd(1); y <= v.bigEnd(1); ++y) for (int x = v.smallEnd(0); x <= v.bigEnd(0); ++x)
        fold[mfi](x, y, z, c) *= 1.5;
    }
    const double before = hydrovsbar.at(0, 0, nz > 4 ? 4 : 0, 0);
    LBM_hydrovars_density(geom, fold, gold, hydrovsbar);
    std::printf("edit_check rho_before %.17g rho_after %.17g\n", before, hydrovsbar.at(0, 0, nz > 4 ? 4 : 0, 0));
  }
  bflbm::shutdown();
  return 0;
}
