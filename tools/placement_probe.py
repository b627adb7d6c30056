#!/usr/bin/env python3
"""Does the step time of one process depend on WHERE its state lands in device memory?  The bench scatters over discrete
levels from process to process (7600 ... 8400 MLUPS at 512^3 on one box).  Here ONE process creates the 512^3 context several
times, holding spacer allocations of different sizes in between, and times 10 steps each time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
n = 512
spacers = []
for trial, mb in enumerate([0, 0, 2, 0, 64, 0, 1024, 0, 3000, 0, 7000]):
    if mb:
        spacers.append(torch.empty(mb * 1024 * 1024, dtype=torch.uint8, device="cuda"))
    l = pkg.BinaryLBM(n, n, n)
    p0 = l.debug_time_kernel(0, 2)        # on the zeroed buffers, right after the allocation
    l.LBM_init_stripe(0.5)
    p1 = l.debug_time_kernel(0, 2)        # on the initial state
    l.LBM_init_stripe(0.5)
    l.LBM_timestep(3); l.sync()
    l.timer_start(); l.LBM_timestep(10); ms = l.timer_stop() / 10
    pull = l.debug_time_kernel(0, 3)      # the pull-copy calibration kernel on the same two buffers (38 shifted reads + 38 writes per site)
    dens = l.debug_time_kernel(1, 3)      # the density pass (38 reads)
    print(f"trial {trial:2d}  spacer held so far {sum(s.numel() for s in spacers) >> 20:6d} MiB   {n**3 / ms / 1e3:8.1f} MLUPS  {ms:.3f} ms   pull-copy {pull:.3f} ms (zeroed {p0:.3f}, initial state {p1:.3f})  density pass {dens:.3f} ms", flush=True)
    l.close()
