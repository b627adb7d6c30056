#!/usr/bin/env python3
"""Why does the noisy step get slower over the first second of a run (profiles/r04_sustained.txt)?  Times blocks of 50 steps at 512^3:
  A  noise on, 12 blocks from a fresh mixture          -- the trend
  B  a pause of 8 s, then 4 more blocks                 -- clocks / power recover in a pause, data does not
  C  kBT = 0 on the SAME (by now noisy) state, 6 blocks -- the quiet kernel on random data
  D  fresh stripe state, kBT = 0, 6 blocks              -- the quiet kernel on smooth data
usage: sustain_probe.py [size]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
l = pkg.BinaryLBM(n, n, n, params=pkg.default_params(kBT=1e-5, alpha0=0.0))
print("placement", l.placement_report(), flush=True)
def blocks(tag, k, steps=50):
    out = []
    for _ in range(k):
        l.timer_start(); l.LBM_timestep(steps); out.append(round(l.timer_stop() / steps, 3))
    print(tag, out, flush=True)
l.LBM_init_mixture(); l.LBM_timestep(5); l.sync()
blocks("A noise, fresh mixture      ", 12)
time.sleep(8)
blocks("B noise, after an 8 s pause ", 4)
l.set_params(kBT=0.0)
print("schedule now", l.resolved_schedule(), flush=True)
blocks("C quiet kernel, noisy state ", 6)
l.LBM_init_stripe(0.5); l.LBM_timestep(5); l.sync()
blocks("D quiet kernel, stripe state", 6)
l.set_params(kBT=1e-5)
l.LBM_init_stripe(0.5); l.LBM_timestep(5); l.sync()
blocks("E noise kernel, stripe state (alpha0 = 0)", 6)
l.close()
