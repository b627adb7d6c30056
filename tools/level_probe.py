#!/usr/bin/env python3
"""Round 4: the bench scatters over two levels ~8 % apart from process to process at every size (profiles/r04_pitch_scan.txt).
Is the level a property of the ALLOCATION that a context keeps, and does re-creating the context inside one process draw it
again?  Creates the context `trials` times (optionally holding the previous one alive while the next is made, so that the
allocator cannot hand the same memory back), times 3 blocks of steps each time.
usage: level_probe.py SIZE [trials] [hold]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes
import __graft_entry__ as ge
pkg = ge.load_package()
lib = pkg._lib.load()


def addresses(l):
    out = (ctypes.c_ulonglong * 8)()
    assert lib.bflbm_debug_addresses(l._h, out) == 0
    return list(out)
n = int(sys.argv[1]); trials = int(sys.argv[2]) if len(sys.argv) > 2 else 10; hold = len(sys.argv) > 3 and sys.argv[3] == "hold"
steps = max(10, int(4e9 / n ** 3))
prev = None
for t in range(trials):
    l = pkg.BinaryLBM(n, n, n)
    if prev is not None:
        prev.close()
    l.LBM_init_stripe(0.5)
    l.LBM_timestep(5); l.sync()
    vals = []
    for _ in range(3):
        l.timer_start(); l.LBM_timestep(steps); ms = l.timer_stop() / steps
        vals.append(n ** 3 / ms / 1e3)
    # the same context re-initialised: does the level follow the allocation or the state?
    l.LBM_init_stripe(0.5); l.LBM_timestep(5); l.sync()
    l.timer_start(); l.LBM_timestep(steps); ms = l.timer_stop() / steps
    a = addresses(l)
    print(f"{n}^3 trial {t:2d} {'(held)' if hold and t else ''} placement {l.placement_report()}: blocks {[round(v) for v in vals]} MLUPS; after re-init {n ** 3 / ms / 1e3:.0f}"
          f" | A {a[0]:#x} B-A {a[1] - a[0]:#x} rho {a[2]:#x} phi {a[3]:#x} scratch {a[4]:#x} frames {a[5]:#x} {a[6]:#x} vol {a[7]}", flush=True)
    if hold:
        prev = l
    else:
        l.close()
if prev is not None:
    prev.close()
