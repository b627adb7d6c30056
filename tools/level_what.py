#!/usr/bin/env python3
"""Which allocation decides a context's level (profiles/r04_level_probe.txt): the state or the hand-over frames?  Draws only one of
them again (BFLBM_TUNE_WHAT=state|frames, read by bflbm_tune_placement) four times per round and prints the candidates' step times.
usage: level_what.py SIZE state|frames|both [rounds]"""
import os, sys
what = sys.argv[2]
if what != "both":
    os.environ["BFLBM_TUNE_WHAT"] = what
os.environ["BFLBM_PLACEMENT_CANDIDATES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
n = int(sys.argv[1]); rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
l = pkg.BinaryLBM(n, n, n)
for r in range(rounds):
    print(f"{n}^3 draw {what:6s} round {r}: {l.tune_placement(4)}", flush=True)
l.close()
