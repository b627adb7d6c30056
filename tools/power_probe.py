#!/usr/bin/env python3
"""Is the data dependence of the step time (profiles/r04_sustained.txt) clock / power management?  Samples `rocm-smi --showclocks --showpower`
every ~0.3 s in a thread while the main thread runs, 6 s each: the quiet kernel on the smooth stripe, the quiet kernel on a state that noise
has randomised, the noisy kernel on that state.  Prints the step time and the median / range of sclk and package power per phase."""
import os, re, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
n = 512
samples, stop, phase = [], False, ["idle"]


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            mclk = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", out)
            pw = re.search(r"Power \(W\): ([\d.]+)", out)
            samples.append((phase[0], int(sclk.group(1)) if sclk else -1, int(mclk.group(1)) if mclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception as e:      # noqa: BLE001
            samples.append((phase[0], -1, -1, -1.0))
        time.sleep(0.25)


l = pkg.BinaryLBM(n, n, n, params=pkg.default_params(kBT=0.0, alpha0=0.0))
th = threading.Thread(target=sampler, daemon=True); th.start()


def run(tag, seconds=6.0):
    phase[0] = tag
    l.sync(); t0 = time.time(); steps = 0
    l.timer_start()
    while time.time() - t0 < seconds:
        l.LBM_timestep(20); l.sync(); steps += 20
    ms = l.timer_stop() / steps
    phase[0] = "idle"
    mine = [s for s in samples if s[0] == tag and s[1] > 0]
    import statistics
    if mine:
        print(f"{tag:42s} {ms:7.3f} ms/step  sclk median {statistics.median(s[1] for s in mine):6.0f} MHz (min {min(s[1] for s in mine)}, max {max(s[1] for s in mine)})"
              f"  mclk {statistics.median(s[2] for s in mine):5.0f}  power median {statistics.median(s[3] for s in mine):6.0f} W (max {max(s[3] for s in mine):.0f})  [{len(mine)} samples]", flush=True)
    else:
        print(f"{tag:42s} {ms:7.3f} ms/step  (no rocm-smi samples)", flush=True)


l.LBM_init_stripe(0.5); l.LBM_timestep(5)
run("quiet kernel, smooth stripe")
l.set_params(kBT=1e-5); l.LBM_init_mixture(); l.LBM_timestep(60)
run("noisy kernel, randomised mixture")
l.set_params(kBT=0.0)
run("quiet kernel, the randomised state")
l.LBM_init_stripe(0.5); l.LBM_timestep(5)
run("quiet kernel, smooth stripe again")
stop = True
l.close()
