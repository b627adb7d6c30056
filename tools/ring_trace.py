#!/usr/bin/env python3
"""Workload for the overlap trace of the native ring on ONE GPU: k slabs of NX x NY x NZ, default
schedule (auto -> hand-over in the interior sweeps), a few steps.  Run under rocprofv3 --kernel-trace and feed the
kernel_trace.csv to tools/ring_overlap_report.py.
usage: ring_trace.py NX NY NZ K [steps] [kernel|copy]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
pkg = ge.load_package()
nx, ny, nz, k = (int(v) for v in sys.argv[1:5])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
transport = sys.argv[6] if len(sys.argv) > 6 else "kernel"
r = pkg.RingLBM(nx, ny, nz * k, nslabs=k, devices=(0,))
r.set_transport(transport)
r.LBM_init_droplet(0.2)
r.LBM_timestep(steps)
r.sync()
print("schedules", [s.resolved_schedule() for s in r.slabs], "faces (kernel, copy)", r.last_transport())
r.close()
