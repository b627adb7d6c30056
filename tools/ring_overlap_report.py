#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of tools/ring_trace.py: does the halo transport run INSIDE the interior sweeps?
Prints every kernel of the last step with start / end relative to the step's first kernel and its hardware queue, and for
each halo kernel (k_halo_pull) how much of its duration lies inside a running sweep kernel (k_fused_ho) of ANOTHER queue.
usage: ring_overlap_report.py kernel_trace.csv nslabs"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nslabs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ks = []
for r in rows:
    name = r.get("Kernel_Name") or r.get("kernel_name")
    ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name.split("(")[0].replace("void ", ""), r.get("Queue_Id", r.get("Stream_Id", "?"))))
ks.sort()
sweeps = [k for k in ks if "k_fused_ho" in k[2]]
pulls = [k for k in ks if "k_halo_pull" in k[2]]
print(f"{len(ks)} kernels, {len(sweeps)} k_fused_ho launches, {len(pulls)} k_halo_pull launches, hardware queues {sorted(set(k[3] for k in ks))}")
if not sweeps or not pulls:
    sys.exit("no k_fused_ho / k_halo_pull in the trace")
# a step = per slab one boundary launch + one interior sweep (2 nslabs k_fused_ho) and 2 nslabs k_halo_pull; the last step begins
# at the (2 nslabs)-th last sweep launch and ends with the last pull
t0 = sweeps[-2 * nslabs][0]
step_k = [k for k in ks if k[0] >= t0]
t1 = max(k[1] for k in step_k)
print(f"last step: {len(step_k)} kernels over {(t1 - t0) / 1e6:.3f} ms")
for k in step_k:
    print(f"  {(k[0] - t0) / 1e6:9.3f} -> {(k[1] - t0) / 1e6:9.3f} ms  {(k[1] - k[0]) / 1e6:8.3f} ms  queue {k[3]}  {k[2][:48]}")
step_sweeps = [k for k in step_k if "k_fused_ho" in k[2]]
step_pulls = [k for k in step_k if "k_halo_pull" in k[2]]
tot = cov = 0
fully = 0
for p in step_pulls:
    segs = sorted((max(p[0], s[0]), min(p[1], s[1])) for s in step_sweeps if s[3] != p[3] and s[0] < p[1] and p[0] < s[1])
    c, cur = 0, p[0]
    for a, b in segs:
        a = max(a, cur)
        if b > a:
            c += b - a
            cur = b
    tot += p[1] - p[0]; cov += c
    fully += c >= 0.999 * (p[1] - p[0])
print(f"halo kernels of the last step: {len(step_pulls)}, {tot / 1e6:.3f} ms in total; {100.0 * cov / max(tot, 1):.1f} % of that time a sweep kernel of another queue "
      f"was running; {fully} of {len(step_pulls)} ran entirely inside running sweeps")
last_sweep_end = max(s[1] for s in step_sweeps)
print(f"step tail after the last sweep ended: {(t1 - last_sweep_end) / 1e6:.3f} ms of {(t1 - t0) / 1e6:.3f} ms")
