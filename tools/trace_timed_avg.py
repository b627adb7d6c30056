#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace CSV of a bench.py run: the average launch duration of the step's kernel over the TIMED region
only -- the last (steps x blocks) dispatches of that kernel -- beside the average over all of its dispatches.  Since round 4 a
context times a few steps on up to four candidate allocations when it is created (bflbm_tune_placement); those probe
launches, some of them on allocations that were then discarded, are in the trace and in rocprofv3's --stats average, and are
not the step the bench reports.  usage: trace_timed_avg.py kernel_trace.csv steps blocks [kernel-substring]"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps, blocks = int(sys.argv[2]), int(sys.argv[3])
want = sys.argv[4] if len(sys.argv) > 4 else None
by = {}
for r in rows:
    name = (r.get("Kernel_Name") or r.get("kernel_name")).split("(")[0].replace("void ", "")
    by.setdefault(name, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
if want is None:
    want = max(by, key=lambda k: sum(e - s for s, e in by[k]))
else:
    want = next(k for k in by if want in k)
d = sorted(by[want])
dur = [(e - s) * 1e-6 for s, e in d]
timed = dur[-steps * blocks:]
print(json.dumps({"kernel": want, "launches_all": len(dur), "avg_all_ms": round(sum(dur) / len(dur), 4),
                  "launches_timed": len(timed), "avg_timed_ms": round(sum(timed) / len(timed), 4), "min_timed_ms": round(min(timed), 4),
                  "rule": f"last {steps} x {blocks} dispatches of the kernel = the bench's timed blocks (warm-up, init and the placement probes come before them)"}))
