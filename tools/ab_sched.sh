#!/bin/bash
# A/B of kernel schedules:  tools/ab_sched.sh "<sizes>" sched1 sched2 ...   (interleaved, two repetitions)
sizes="$1"; shift
for size in $sizes; do
  echo "== size $size"
  for rep in 1 2; do
    for s in "$@"; do
      out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --size $size --steps 60 --warmup 5 --schedule $s 2>&1 | tail -1)
      python - "$s" "$out" <<'PY'
import sys, json
d = json.loads(sys.argv[2]); print("%-24s %8.1f MLUPS %8.4f ms  frac %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
    done
  done
done
