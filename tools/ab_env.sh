#!/bin/bash
# A/B of environment settings on the GPU box:  tools/ab_env.sh "<bench args>" "VAR=a" "VAR=b" ...   ("-" = no setting)
args="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then out=$(timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1)
    else out=$(env $v timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1); fi
    python - "$v" "$out" <<'PY'
import sys, json
d = json.loads(sys.argv[2]); print("%-40s %8.1f MLUPS %8.4f ms  frac %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
  done
done
