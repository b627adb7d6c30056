#!/bin/bash
# A/B timing of library variants on the GPU box:  tools/ab.sh "<bench args>" lib1 lib2 ...   ("default" = in-tree lib)
# Each variant is run twice, interleaved, to see box noise.
args="$1"; shift
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = default ]; then unset BFLBM_LIB; else export BFLBM_LIB="$v"; fi
    out=$(timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1) || { echo "FAILED $v: $out"; exit 1; }
    python - "$v" "$out" <<'PY'
import sys, json
d = json.loads(sys.argv[2]); print("%-40s %8.1f MLUPS %8.4f ms  frac %.3f" % (sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["frac"]))
PY
  done
done
