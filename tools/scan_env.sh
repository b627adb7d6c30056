#!/bin/bash
# scan one environment variable over values, R runs each: tools/scan_env.sh VAR "v1 v2 ..." R "<bench args>" [extra env]
var=$1; vals="$2"; reps=$3; args="$4"; extra="$5"
for v in $vals; do
  res=""
  for r in $(seq $reps); do
    out=$(env $extra $var=$v timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1)
    res="$res $(python -c "import sys,json; print(json.loads(sys.argv[1])['value'])" "$out")"
  done
  echo "$var=$v :$res"
done
