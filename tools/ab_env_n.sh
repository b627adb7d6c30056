#!/bin/bash
# like ab_env.sh with N interleaved repetitions and a median per setting:  tools/ab_env_n.sh N "<bench args>" "VAR=a" "-" ...
n=$1; args="$2"; shift 2
tmp=$(mktemp)
for rep in $(seq $n); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then out=$(timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1)
    else out=$(env $v timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>&1 | tail -1); fi
    python - "$v" "$out" >> $tmp <<'PY'
import sys, json
d = json.loads(sys.argv[2]); print(sys.argv[1], d["value"])
PY
  done
done
python - $tmp <<'PY'
import sys, statistics, collections
r = collections.defaultdict(list)
for l in open(sys.argv[1]):
    k, v = l.rsplit(" ", 1); r[k].append(float(v))
for k, v in r.items():
    print("%-32s median %8.1f  min %8.1f  max %8.1f  (%d runs)" % (k, statistics.median(v), min(v), max(v), len(v)))
PY
