#!/bin/bash
# N interleaved repetitions of library variants, median per variant:  tools/ab_n.sh N "<bench args>" lib1 lib2 ...  ("default" = in-tree lib)
n=$1; args="$2"; shift 2
tmp=$(mktemp)
for rep in $(seq $n); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset BFLBM_LIB; else export BFLBM_LIB="$v"; fi
    out=$(timeout -k 10 300 python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1)
    python - "$(basename $v)" "$out" >> $tmp <<'PY'
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
    print("%-28s median %8.1f  min %8.1f  max %8.1f  runs %s" % (k, statistics.median(v), min(v), max(v), [round(x) for x in v]))
PY
