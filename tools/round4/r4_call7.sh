#!/bin/bash
# Round 4, GPU call 7: row-pitch scan (BFLBM_PITCH = row pitch in doubles), 3 interleaved repetitions.
out=gpurun_out/r4_call7; rm -rf $out; mkdir -p $out
run() { timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"; }
for rep in 1 2 3; do
  for s in 256 384 448 512; do
    for d in 0 32 64 96 128 160 192 224 256; do
      p=$((s + d))
      v=$(BFLBM_PITCH=$p run --size $s)
      echo "rep $rep size $s pitch $p -> $v" >> $out/pitch_scan.txt
    done
  done
  echo "rep $rep done"
done
python3 - $out/pitch_scan.txt <<'PY'
import sys, collections, statistics
r = collections.defaultdict(list)
for l in open(sys.argv[1]):
    t = l.split()
    r[(int(t[3]), int(t[5]))].append(float(t[7]))
for (s, p), v in sorted(r.items()):
    print(f"size {s} pitch {p} (+{p - s}): median {statistics.median(v):7.0f}  runs {[round(x) for x in v]}")
PY
