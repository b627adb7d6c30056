#!/bin/bash
# one box: how the length of a march (planes per chunk) and the number of rounds change the cost per plane
run() { python bench.py --shape $1 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-14s %-22s %8.1f MLUPS %8.4f ms' % ('$1', '$2', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  run 256,256,128 "-"
  run 256,256,256 "-"
  BFLBM_FUSED_WG=512 run 256,256,256 "WG=512"
  run 256,256,512 "-"
  BFLBM_FUSED_WG=256 run 256,256,512 "WG=256"
  BFLBM_FUSED_WG=1024 run 256,256,512 "WG=1024"
  run 512,512,128 "-"
  run 512,512,256 "-"
done
