#!/bin/bash
# Round 4, GPU call 17: the layout knobs of rounds 1-3 (component pad, A/B displacement, workgroup strip width) re-scanned with the placement tuning on.
out=gpurun_out/r4_call17; rm -rf $out; mkdir -p $out
run() { timeout -k 10 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"; }
for rep in 1 2; do
  for s in 256 512; do
    for cfg in "default" "BFLBM_PAD=0" "BFLBM_PAD=528" "BFLBM_PAD=65552" "BFLBM_AB_OFF=0" "BFLBM_AB_OFF=1048592" "BFLBM_MAP_SX=1" "BFLBM_MAP_SX=2" "BFLBM_MAP_SX=4"; do
      if [ "$cfg" = default ]; then v=$(run --size $s); else v=$(env $cfg python bench.py --size $s --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'])"); fi
      echo "rep $rep size $s $cfg -> $v" | tee -a $out/knobs.txt
    done
  done
done
