#!/bin/bash
# Round 4, GPU call 19: does the step time rise over a longer run (clock management)?  blocks of 100 and of 400 steps at 512^3, quiet and with noise.
out=gpurun_out/r4_call19; rm -rf $out; mkdir -p $out
for args in "--steps 100 --warmup 10" "--steps 400 --warmup 10 --blocks 2" "--noise --steps 100 --warmup 10" "--noise --steps 400 --warmup 10 --blocks 2"; do
  timeout -k 10 400 python bench.py --size 512 $args --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', d['value'], d['spread']['blocks_ms_per_step'], d['config']['placement'])" | tee -a $out/sustained.txt
done
rocm-smi --showclocks --showpower 2>/dev/null | head -30 | tee -a $out/sustained.txt
