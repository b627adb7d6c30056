#!/bin/bash
# Round 4, GPU call 38: the size sweep of the round with the final code (one box, one process per lattice, bench.py median of 3 blocks).
out=gpurun_out/r4_call38; rm -rf $out; mkdir -p $out
bash tools/size_sweep.sh 2>&1 | tee $out/size_sweep.txt
python bench.py --size 500 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('500', d['value'], d['config']['schedule'], d['ms_per_step'])" | tee -a $out/size_sweep.txt
python bench.py --shape 8,256,64 --steps 2000 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('8x256x64', d['value'], d['config']['schedule'], d['ms_per_step'])" | tee -a $out/size_sweep.txt
