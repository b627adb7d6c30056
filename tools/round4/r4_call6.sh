#!/bin/bash
# Round 4, GPU call 6: does the row pitch explain why lattices narrower than 512 run slower per wave?  BFLBM_PITCH = row pitch in doubles.
out=gpurun_out/r4_call6; rm -rf $out; mkdir -p $out
run() { timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  for cfg in "256 0" "256 320" "256 512" "384 0" "384 512" "448 0" "448 512" "320 0" "320 512" "512 0" "512 576" "300 0" "300 512"; do
    set -- $cfg
    if [ "$2" = 0 ]; then v=$(run --size $1); else v=$(BFLBM_PITCH=$2 run --size $1); fi
    echo "rep $rep size $1 pitch ${2/#0/default} -> $v" | tee -a $out/pitch_ab.txt
  done
done
timeout -k 10 400 python -m pytest tests/test_gpu_droplet.py -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -3 $out/pytest.log
timeout -k 10 600 python tools/ho_stress.py --seeds 1 2 --cases 10 > $out/ho_stress.log 2>&1; tail -3 $out/ho_stress.log
