#!/bin/bash
# Round 4, GPU call 10: placement tuning at creation (best of 3 candidate allocations) against none, process by process.
out=gpurun_out/r4_call10; rm -rf $out; mkdir -p $out
run() { timeout -k 10 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('placement'))"; }
for rep in 1 2 3 4 5 6; do
  for s in 256 512; do
    a=$(BFLBM_PLACEMENT_CANDIDATES=1 run --size $s); b=$(run --size $s)
    echo "rep $rep size $s  no tuning: $a   best of 3: $b" | tee -a $out/placement_ab.txt
  done
done
for rep in 1 2 3; do
  for s in 256 512; do
    a=$(BFLBM_PLACEMENT_CANDIDATES=1 run --size $s --noise); b=$(run --size $s --noise)
    echo "rep $rep noise size $s  no tuning: $a   best of 3: $b" | tee -a $out/placement_ab.txt
  done
  a=$(BFLBM_PLACEMENT_CANDIDATES=1 run --shape 1024,1024,64); b=$(run --shape 1024,1024,64)
  echo "rep $rep 1024x1024x64  no tuning: $a   best of 3: $b" | tee -a $out/placement_ab.txt
  a=$(BFLBM_PLACEMENT_CANDIDATES=1 run --size 384); b=$(run --size 384)
  echo "rep $rep size 384  no tuning: $a   best of 3: $b" | tee -a $out/placement_ab.txt
done
timeout -k 10 200 python tools/level_probe.py 256 6 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $out/level_probe_tuned.txt
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log; tail -4 $out/pytest.log
