#!/bin/bash
out=gpurun_out/r4_call9; rm -rf $out; mkdir -p $out
for s in 256 512; do
  for rep in 1 2 3; do timeout -k 10 300 python tools/level_probe.py $s $([ $s = 256 ] && echo 10 || echo 5) 2>&1 | grep -v amdgpu.ids | tee -a $out/level_probe.txt; echo "--- new process" | tee -a $out/level_probe.txt; done
done
