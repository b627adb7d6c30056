#!/bin/bash
out=gpurun_out/r4_call8; rm -rf $out; mkdir -p $out
for s in 256 384; do
  for mode in "" hold; do
    for rep in 1 2; do timeout -k 10 200 python tools/level_probe.py $s 8 $mode 2>&1 | grep -v amdgpu.ids | tee -a $out/level_probe.txt; echo "--- new process" | tee -a $out/level_probe.txt; done
  done
done
