#!/bin/bash
out=gpurun_out/r4_call15; rm -rf $out; mkdir -p $out
for s in 256 512; do for w in state frames both; do timeout -k 10 300 python tools/level_what.py $s $w 4 2>&1 | grep -v amdgpu.ids | tee -a $out/level_what.txt; done; done
