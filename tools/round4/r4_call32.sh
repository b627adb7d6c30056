#!/bin/bash
out=gpurun_out/r4_call32; rm -rf $out; mkdir -p $out
for shape in "512 512 512" "512 512 128" "1024 1024 64"; do timeout -k 10 300 python tools/slab_alone.py $shape 2>&1 | grep -v amdgpu.ids | tee -a $out/slab_alone.txt; done
