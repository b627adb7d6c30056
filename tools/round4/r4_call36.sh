#!/bin/bash
out=gpurun_out/r4_call36; rm -rf $out; mkdir -p $out
timeout -k 10 500 python tools/spinodal_check.py 2>&1 | grep -v amdgpu.ids | tee $out/spinodal_check.txt
