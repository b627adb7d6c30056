#!/bin/bash
out=gpurun_out/r4_call26; rm -rf $out; mkdir -p $out
timeout -k 10 300 python tools/power_probe.py 2>&1 | grep -v amdgpu.ids | tee $out/power_probe.txt
