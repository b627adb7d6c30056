#!/bin/bash
out=gpurun_out/r4_call20; rm -rf $out; mkdir -p $out
timeout -k 10 600 python tools/sustain_probe.py 512 2>&1 | grep -v amdgpu.ids | tee $out/sustain_probe.txt
