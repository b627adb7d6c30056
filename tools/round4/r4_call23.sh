#!/bin/bash
# Round 4, GPU call 23: the launcher against a REAL RCCL failure: two ranks on the one GPU of this box with the nccl backend (RCCL refuses
# several ranks per device), no injected failure.  Expect: rccl fails or hangs -> fresh workers with the native ring complete.
out=gpurun_out/r4_call23; rm -rf $out; mkdir -p $out
( time timeout -k 10 700 python bench.py --gpus 2 --shape 256,256,64 --steps 5 --warmup 2 --attempt-timeout 150 --no-second-transport > $out/line.json 2> $out/stderr.txt ) 2> $out/time.txt
echo "rc=$?"; cat $out/time.txt | tail -4; cut -c1-1200 $out/line.json; grep -E "^\[bench\]|Error|error|NCCL|RCCL" $out/stderr.txt | head -20
