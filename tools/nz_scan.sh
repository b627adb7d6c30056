#!/bin/bash
# one box: cost per site against the number of planes (256 x 256 x nz) and against the component-stride pad at 256^3
run() { python bench.py --shape $1 --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-14s %-22s %8.1f MLUPS %8.4f ms' % ('$1', '$2', d['value'], d['ms_per_step']))"; }
for nz in 128 192 224 240 248 256 264 272 288 320 384; do run 256,256,$nz "-"; done
for pad in 0 1040 2064 16400 65552 262160 1048592; do BFLBM_PAD=$pad run 256,256,256 "PAD=$pad"; done
for off in 33280 66576 1048592 4194320; do BFLBM_AB_OFF=$off run 256,256,256 "AB_OFF=$off"; done
