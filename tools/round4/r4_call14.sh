#!/bin/bash
# Round 4, GPU call 14: the quiet kernel's request spacing again, now that the placement tuning has removed the level scatter.
out=gpurun_out/r4_call14; rm -rf $out; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
tools/ab_n.sh 3 "--size 512 --steps 20 --warmup 5" default $B/libbflbm_q20.so $B/libbflbm_q24.so $B/libbflbm_q32.so $B/libbflbm_q36.so $B/libbflbm_q44.so > $out/quiet_spacing_512.txt 2>&1; cat $out/quiet_spacing_512.txt
tools/ab_n.sh 3 "--size 256 --steps 50 --warmup 5" default $B/libbflbm_q20.so $B/libbflbm_q24.so $B/libbflbm_q32.so $B/libbflbm_q36.so $B/libbflbm_q44.so > $out/quiet_spacing_256.txt 2>&1; cat $out/quiet_spacing_256.txt
timeout -k 10 300 python -m pytest tests/test_gpu_slabs.py -q -m gpu --durations=5 2>&1 | tail -12
