#!/bin/bash
# Round 4, GPU call 30: non-temporal LOADS of the populations that do not travel in x (1) / of all populations (2) in the hand-over kernel.
out=gpurun_out/r4_call30; rm -rf $out; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
tools/ab_n.sh 3 "--size 512 --steps 40 --warmup 5" default $B/libbflbm_ntl1.so $B/libbflbm_ntl2.so > $out/ntl_512.txt 2>&1; cat $out/ntl_512.txt
tools/ab_n.sh 3 "--size 256 --steps 100 --warmup 5" default $B/libbflbm_ntl1.so $B/libbflbm_ntl2.so > $out/ntl_256.txt 2>&1; cat $out/ntl_256.txt
