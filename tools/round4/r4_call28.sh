#!/bin/bash
# Round 4, GPU call 28: population stores of the hand-over kernel with the non-temporal hint (BFLBM_HO_NT_STORES) against plain stores.
out=gpurun_out/r4_call28; rm -rf $out; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
tools/ab_n.sh 3 "--size 512 --steps 40 --warmup 5" default $B/libbflbm_nt.so > $out/nt_512.txt 2>&1; cat $out/nt_512.txt
tools/ab_n.sh 3 "--size 256 --steps 100 --warmup 5" default $B/libbflbm_nt.so > $out/nt_256.txt 2>&1; cat $out/nt_256.txt
tools/ab_n.sh 2 "--noise --size 512 --steps 40 --warmup 5" default $B/libbflbm_nt.so > $out/nt_512n.txt 2>&1; cat $out/nt_512n.txt
