#!/bin/bash
# Round 4, GPU call 27: Philox rounds with one v_mad_u64_u32 per product (BFLBM_PHILOX_MUL64) against mul_hi + mul_lo, noisy kernel from the mixture.
out=gpurun_out/r4_call27; rm -rf $out; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
tools/ab_n.sh 3 "--noise --size 512 --steps 40 --warmup 5" default $B/libbflbm_mul64.so > $out/mul64_512.txt 2>&1; cat $out/mul64_512.txt
tools/ab_n.sh 3 "--noise --size 256 --steps 100 --warmup 5" default $B/libbflbm_mul64.so > $out/mul64_256.txt 2>&1; cat $out/mul64_256.txt
