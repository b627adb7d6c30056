#!/bin/bash
# Round 4, GPU call 34: request spacing of the noisy kernel once more, on the mixture and with the non-temporal stores (default 44).
out=gpurun_out/r4_call34; rm -rf $out; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
tools/ab_n.sh 3 "--noise --size 512 --steps 40 --warmup 5" default $B/libbflbm_n36.so $B/libbflbm_n52.so $B/libbflbm_n60.so > $out/noise_spacing_512.txt 2>&1; cat $out/noise_spacing_512.txt
tools/ab_n.sh 3 "--noise --size 256 --steps 100 --warmup 5" default $B/libbflbm_n36.so $B/libbflbm_n52.so $B/libbflbm_n60.so > $out/noise_spacing_256.txt 2>&1; cat $out/noise_spacing_256.txt
