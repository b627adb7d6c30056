#!/bin/bash
# Round 4, GPU call 31: non-temporal population stores in the two bit-exact schedules as well?
out=gpurun_out/r4_call31; rm -rf $out; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
for a in "--size 256 --schedule fused --steps 100" "--size 512 --schedule fused --steps 30" "--size 256 --schedule two_pass --steps 100" "--size 512 --schedule two_pass --noise --steps 20" "--size 64 --steps 2000" "--size 32 --steps 5000"; do
  echo "## $a" | tee -a $out/nt2.txt
  tools/ab_n.sh 3 "$a --warmup 5" default $B/libbflbm_nt2.so 2>&1 | tee -a $out/nt2.txt
done
