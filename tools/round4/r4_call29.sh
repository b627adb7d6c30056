#!/bin/bash
# Round 4, GPU call 29: non-temporal population stores adopted; frame stores too (BFLBM_HO_NT_FRAMES=1)?  Parity of the changed kernel.
out=gpurun_out/r4_call29; rm -rf $out; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
tools/ab_n.sh 3 "--size 512 --steps 40 --warmup 5" default $B/libbflbm_ntf.so $B/libbflbm_nont.so > $out/nt_512.txt 2>&1; cat $out/nt_512.txt
tools/ab_n.sh 3 "--size 256 --steps 100 --warmup 5" default $B/libbflbm_ntf.so $B/libbflbm_nont.so > $out/nt_256.txt 2>&1; cat $out/nt_256.txt
timeout -k 10 600 python -m pytest tests/test_gpu_handover_oracle.py tests/test_gpu_golden.py tests/test_gpu_slabs.py -q -m gpu 2>&1 | tail -3
