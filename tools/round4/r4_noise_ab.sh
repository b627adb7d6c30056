#!/bin/bash
# Round 4, on the GPU box: parity of the new generator, then interleaved A/B of the noise-kernel variants.
# usage: tools/round4/r4_noise_ab.sh  (writes gpurun_out/r4_noise_ab/)
out=gpurun_out/r4_noise_ab; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
set -x
timeout -k 10 900 python -m pytest tests/test_gpu_noise.py tests/test_gpu_handover_oracle.py tests/test_gpu_golden.py tests/test_cabi.py "tests/test_gpu_configs.py::test_config2_noise_mode_variances_at_256_cubed" -x -q -m gpu > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -3 $out/pytest.log
tools/ab_n.sh 3 "--noise --size 512 --steps 20 --warmup 5" $B/libbflbm_old.so default $B/libbflbm_s16.so $B/libbflbm_s24.so $B/libbflbm_s32.so $B/libbflbm_s44.so $B/libbflbm_ph2.so > $out/ab_512.txt 2>&1
cat $out/ab_512.txt
tools/ab_n.sh 3 "--noise --size 256 --steps 50 --warmup 5" $B/libbflbm_old.so default $B/libbflbm_s16.so $B/libbflbm_s24.so $B/libbflbm_s32.so $B/libbflbm_s44.so $B/libbflbm_ph2.so > $out/ab_256.txt 2>&1
cat $out/ab_256.txt
BFLBM_LIB=$B/libbflbm_stamp.so timeout -k 10 300 python tools/ho_stamps.py --size 512 --noise > $out/stamps_noise_512.txt 2>&1
cat $out/stamps_noise_512.txt
BFLBM_LIB=$B/libbflbm_stamp.so timeout -k 10 300 python tools/ho_stamps.py --size 512 > $out/stamps_quiet_512.txt 2>&1
cat $out/stamps_quiet_512.txt
