#!/bin/bash
# Round 4, GPU call 4: strict-exception yardstick tests, per-size counter table, chunk-count scans, noise-kernel request spacing, stamps at 256^3.
out=gpurun_out/r4_call4; rm -rf $out; mkdir -p $out
B=binary-fluctuating-lattice-boltzmann_amd/csrc/build
timeout -k 10 900 python -m pytest tests/test_gpu_handover_oracle.py tests/test_gpu_cpp_adapter.py tests/test_gpu_api.py -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -8 $out/pytest.log
tools/round4/r4_pmc_sizes.sh > $out/pmc_sizes.log 2>&1; cp gpurun_out/r4_pmc_sizes/per_site_table.txt $out/; cat $out/per_site_table.txt
# chunk counts: BFLBM_FUSED_WG = tile columns x chunks.  512^3: 1024 columns; 256^3: 256 columns; 448^3: 784; 384^3: 576
for cfg in "512 2048 4096 8192 16384" "256 256 512 1024 2048" "448 1568 3136 5488 6272 10976" "384 1152 2304 4608"; do
  set -- $cfg; s=$1; shift
  for wg in "$@"; do
    v=$(BFLBM_FUSED_WG=$wg timeout -k 10 200 python bench.py --size $s --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "size $s BFLBM_FUSED_WG=$wg -> $v" | tee -a $out/chunk_scan.txt
  done
  v=$(timeout -k 10 200 python bench.py --size $s --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "size $s planner default -> $v" | tee -a $out/chunk_scan.txt
done
tools/ab_n.sh 3 "--noise --size 512 --steps 20 --warmup 5" default $B/libbflbm_s44.so $B/libbflbm_s56.so $B/libbflbm_s68.so > $out/noise_ab_512.txt 2>&1; cat $out/noise_ab_512.txt
tools/ab_n.sh 3 "--noise --size 256 --steps 50 --warmup 5" default $B/libbflbm_s44.so $B/libbflbm_s56.so $B/libbflbm_s68.so > $out/noise_ab_256.txt 2>&1; cat $out/noise_ab_256.txt
for s in 256 512; do BFLBM_LIB=$B/libbflbm_stamp.so timeout -k 10 300 python tools/ho_stamps.py --size $s 2>&1 | grep -v amdgpu.ids | tee -a $out/stamps_quiet.txt; done
