#!/bin/bash
# Round 4, GPU call 33: the final kernel (non-temporal population stores): profile sessions, the whole GPU suite, the default bench lines.
tools/profile_round.sh r04_512_handover --size 512 > gpurun_out/r04_512.log 2>&1; tail -2 gpurun_out/r04_512.log | cut -c1-260
tools/profile_round.sh r04_256_handover --size 256 > gpurun_out/r04_256.log 2>&1; tail -2 gpurun_out/r04_256.log | cut -c1-260
tools/profile_round.sh r04_512_noise --size 512 --noise > gpurun_out/r04_512n.log 2>&1; tail -2 gpurun_out/r04_512n.log | cut -c1-260
tools/profile_round.sh r04_256_noise --size 256 --noise > gpurun_out/r04_256n.log 2>&1; tail -2 gpurun_out/r04_256n.log | cut -c1-260
for t in r04_512_handover r04_256_handover r04_512_noise r04_256_noise; do rm -rf gpurun_out/$t/trace gpurun_out/$t/fetch gpurun_out/$t/write gpurun_out/$t/sq gpurun_out/$t/grbm; done
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r4_final_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_final_pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; cut -c1-260 gpurun_out/r4_bench_default.json
