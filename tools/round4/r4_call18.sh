#!/bin/bash
# Round 4, GPU call 18: 256^3 process by process with 8 placement candidates; final profile sessions; the whole GPU suite.
out=gpurun_out/r4_call18; rm -rf $out; mkdir -p $out
for rep in 1 2 3 4 5 6; do timeout -k 10 200 python bench.py --size 256 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['placement'])" | tee -a $out/p256.txt; done
for rep in 1 2 3; do timeout -k 10 200 python bench.py --size 256 --noise --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('noise', d['value'], d['config']['placement'])" | tee -a $out/p256.txt; done
tools/profile_round.sh r04_512_handover --size 512 > gpurun_out/r04_512.log 2>&1; tail -2 gpurun_out/r04_512.log | cut -c1-300
tools/profile_round.sh r04_256_handover --size 256 > gpurun_out/r04_256.log 2>&1; tail -2 gpurun_out/r04_256.log | cut -c1-300
tools/profile_round.sh r04_512_noise --size 512 --noise > gpurun_out/r04_512n.log 2>&1; tail -2 gpurun_out/r04_512n.log | cut -c1-300
tools/profile_round.sh r04_256_noise --size 256 --noise > gpurun_out/r04_256n.log 2>&1; tail -2 gpurun_out/r04_256n.log | cut -c1-300
for t in r04_512_handover r04_256_handover r04_512_noise r04_256_noise; do rm -rf gpurun_out/$t/trace gpurun_out/$t/fetch gpurun_out/$t/write gpurun_out/$t/sq gpurun_out/$t/grbm; done
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/r4_final_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4_final_pytest.log; tail -4 gpurun_out/r4_final_pytest.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; cut -c1-400 gpurun_out/r4_bench_default.json
