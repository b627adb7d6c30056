#!/bin/bash
# Round 4, GPU call 21: the noise profile sessions again, now from the mixture (configs[2]); sustained check.
tools/profile_round.sh r04_512_noise --size 512 --noise > gpurun_out/r04_512n.log 2>&1; tail -2 gpurun_out/r04_512n.log | cut -c1-300
tools/profile_round.sh r04_256_noise --size 256 --noise > gpurun_out/r04_256n.log 2>&1; tail -2 gpurun_out/r04_256n.log | cut -c1-300
for t in r04_512_noise r04_256_noise; do rm -rf gpurun_out/$t/trace gpurun_out/$t/fetch gpurun_out/$t/write gpurun_out/$t/sq gpurun_out/$t/grbm; done
for args in "--size 512 --noise --steps 100 --warmup 10" "--size 256 --noise --steps 400 --warmup 10"; do
  timeout -k 10 400 python bench.py $args --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$args', d['value'], d['config']['workload'], d['spread']['blocks_ms_per_step'])"
done
