#!/bin/bash
# Run on the GPU box: sample the clocks and the power of the device while bench.py runs the 512^3 case
( for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 1; done ) > gpurun_out/clock_probe.txt &
SAMPLER=$!
python bench.py --size 512 --steps 400 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['frac'])"
wait $SAMPLER
cat gpurun_out/clock_probe.txt
