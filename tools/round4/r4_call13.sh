#!/bin/bash
# Round 4, GPU call 13: whole GPU suite with durations; the stress tool's tally; the default bench line.
timeout -k 10 900 python -m pytest tests -q -m gpu --durations=25 > gpurun_out/r4_suite.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4_suite.log; tail -40 gpurun_out/r4_suite.log
timeout -k 10 500 python tools/ho_stress.py --seeds 1 2 3 --cases 12 > gpurun_out/r4_ho_stress.log 2>&1; tail -2 gpurun_out/r4_ho_stress.log
timeout -k 10 400 python tools/ho_stress.py --seeds 1 2 --cases 12 --ragged > gpurun_out/r4_ho_stress_ragged.log 2>&1; tail -2 gpurun_out/r4_ho_stress_ragged.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; cat gpurun_out/r4_bench_default.json | cut -c1-1500
