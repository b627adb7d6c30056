#!/bin/bash
# Round 4, final check as the driver does it: build, smoke, GPU suite, default bench line.
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_final_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4_final_pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_default.json 2> gpurun_out/r4_bench_default.err; cut -c1-300 gpurun_out/r4_bench_default.json
python bench.py --noise --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('noise default', d['value'], d['config']['workload'], {k: v['value'] for k, v in d['config']['also'].items()})"
