#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + stats of the default bench, then the PMC passes.
# usage: tools/profile_round.sh <tag> [bench args]
set -e
tag=$1; shift
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out     # one session per tag: the summary must describe this run only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/sq -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/grbm.log 2>&1
python3 bench.py --steps 100 --warmup 10 "$@" > $out/bench.json 2> $out/bench.err
cat $out/trace/*/*kernel_stats.csv > $out/kernel_stats.csv
python3 tools/pmc_summary.py $out > $out/pmc_summary.txt
tail -1 $out/bench.json
