#!/bin/bash
# Run on the GPU box (via gpurun): HBM traffic of one bench configuration, one counter pass each.
# usage: tools/pmc_quick.sh <tag> [bench args]      (BFLBM_LIB etc. are taken from the environment)
set -e
tag=$1; shift
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/write -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/write.log 2>&1
python3 tools/pmc_summary.py $out > $out/pmc_summary.txt
cat $out/pmc_summary.txt
