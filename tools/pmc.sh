#!/bin/bash
# usage: tools/pmc.sh <outdir-name> <bench args...>   (run on the GPU box via gpurun)
# Collects the PMC passes one at a time (counters never combined with tracing).
set -e
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py $BENCH_ARGS > $out/$name.log 2>&1; }
export BENCH_ARGS="$*"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
run sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run grbm GRBM_GUI_ACTIVE
echo done
