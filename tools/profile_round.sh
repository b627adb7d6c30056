#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + stats of the bench (3 blocks of 20 steps after 5 warm-up steps: the first,
# pulled-ring step of a run is in the warm-up but still in the trace), then the PMC passes, then the bench line itself --
# ONE session on ONE box, so that the profile's average launch time and the bench line's ms_per_step can be compared.
# usage: tools/profile_round.sh <tag> [bench args]
set -e
tag=$1; shift
out=gpurun_out/$tag; rm -rf $out; mkdir -p $out     # one session per tag: the summary must describe this run only
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline "$@" > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/write -- python3 bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline "$@" > $out/write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $out/sq -- python3 bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline "$@" > $out/sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm -- python3 bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline "$@" > $out/grbm.log 2>&1
python3 bench.py --steps 20 --warmup 5 "$@" > $out/bench.json 2> $out/bench.err     # the driver's own invocation shape (K = 20, median of 3 blocks)
cat $out/trace/*/*kernel_stats.csv > $out/kernel_stats.csv
case " $* " in *" --noise "*) kn="k_fused_ho<4, 1";; *) kn="k_fused_ho<4, 0";; esac
python3 tools/trace_timed_avg.py $(ls $out/trace/*/*kernel_trace.csv | head -1) 20 3 "$kn" > $out/kernel_timed.json; cat $out/kernel_timed.json
python3 tools/pmc_summary.py $out > $out/pmc_summary.txt
tail -1 $out/bench.json
