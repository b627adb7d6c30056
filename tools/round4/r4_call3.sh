#!/bin/bash
# Round 4, GPU call 3: the whole GPU suite with the strict-form collection on, the grid-barrier probe, ring overlap traces.
out=gpurun_out/r4_call3; rm -rf $out; mkdir -p $out
export BFLBM_STRICT_COLLECT=$PWD/$out/strict_collect.jsonl
timeout -k 10 1000 python -m pytest tests -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -25 $out/pytest.log
unset BFLBM_STRICT_COLLECT
timeout -k 10 120 build/probe/grid_barrier_probe > $out/grid_barrier.txt 2>&1; cat $out/grid_barrier.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export GPU_MAX_HW_QUEUES=8
for cfg in "512 512 128 4" "1024 1024 64 2"; do
  tag=$(echo $cfg | tr ' ' 'x')
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 tools/ring_trace.py $cfg 6 kernel > $out/trace_$tag.log 2>&1
  f=$(ls $out/trace_$tag/*/*kernel_trace.csv | head -1)
  python3 tools/ring_overlap_report.py $f $(echo $cfg | awk '{print $4}') > $out/overlap_$tag.txt 2>&1
  tail -4 $out/overlap_$tag.txt
  rm -rf $out/trace_$tag
done
