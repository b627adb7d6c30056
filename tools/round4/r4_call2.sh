#!/bin/bash
# Round 4, GPU call 2: the whole GPU suite with the strict-form collection on, then the ring overlap traces and slab overheads.
out=gpurun_out/r4_call2; rm -rf $out; mkdir -p $out
export BFLBM_STRICT_COLLECT=$PWD/$out/strict_collect.jsonl
timeout -k 10 1500 python -m pytest tests -q -m gpu -x > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -15 $out/pytest.log
unset BFLBM_STRICT_COLLECT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "512 512 128 4" "1024 1024 64 2"; do
  tag=$(echo $cfg | tr ' ' 'x')
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 tools/ring_trace.py $cfg 6 kernel > $out/trace_$tag.log 2>&1
  f=$(ls $out/trace_$tag/*/*kernel_trace.csv | head -1)
  python3 tools/ring_overlap_report.py $f $(echo $cfg | awk '{print $4}') > $out/overlap_$tag.txt 2>&1
  tail -5 $out/overlap_$tag.txt
  rm -rf $out/trace_$tag
done
timeout -k 10 300 python3 tools/ring_bench.py 512 512 128 20 1,4 > $out/ring_bench_512x512x128.json 2>$out/ring_bench.err; cat $out/ring_bench_512x512x128.json
timeout -k 10 300 python3 tools/ring_bench.py 1024 1024 64 20 1,2 > $out/ring_bench_1024x1024x64.json 2>>$out/ring_bench.err; cat $out/ring_bench_1024x1024x64.json
