#!/bin/bash
# Round 4: the same counter set at 256^3, 384^3, 448^3, 512^3 and 1024x1024x64 in ONE session on ONE box,
# quiet default schedule, one counter group per pass (never combined with tracing).  -> gpurun_out/r4_pmc_sizes/<size>/...
# usage: tools/round4/r4_pmc_sizes.sh
out=gpurun_out/r4_pmc_sizes; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in 256,256,256 384,384,384 448,448,448 512,512,512 1024,1024,64; do
  d=$out/$(echo $shape | tr ',' 'x'); mkdir -p $d
  A="--shape $shape --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $d/sq -- python3 bench.py $A > $d/sq.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -- python3 bench.py $A > $d/fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $d/write -- python3 bench.py $A > $d/write.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $d/grbm -- python3 bench.py $A > $d/grbm.log 2>&1
  python3 bench.py --shape $shape --steps 20 --warmup 5 --no-cpu-baseline > $d/bench.json 2> $d/bench.err
  python3 tools/pmc_summary.py $d > $d/pmc_summary.txt
  rm -rf $d/sq $d/fetch $d/write $d/grbm
  tail -1 $d/bench.json | cut -c1-200
done
python3 tools/pmc_per_site.py $out > $out/per_site_table.txt; cat $out/per_site_table.txt
