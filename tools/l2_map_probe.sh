#!/bin/bash
# does the order in which tile columns are handed to the XCDs change how many of the x-shifted /
# ring lines hit in L2?  TCC hit and miss counts and the step time of the 512^3 hand-over kernel for strips of 1, 2, 4
# tiles in x and for row-major order (default = 8 tiles = a whole tile row).
# usage (GPU box): tools/l2_map_probe.sh > gpurun_out/l2_map_probe.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sx in 1 2 4 8; do
  export BFLBM_MAP_SX=$sx
  out=gpurun_out/l2map_$sx; rm -rf $out; mkdir -p $out
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -- python3 bench.py --size 512 --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline > $out/tcc.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --size 512 --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline > $out/fetch.log 2>&1
  python3 tools/pmc_summary.py $out | grep -A4 "k_fused_ho" | sed "s/^/sx=$sx /"
  for rep in 1 2 3; do python3 bench.py --size 512 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('sx=$sx', d['value'], 'MLUPS', d['spread']['blocks_ms_per_step'])"; done
done
