#!/bin/bash
# A/B of schedule 3 (hand-over, ragged variant where the lattice is not whole tiles) against schedule 1 (fused, pulled
# ring) on lattices that round 2's hand-over refused: one process per measurement, interleaved, median block of 3.
# usage (GPU box): tools/ragged_ab.sh > gpurun_out/ragged_ab.log
run() { python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%-22s %-9s %8.1f MLUPS frac %.3f  %s' % (d['config']['workload'].split(' ')[0], d['config']['schedule'], d['value'], d['roofline']['frac'], d['spread']['blocks_ms_per_step']))"; }
for shape in ${SHAPES:-64,64,64 96,96,96 250,250,250 300,300,300 320,320,320 200,200,200 448,448,448 64,256,64 64,64,512 500,500,500}; do
  for sch in handover fused handover fused; do run --shape $shape --schedule $sch; done
done
for shape in ${NOISE_SHAPES:-64,64,64 250,250,250 300,300,300}; do
  for sch in handover two_pass; do run --shape $shape --schedule $sch --noise; done
done
