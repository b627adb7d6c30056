#!/bin/bash
# Round 4, GPU call 24: what does the ragged-tile kernel cost on full tiles (BFLBM_FORCE_RAG=1), and where do ragged lattices stand with the placement tuning?
out=gpurun_out/r4_call24; rm -rf $out; mkdir -p $out
run() { timeout -k 10 300 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['schedule'])"; }
for rep in 1 2; do
  for s in 256 320 512; do
    a=$(run --size $s); b=$(BFLBM_FORCE_RAG=1 run --size $s)
    echo "rep $rep size $s full-tile kernel: $a   ragged kernel forced: $b" | tee -a $out/rag.txt
  done
  for s in 250 300 500; do
    a=$(run --size $s); b=$(run --size $s --schedule fused)
    echo "rep $rep size $s auto: $a   fused (schedule 1): $b" | tee -a $out/rag.txt
  done
done
