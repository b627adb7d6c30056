#!/bin/bash
# round-2 ablation: upper bound of what removing the ring pulls buys, per tile shape (results are wrong by construction)
A=build/ab
./build/probe/dpp_probe
for size in 256 512; do
echo "== size $size"
tools/ab_env.sh "--size $size --steps 60 --warmup 5 --schedule fused" "-" \
  "BFLBM_LIB=$A/abl1_64x8.so" \
  "BFLBM_LIB=$A/abl1_64x4.so BFLBM_FUSED_WG=$((size*size/256*2))" \
  "BFLBM_LIB=$A/abl1_64x4.so BFLBM_FUSED_WG=$((size*size/256*4))" \
  "BFLBM_LIB=$A/abl1_32x8.so BFLBM_FUSED_WG=$((size*size/256*2))" \
  "BFLBM_LIB=$A/abl1_128x4.so"
done
