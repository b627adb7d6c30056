#!/bin/bash
# Round 4, GPU call 5: the fitted chunk cost model against round 3's on a range of lattices; fine chunk scan at 512^3; GPU tests touched since.
out=gpurun_out/r4_call5; rm -rf $out; mkdir -p $out
run() { timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['schedule'])"; }
for rep in 1 2; do
for shape in 128,128,128 192,192,192 256,256,256 320,320,320 384,384,384 448,448,448 512,512,512 250,250,250 300,300,300 500,500,500 1024,1024,64 512,512,128 64,64,256 640,640,320; do
  a=$(BFLBM_PLAN_MODEL=3 run --shape $shape); b=$(BFLBM_PLAN_MODEL=4 run --shape $shape)
  echo "rep $rep $shape  round-3 model: $a   fitted model: $b" | tee -a $out/plan_ab.txt
done
done
for shape in 256,256,256 512,512,512; do
  a=$(BFLBM_PLAN_MODEL=3 run --shape $shape --noise); b=$(BFLBM_PLAN_MODEL=4 run --shape $shape --noise)
  echo "noise $shape  round-3 model: $a   fitted model: $b" | tee -a $out/plan_ab.txt
done
for wg in 2048 3072 4096 5120 6144; do
  v=$(BFLBM_FUSED_WG=$wg run --size 512); echo "size 512 BFLBM_FUSED_WG=$wg -> $v" | tee -a $out/chunk_scan_512.txt
done
timeout -k 10 600 python -m pytest tests/test_gpu_droplet.py tests/test_gpu_cpp_adapter.py tests/test_gpu_slabs.py tests/test_gpu_configs.py -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $out/pytest.log
tail -5 $out/pytest.log
