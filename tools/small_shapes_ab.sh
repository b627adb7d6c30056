for shp in 128,128,128 192,192,192 128,128,512 256,256,64 256,256,128 192,192,384 128,256,256; do
  for s in fused handover; do
    python bench.py --shape $shp --steps 60 --warmup 10 --no-cpu-baseline --schedule $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$shp', '$s', d['value'])"
  done
  for s in fused handover; do
    python bench.py --shape $shp --steps 60 --warmup 10 --no-cpu-baseline --schedule $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$shp', '$s', d['value'])"
  done
done
