#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_droplet.py tests/test_gpu_parity.py -q -m gpu 2>&1 | tail -3
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
