#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_slabs.py -q -m gpu -k "bench" 2>&1 | tail -4
