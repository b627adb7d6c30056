#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_handover_oracle.py -q -m gpu -k "spinodal or against_the_oracle" 2>&1 | tail -4
