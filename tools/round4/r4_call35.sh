#!/bin/bash
# Round 4, GPU call 35: the whole GPU suite with the placement tuning ON in every context it applies to (the suite switches it off by default).
BFLBM_PLACEMENT_CANDIDATES=4 timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/r4_suite_tuned.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r4_suite_tuned.log
