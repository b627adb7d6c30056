#!/bin/bash
# FETCH/WRITE/L2-hit counters for the calibration kernels and both step schedules.
set -e
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/calibrate.py ${2:-256} > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/write -- python3 tools/calibrate.py ${2:-256} > $out/write.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $out/ea -- python3 tools/calibrate.py ${2:-256} > $out/ea.log 2>&1 || true
