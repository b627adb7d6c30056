#!/bin/bash
# Round 4, GPU call 16: tests touched since the last full run.
timeout -k 10 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_slabs.py tests/test_gpu_api.py "tests/test_gpu_handover_oracle.py::test_auto_stays_exact_when_asked_or_out_of_range" tests/test_gpu_cpp_adapter.py -q -m gpu --durations=6 > gpurun_out/r4_call16_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4_call16_pytest.log; tail -16 gpurun_out/r4_call16_pytest.log
