#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dw_slices.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_known_answers.py tests/test_gpu_ddp.py -x -q -m gpu > gpurun_out/r04_matry_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r04_matry_tests.log
for r in slices slices_a slices slices_a; do echo -n "$r: "; SAEV_AMD_DW=$r python tools/experiments/r4_matry.py 10 2>/dev/null | tail -1; done
python tools/experiments/r4_matry.py 1 2>/dev/null | tail -1
