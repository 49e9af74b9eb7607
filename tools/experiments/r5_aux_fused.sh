#!/bin/bash
# the dense AuxK route after its fusions: tests that reach it, then per-step tables at forced dead counts
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py tests/test_gpu_configs.py -m gpu -q -x -k "aux or dead" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/${1}_aux_tests.txt
tail -12 gpurun_out/${1}_aux_tests.txt
bash tools/experiments/r5_aux_dense_steps.sh $1 1000 64 > /dev/null 2>&1
grep -v "^$" gpurun_out/${1}_aux_steps.txt | head -60
