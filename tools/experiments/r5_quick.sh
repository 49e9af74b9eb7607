#!/bin/bash
# quick check of the backward's kernels + profile: r5_quick.sh <tag>
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dw_slices.py tests/test_gpu_ddp.py tests/test_gpu_stream.py tests/test_gpu_known_answers.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -15 > gpurun_out/${1}_quick_tests.txt
tail -4 gpurun_out/${1}_quick_tests.txt
timeout 600 bash tools/experiments/r5_run.sh $1 notests
