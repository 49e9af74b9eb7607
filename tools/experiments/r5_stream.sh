#!/bin/bash
# the streamed-step tests first (fast feedback), then the suite + profile
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_dw_slices.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -40 > gpurun_out/${1}_stream_tests.txt
tail -15 gpurun_out/${1}_stream_tests.txt
bash tools/experiments/r5_run.sh $1 ${2:-tests}
