#!/bin/bash
# round 4, call 2: new known-answer tests, the whole gpu suite on the debug-cfg build, and the step time over a long run
export PYTHONPATH=$PWD
mkdir -p gpurun_out
python -m pytest tests/test_gpu_known_answers.py -x -q -m gpu > gpurun_out/r04_known.log 2>&1; echo "known rc=$?"
tail -15 gpurun_out/r04_known.log
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_all.log 2>&1; echo "all rc=$?"
tail -5 gpurun_out/r04_gpu_all.log
python tools/experiments/long_run_probe.py 512 40 > gpurun_out/r04_long_run.txt 2>&1
cat gpurun_out/r04_long_run.txt
