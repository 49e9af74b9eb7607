#!/bin/bash
# the round's closing evidence on ONE box: GPU suite, bench line + early kernel stats + PMC passes, steady kernel stats + queue gaps,
# stall counters, per-step table of the steady loop
export PYTHONPATH=$PWD
mkdir -p gpurun_out
TAG=${1:-r04f}
( time timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1 ) 2>&1 | grep real
grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -2
timeout 900 bash tools/collect_profiles.sh $TAG > /dev/null 2>&1
timeout 600 bash tools/experiments/r4_steady_profile.sh $TAG > /dev/null 2>&1
python tools/rocpd_per_step.py "$(find /tmp/prof_st -name '*.db' | head -1)" --steps 100 > gpurun_out/${TAG}_per_step_steady.txt 2>&1
timeout 600 bash tools/collect_stalls.sh $TAG > /dev/null 2>&1
head -c 600 gpurun_out/${TAG}_bench_line.json; echo; head -12 gpurun_out/${TAG}_per_step_steady.txt; head -5 gpurun_out/${TAG}_gaps_steady.txt
