#!/bin/bash
# round 5 work-horse: r5_run.sh <tag> [tests|notests] -- the -m gpu suite (optional), then a steady-state profile of the current build
export PYTHONPATH=$PWD
TAG=${1:-r5x}
mkdir -p gpurun_out
if [ "${2:-tests}" = "tests" ]; then
  timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -60 > gpurun_out/${TAG}_gputests.txt
  grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_gputests.txt | tail -30
fi
bash tools/experiments/r4_steady_profile.sh $TAG > /dev/null 2>&1
python tools/rocpd_per_step.py "$(find /tmp/prof_st -name '*.db' | head -1)" --steps 100 > gpurun_out/${TAG}_per_step_steady.txt 2>&1
head -48 gpurun_out/${TAG}_per_step_steady.txt
python -c "
import json,sys
d=json.loads(open('gpurun_out/${TAG}_steady_bench_line.json').read())
print({k:d[k] for k in ('value','ms_per_step','mse_last')}, d['from_random_init']['ms_per_step'], d.get('mse_rel_err_vs_oracle'))
"
