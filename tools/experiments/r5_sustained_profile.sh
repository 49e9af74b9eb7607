#!/bin/bash
# per-step kernel table of the LAST 300 steps of the sustained segment (the headline's loop continued: dead count drifting 0 ... 35)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=$PWD
mkdir -p gpurun_out
rm -rf /tmp/prof_sus
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_sus -o run -- python bench.py --steps 20 --warmup 5 --sustained-steps 1200 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras > /tmp/prof_sus.log 2>&1
grep '^{' /tmp/prof_sus.log | tail -1 > gpurun_out/${1}_sustained_bench_line.json
python tools/rocpd_per_step.py "$(find /tmp/prof_sus -name '*.db' | head -1)" --steps 300 > gpurun_out/${1}_per_step_sustained.txt 2>&1
head -40 gpurun_out/${1}_per_step_sustained.txt
python -c "
import json;b=json.load(open('gpurun_out/${1}_sustained_bench_line.json'));print(b['ms_per_step'], b['sustained'])"
