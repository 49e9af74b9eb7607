#!/bin/bash
# P = 10 Matryoshka against P = 1 on the current build: step times and the per-kernel table of the P = 10 step
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04f}
{ python tools/experiments/r4_matry.py 1; python tools/experiments/r4_matry.py 10; python tools/experiments/r4_matry.py 1; python tools/experiments/r4_matry.py 10; } 2>/dev/null | tee gpurun_out/${TAG}_p10_times.txt
rm -rf /tmp/prof_m
rocprofv3 --kernel-trace -d /tmp/prof_m -o run -- python tools/experiments/r4_matry.py 10 > /tmp/prof_m.log 2>&1
python tools/rocpd_stats.py "$(find /tmp/prof_m -name '*.db' | head -1)" --last 20 > gpurun_out/${TAG}_matry_kernel_stats.txt
head -24 gpurun_out/${TAG}_matry_kernel_stats.txt
