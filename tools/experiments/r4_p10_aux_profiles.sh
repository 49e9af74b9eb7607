#!/bin/bash
# per-kernel profiles behind review item 7: P = 10 Matryoshka vs P = 1, and AuxK with 1 000 dead latents
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04}
{
python tools/experiments/r4_matry.py 1
python tools/experiments/r4_matry.py 10
python tools/experiments/r3_aux1000.py
} 2>/dev/null | tee gpurun_out/${TAG}_p10_aux_times.txt
for what in "matry:tools/experiments/r4_matry.py 10" "p1:tools/experiments/r4_matry.py 1" "aux1000:tools/experiments/r3_aux1000.py"; do
  name=${what%%:*}; cmd=${what#*:}
  rm -rf /tmp/prof_$name
  rocprofv3 --kernel-trace -d /tmp/prof_$name -o run -- python $cmd > /tmp/prof_$name.log 2>&1
  python tools/rocpd_stats.py "$(find /tmp/prof_$name -name '*.db' | head -1)" --last 20 > gpurun_out/${TAG}_${name}_kernel_stats.txt
done
head -30 gpurun_out/${TAG}_matry_kernel_stats.txt; head -45 gpurun_out/${TAG}_aux1000_kernel_stats.txt
