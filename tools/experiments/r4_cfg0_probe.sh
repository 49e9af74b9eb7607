#!/bin/bash
# configs[0] (D 768, S 6144, k 32, B 4096): per-kernel time, span vs kernel sum, idle time on the queue; then the whole gpu suite
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04d}
python tools/probe_shape.py --d-model 768 --d-sae 6144 --top-k 32 --batch 4096 --steps 300 --k-aux 512
rm -rf /tmp/prof_c0
rocprofv3 --kernel-trace -d /tmp/prof_c0 -o run -- python tools/probe_shape.py --d-model 768 --d-sae 6144 --top-k 32 --batch 4096 --steps 300 --k-aux 512 > /tmp/prof_c0.log 2>&1
tail -1 /tmp/prof_c0.log
DB=$(find /tmp/prof_c0 -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" --last 200 > gpurun_out/${TAG}_cfg0_kernel_stats.txt
python tools/rocpd_gaps.py "$DB" --last 200 > gpurun_out/${TAG}_cfg0_gaps.txt
head -45 gpurun_out/${TAG}_cfg0_kernel_stats.txt; head -30 gpurun_out/${TAG}_cfg0_gaps.txt
# (the gpu suite used to run here: tools/experiments/r4_third.sh)
