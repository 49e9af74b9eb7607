#!/bin/bash
# per-kernel times of bench.py's auxk_active probe (n_dead = 8 and 1000 forced): everything after the headline loop
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=$PWD
rm -rf /tmp/prof_aux
rocprofv3 --kernel-trace -d /tmp/prof_aux -o run -- python bench.py --steps 2 --warmup 1 --sustained-steps 0 --no-cpu-baseline --no-other-configs > /tmp/prof_aux.log 2>&1
DB=$(find /tmp/prof_aux -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" --last 30 > gpurun_out/${1:-r3}_kernel_stats_aux1000.txt
