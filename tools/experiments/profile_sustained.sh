#!/bin/bash
# per-kernel times of the steady-state tail of the bench loop (last 200 dispatches of every kernel) next to the first steps
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=$PWD
rm -rf /tmp/prof_sus
rocprofv3 --kernel-trace -d /tmp/prof_sus -o run -- python bench.py --steps 20 --warmup 5 --sustained-after 600 --sustained-steps 400 --no-cpu-baseline --no-auxk-probe --no-other-configs > /tmp/prof_sus.log 2>&1
DB=$(find /tmp/prof_sus -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" --last 200 > gpurun_out/${1:-r02}_kernel_stats_sustained.txt
tail -1 /tmp/prof_sus.log > gpurun_out/${1:-r02}_sustained_bench_line.json
