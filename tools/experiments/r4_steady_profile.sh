#!/bin/bash
# steady-state kernel stats + queue gaps of the current build (100 timed steps after 1 500 training steps)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04e}
rm -rf /tmp/prof_st
rocprofv3 --kernel-trace -d /tmp/prof_st -o run -- python bench.py --steps 100 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras > /tmp/prof_st.log 2>&1
DB=$(find /tmp/prof_st -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" --last 100 > gpurun_out/${TAG}_kernel_stats_steady.txt
python tools/rocpd_gaps.py "$DB" --last 100 > gpurun_out/${TAG}_gaps_steady.txt
grep '^{' /tmp/prof_st.log | tail -1 > gpurun_out/${TAG}_steady_bench_line.json
head -32 gpurun_out/${TAG}_kernel_stats_steady.txt; head -8 gpurun_out/${TAG}_gaps_steady.txt
