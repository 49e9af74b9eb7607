#!/bin/bash
# Round 4, first measurement: where does the non-kernel time of a post-threshold step go?
#   (1) the bench loop without a profiler, sustained segments of 400 and of 2000 steps (does the window matter?)
#   (2) the same loop under rocprofv3 --kernel-trace: per-step span vs kernel-time sum, idle time by following kernel
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04a}
for n in 400 2000; do
  python bench.py --no-cpu-baseline --no-other-configs --no-auxk-probe --sustained-after 600 --sustained-steps $n 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['sustained']
print(f\"sustained {s['steps']} steps from {s['first_step']}: {s['ms_per_step']:.4f} ms/step, early {d['ms_per_step']:.4f}, enc {d['roofline']['kernel_ms']:.4f}, n_dead_last {s['n_dead_last']}, route {s['aux_route_last']}\")" >> gpurun_out/${TAG}_windows.txt
done
rm -rf /tmp/prof_gap
rocprofv3 --kernel-trace -d /tmp/prof_gap -o run -- python bench.py --steps 20 --warmup 5 --sustained-after 600 --sustained-steps 400 --no-cpu-baseline --no-auxk-probe --no-other-configs > /tmp/prof_gap.log 2>&1
DB=$(find /tmp/prof_gap -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" --last 200 > gpurun_out/${TAG}_kernel_stats_sustained.txt
python tools/rocpd_gaps.py "$DB" --last 200 > gpurun_out/${TAG}_gaps_sustained.txt
python tools/rocpd_gaps.py "$DB" --last 15 --skip-last 1005 > gpurun_out/${TAG}_gaps_early.txt
tail -1 /tmp/prof_gap.log > gpurun_out/${TAG}_sustained_bench_line.json
cat gpurun_out/${TAG}_windows.txt gpurun_out/${TAG}_gaps_sustained.txt gpurun_out/${TAG}_gaps_early.txt
