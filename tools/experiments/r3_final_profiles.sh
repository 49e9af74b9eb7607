#!/bin/bash
# The profile set of the shipped build, one gpurun call: kernel stats (early), PMC traffic, stall counters, sustained kernel stats,
# three bench runs.  Outputs under gpurun_out/r03_*; copy to profiles/.
export PYTHONPATH=$PWD
tools/collect_profiles.sh r03 > /dev/null 2>&1
tools/collect_stalls.sh r03 > /dev/null 2>&1
tools/experiments/profile_sustained.sh r03 > /dev/null 2>&1
{
  echo "python bench.py --no-cpu-baseline --no-other-configs, three runs on the box of profiles/r03_* (same build, same gpurun call)."
  echo " ms/step     acts/s  enc ms   frac  sustained ms readbacks  auxk1000 ms readbacks"
  for i in 1 2 3; do
    python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['sustained']; a=d['auxk_active'][1]
print(f\"{d['ms_per_step']:8.3f} {d['value']:10.0f} {r['kernel_ms']:7.3f} {r['frac']:6.3f} {s['ms_per_step']:13.3f} {s['n_dead_readbacks_in_segment']:9d} {a['ms_per_step']:12.3f} {a['n_dead_readbacks']:9d}\")"
  done
} > gpurun_out/r03_bench_runs.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_line.json
