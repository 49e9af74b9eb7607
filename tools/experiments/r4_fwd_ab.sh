#!/bin/bash
# A/B of the forward gathers: exact refinement (and, later, the decode) from 32-column slices vs whole-row gathers, same box
export PYTHONPATH=$PWD
mkdir -p gpurun_out
TAG=${1:-r04_fwd}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['from_random_init']
print(f\"$1: steady {d['ms_per_step']:.3f} ms ({d['value']:.0f}/s) enc {d['roofline']['kernel_ms']:.3f}  early {e['ms_per_step']:.3f} ms  mse {d['mse_last']:.6f} n_dead {d['timed_region']['n_dead_last']}\")"; }
for i in 1 2; do
  SAEV_AMD_FWD=rows python bench.py --no-cpu-baseline --no-other-configs --no-auxk-probe --sustained-steps 0 --steps 100 2>/dev/null | tail -1 | line rows
  python bench.py --no-cpu-baseline --no-other-configs --no-auxk-probe --sustained-steps 0 --steps 100 2>/dev/null | tail -1 | line slices
done > gpurun_out/${TAG}_ab.txt
cat gpurun_out/${TAG}_ab.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_fwd
rocprofv3 --kernel-trace -d /tmp/prof_fwd -o run -- python bench.py --steps 100 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 > /tmp/prof_fwd.log 2>&1
DB=$(find /tmp/prof_fwd -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" --last 100 > gpurun_out/${TAG}_kernel_stats_steady.txt
head -32 gpurun_out/${TAG}_kernel_stats_steady.txt
