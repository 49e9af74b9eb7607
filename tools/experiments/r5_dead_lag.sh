#!/bin/bash
# sustained segment under shorter tracker-record lags (tighter bound of the dead count -> the dense route less often; less host run-ahead)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for round in 1 2; do
for lag in 4 2 1; do
  SAEV_AMD_DEAD_LAG=$lag timeout 600 python bench.py --steps 20 --warmup 5 --sustained-steps 1500 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras 2>/dev/null | grep '^{' | tail -1 | python -c "
import json,sys
b=json.loads(sys.stdin.read())
print('lag $lag: steady %.4f sustained %.4f ms  n_dead_last %d route_last %d readbacks %d' % (b['ms_per_step'], b['sustained']['ms_per_step'], b['sustained']['n_dead_last'], b['sustained']['aux_route_last'], b['sustained']['n_dead_readbacks_in_segment']))"
done; done | tee gpurun_out/${1:-r05b}_dead_lag.txt
