#!/bin/bash
# bound-refresh cadence of the fused encoder epilogue with the measured margin (lists are shorter than when r3_cadence_sweep.sh ran):
# encoder kernel ms, steady step ms, early step ms, longest list; two passes over the grid
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for rep in 1 2; do for first in 4 8 16; do for every in 1 2 4; do
  r=$(SAEV_AMD_REFRESH_FIRST=$first SAEV_AMD_REFRESH_EVERY=$every timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4), round(d['from_random_init']['ms_per_step'],4), d['cand_max'])")
  echo "first=$first every=$every: $r"
done; done; done 2>&1 | tee gpurun_out/r04_cadence_sweep.txt
