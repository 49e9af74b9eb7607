#!/bin/bash
# A/B on one box: the decoder rows' finalize launches on a side stream under the dval sums and pass B (SAEV_AMD_DW_SIDE=1) or in line
TAG=${1:-r3_side}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$PWD
for r in 0 1 0 1; do
  echo "== SAEV_AMD_DW_SIDE=$r" >> $OUT/${TAG}_ab.txt
  SAEV_AMD_DW_SIDE=$r python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --sustained-steps 300 --sustained-after 600 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['sustained']['ms_per_step'], d['mse_last'])" >> $OUT/${TAG}_ab.txt
done
