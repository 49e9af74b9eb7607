#!/bin/bash
# A/B on one box: CSC build in front of the backward (SAEV_AMD_CSC_EARLY=0) against on the side stream behind the select (default)
TAG=${1:-r3_csc}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$PWD
OPTS="--steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0"
for r in 0 1 0 1; do
  echo "== SAEV_AMD_CSC_EARLY=$r" >> $OUT/${TAG}_ab.txt
  SAEV_AMD_CSC_EARLY=$r python bench.py $OPTS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['mse_last'])" >> $OUT/${TAG}_ab.txt
done
