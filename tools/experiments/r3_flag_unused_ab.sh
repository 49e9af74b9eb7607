#!/bin/bash
# A/B on one box: unused latents flagged for the fused Adam (default) against their dW_enc^T rows zeroed and both rows read (SAEV_AMD_FLAG_UNUSED=0)
TAG=${1:-r3_flag}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$PWD
OPTS="--steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --sustained-steps 0"
for r in 0 1 0 1; do
  echo "== SAEV_AMD_FLAG_UNUSED=$r" >> $OUT/${TAG}_ab.txt
  SAEV_AMD_FLAG_UNUSED=$r python bench.py $OPTS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], [o['ms_per_step'] for o in d['other_configs']])" >> $OUT/${TAG}_ab.txt
done
