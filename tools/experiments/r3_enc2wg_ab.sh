#!/bin/bash
# A/B of the encoder layouts on one box: one workgroup of eight waves per CU (default) against two workgroups of four waves
# (SAEV_AMD_ENC_2WG=1).   tools/experiments/r3_enc2wg_ab.sh <tag>
TAG=${1:-r3_enc2wg}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$PWD
OPTS="--steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0"
for r in 0 1 0 1; do
  echo "== SAEV_AMD_ENC_2WG=$r" >> $OUT/${TAG}_ab.txt
  SAEV_AMD_ENC_2WG=$r python bench.py $OPTS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['cand_max'], d['mse_last'])" >> $OUT/${TAG}_ab.txt
done
