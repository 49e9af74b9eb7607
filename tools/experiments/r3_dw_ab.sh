#!/bin/bash
# A/B of the weight-gradient routes on one box: bench.py's timed region with dw_rows (SAEV_AMD_DW=rows) and with the column
# slices (default), then the per-kernel times of the default build.   tools/experiments/r3_dw_ab.sh <tag>
TAG=${1:-r3_dw}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$PWD
OPTS="--steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0"
for r in rows slices rows slices; do
  echo "== $r" >> $OUT/${TAG}_ab.txt
  SAEV_AMD_DW=$r python bench.py $OPTS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])" >> $OUT/${TAG}_ab.txt
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace -d /tmp/prof_$TAG/kt -o run -- python bench.py $OPTS > /tmp/prof_$TAG.log 2>&1
python tools/rocpd_stats.py "$(find /tmp/prof_$TAG/kt -name '*.db' | head -1)" > $OUT/${TAG}_kernel_stats.txt
