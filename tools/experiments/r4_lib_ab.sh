#!/bin/bash
# step time (steady + early) of several builds on ONE box, alternating: tools/experiments/r4_lib_ab.sh <rounds> lib1.so lib2.so ...
export PYTHONPATH=$PWD
R=$1; shift
for i in $(seq $R); do
  for L in "$@"; do
    SAEV_AMD_LIB=$L python bench.py --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$L', 'steady %.4f ms' % d['ms_per_step'], 'early %.4f' % d['from_random_init']['ms_per_step'], 'enc %.4f' % d['roofline']['kernel_ms'])"
  done
done
