#!/bin/bash
# the training decode sets the bits of the backward's pair-list build (saev_debug_cfg.csc_route 0) against the build's own fill
# pass (SAEV_AMD_CSC=1): the GPU suite on the default, then step time and kernel times of both on ONE box
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
{
( time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) 2>&1 | grep -E "passed|failed|error|real"
for i in 1 2 3; do
  for R in 1 0; do
    SAEV_AMD_CSC=$R timeout 300 python bench.py --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('csc_route $R', 'steady %.4f ms' % d['ms_per_step'], 'early %.4f' % d['from_random_init']['ms_per_step'], 'enc %.4f' % d['roofline']['kernel_ms'])"
  done
done
for R in 1 0; do
  rm -rf /tmp/prof_ab
  SAEV_AMD_CSC=$R timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_ab -o run -- python bench.py --steps 60 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-auxk-probe --no-other-configs > /tmp/prof_ab.log 2>&1
  echo "== csc_route $R"
  python tools/rocpd_stats.py "$(find /tmp/prof_ab -name '*.db' | head -1)" --last 60 | grep -E "csc_|decode_q|dw_slices|aux_small_fused|dead_compact" | cut -c1-60,75-130
done
} 2>&1 | tee gpurun_out/r04_csc_prefill_ab.txt
