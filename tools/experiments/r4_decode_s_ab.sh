#!/bin/bash
# slice decode (SAEV_AMD_DW=slices_s) against the whole-row decode (default): tests, kernel times, step
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_dw_slices.py tests/test_gpu_register_layout.py -x -q -m gpu -k "f16r" 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in slices slices_s slices slices_s; do
  SAEV_AMD_DW=$r python bench.py --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$r', 'steady %.4f ms' % d['ms_per_step'], 'early %.4f' % d['from_random_init']['ms_per_step'], 'enc %.4f' % d['roofline']['kernel_ms'])"
done
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_ds; rocprofv3 --kernel-trace -d /tmp/prof_ds -o run -- python bench.py --steps 60 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-auxk-probe --no-other-configs > /tmp/prof_ds.log 2>&1
python tools/rocpd_stats.py "$(find /tmp/prof_ds -name '*.db' | head -1)" --last 60 | grep -E "decode|normalize|dw_slices|csc_place" | cut -c1-60,75-130
