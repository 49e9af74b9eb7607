#!/bin/bash
# decode-formed dval (default) against pass-A-formed dval (SAEV_AMD_DW=slices_a): tests, then step time early and steady
export PYTHONPATH=$PWD
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dw_slices.py tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_known_answers.py -x -q -m gpu > gpurun_out/r04_dval_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r04_dval_tests.log
for r in slices slices_a slices slices_a; do
  SAEV_AMD_DW=$r python bench.py --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$r', 'steady %.4f ms' % d['ms_per_step'], 'early %.4f' % d['from_random_init']['ms_per_step'], 'enc %.4f' % d['roofline']['kernel_ms'], d['timed_region'].get('n_dead_last'))"
done
