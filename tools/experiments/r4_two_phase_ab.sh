#!/bin/bash
# the fused encoder as one launch (SAEV_AMD_ENC_PHASES=1) against two (=2): parity tests on the two-launch form, then encoder and step
export PYTHONPATH=$PWD
SAEV_AMD_ENC_PHASES=2 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py -x -q -m gpu -k "f16r or bf16" 2>&1 | grep -E "passed|failed|Error" | tail -3
for i in 1 2; do
  for ph in 1 2; do
    echo -n "phases $ph: "; SAEV_AMD_ENC_PHASES=$ph python tools/time_encoder.py f16r 44 2>&1 | tail -2 | tr '\n' ' ' | sed 's/encoder ms:.*median/median/'; echo
  done
done
for i in 1 2; do
  for ph in 1 2; do
    SAEV_AMD_ENC_PHASES=$ph python bench.py --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('phases $ph', 'steady %.4f ms' % d['ms_per_step'], 'early %.4f' % d['from_random_init']['ms_per_step'], 'enc %.4f' % d['roofline']['kernel_ms'])"
  done
done
