#!/bin/bash
# saev_train_step with a handful of dead latents: the one-pass AuxK forward on a stream of the context's own, next to the
# backward's pair-list build (saev_debug_cfg.aux_stream 0), against everything on the caller's stream (SAEV_AMD_AUX_STREAM=1)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
{
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -1
for i in 1 2 3; do
  for R in 1 0; do
    SAEV_AMD_AUX_STREAM=$R timeout 300 python bench.py --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 --steps 100 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('aux_stream $R', 'steady %.4f ms' % d['ms_per_step'], 'early %.4f' % d['from_random_init']['ms_per_step'], 'enc %.4f' % d['roofline']['kernel_ms'], 'mse %.6f' % d['mse_last'])"
  done
done
for nd in 1 3 8; do for R in 1 0; do echo -n "aux_stream $R "; SAEV_AMD_AUX_STREAM=$R timeout 120 python tools/experiments/r4_aux_nd.py $nd 60 2>/dev/null; done; done
} 2>&1 | tee gpurun_out/r04_aux_stream_ab.txt
