#!/bin/bash
# per-step kernel tables of configs[3] (bf16, 81 920 latents, k = 64) and configs[0] on the current build + the folded-sum / wide tests
export PYTHONPATH=$PWD
TAG=${1:-r5i}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_register_layout.py tests/test_gpu_stream.py -m gpu -q 2>&1 | grep -v "^$" | tail -30 > gpurun_out/${TAG}_tests.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_tests.txt | tail -12
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for cfg in "c3 1280 81920 64 16384 bf16 bf16_m16" "c0 768 6144 32 4096 None encode_m16"; do
  set -- $cfg
  rm -rf /tmp/prof_$1
  rocprofv3 --kernel-trace -d /tmp/prof_$1 -o run -- python -c "
import torch, bench
enc = None if '$6' == 'None' else '$6'
r = bench.other_config_record(torch.device('cuda', 0), name='$1', d=$2, s=$3, k=$4, b=$5, encoder=enc, steps=40, warmup=60)
print({k: r[k] for k in ('ms_per_step', 'encoder_kernel_ms', 'dense_route', 'cand_max')})
" > /tmp/prof_$1.log 2>&1
  grep "ms_per_step" /tmp/prof_$1.log
  python tools/rocpd_per_step.py "$(find /tmp/prof_$1 -name '*.db' | head -1)" --steps 30 --anchor encode_m16 > gpurun_out/${TAG}_$1_per_step.txt 2>&1
  head -36 gpurun_out/${TAG}_$1_per_step.txt
done
