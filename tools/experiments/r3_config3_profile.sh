#!/bin/bash
# per-kernel times of configs[3]'s shape (d=1280, 81 920 latents, k=64, bf16 encoder, 16 384 rows) on one GPU
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=$PWD
rm -rf /tmp/prof_c3
rocprofv3 --kernel-trace -d /tmp/prof_c3 -o run -- python -c "
import torch, bench
r = bench.other_config_record(torch.device('cuda', 0), name='configs[3]', d=1280, s=81920, k=64, b=16384, encoder='bf16', steps=20)
print(r)
" > /tmp/prof_c3.log 2>&1
python tools/rocpd_stats.py "$(find /tmp/prof_c3 -name '*.db' | head -1)" > gpurun_out/${1:-r03}_config3_kernel_stats.txt
tail -2 /tmp/prof_c3.log
