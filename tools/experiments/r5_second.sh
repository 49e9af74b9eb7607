#!/bin/bash
# round 5, second GPU call: the whole -m gpu suite (no -x), the vendor GEMM's kernel name
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r5b_gputests.txt
tail -12 gpurun_out/r5b_gputests.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_vg
rocprofv3 --kernel-trace --stats -d /tmp/prof_vg -o run --output-format csv -- python tools/vendor_gemm.py > /tmp/vg.log 2>&1
f=$(find /tmp/prof_vg -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-400 > gpurun_out/r5b_vendor_gemm_kernels.txt
cat gpurun_out/r5b_vendor_gemm_kernels.txt
