#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite on the round's first build, the vendor GEMM anchor, a steady-state profile
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r5a_gputests.txt
tail -5 gpurun_out/r5a_gputests.txt
python tools/vendor_gemm.py > gpurun_out/r5a_vendor_gemm.json 2> gpurun_out/r5a_vendor_gemm.err; cat gpurun_out/r5a_vendor_gemm.json
bash tools/experiments/r4_steady_profile.sh r5a > /dev/null 2>&1
python tools/rocpd_per_step.py "$(find /tmp/prof_st -name '*.db' | head -1)" --steps 100 > gpurun_out/r5a_per_step_steady.txt 2>&1
head -45 gpurun_out/r5a_per_step_steady.txt
