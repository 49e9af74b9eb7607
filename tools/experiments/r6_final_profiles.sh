#!/bin/bash
# round 6 closing evidence on ONE box: GPU suite, bench line + PMC passes, steady / sustained per-step tables + queue gaps, stall
# counters, configs[3] / configs[0] per-step tables, DDP host timing, the 8-rank rehearsal, train() end to end from a shard directory
export PYTHONPATH=$PWD
mkdir -p gpurun_out
TAG=${1:-r06}
( time timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gpu_tests.log 2>&1 ) 2>&1 | grep real
grep -E "passed|failed|error" gpurun_out/${TAG}_gpu_tests.log | tail -2
timeout 900 bash tools/collect_profiles.sh $TAG > /dev/null 2>&1
timeout 600 bash tools/experiments/r4_steady_profile.sh $TAG > /dev/null 2>&1
python tools/rocpd_per_step.py "$(find /tmp/prof_st -name '*.db' | head -1)" --steps 100 > gpurun_out/${TAG}_per_step_steady.txt 2>&1
timeout 600 bash tools/collect_stalls.sh $TAG > /dev/null 2>&1
timeout 600 bash tools/experiments/r5_cfg_profiles.sh $TAG > gpurun_out/${TAG}_cfg_profiles.log 2>&1
timeout 400 python tools/ddp_host_timing.py 2> /dev/null | grep '^{' | tail -1 > gpurun_out/${TAG}_ddp_host_timing.json
rm -f gpurun_out/r5_eight_rank_rehearsal.txt; timeout 900 bash tools/experiments/r5_eight_rank_rehearsal.sh > /dev/null 2>&1; mv gpurun_out/r5_eight_rank_rehearsal.txt gpurun_out/${TAG}_eight_rank_rehearsal_one_gpu_gloo.txt
timeout 900 python tools/bench_train_e2e.py --gb 8 --root /dev/shm --epochs 3 --threads 8 > gpurun_out/${TAG}_train_e2e_tool.txt 2>&1
head -c 400 gpurun_out/${TAG}_bench_line.json; echo; head -30 gpurun_out/${TAG}_per_step_steady.txt; head -5 gpurun_out/${TAG}_gaps_steady.txt
tail -4 gpurun_out/${TAG}_train_e2e_tool.txt | cut -c1-200
