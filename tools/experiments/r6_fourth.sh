#!/bin/bash
# round 6, fourth GPU call: group streaming (tests + A/B against the round-5 group path), train() after the non-blocking permutation upload
export PYTHONPATH=$PWD
mkdir -p gpurun_out
T=r6d
timeout 1200 python -m pytest tests/test_gpu_api.py tests/test_gpu_stream.py tests/test_gpu_ddp.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -25 > gpurun_out/${T}_tests.txt
tail -6 gpurun_out/${T}_tests.txt
for route in 0 1 0 1; do
SAEV_AMD_GROUP=$route python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6d_group_ab.txt
import sys, json, os, torch
sys.path.insert(0, ".")
import bench
r = bench.sweep_group_record(torch.device("cuda:0"), n_saes=4, steps=40, warmup=10)
print("group_route", os.environ["SAEV_AMD_GROUP"], {k: round(r[k], 4) for k in ("group_ms_per_batch", "ms_per_sae", "single_ms_per_step", "per_sae_over_single")})
PY
done
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6d_e2e.txt
import sys, json, torch
sys.path.insert(0, ".")
import bench
for rep in range(2):
    r = bench.train_e2e_record(torch.device("cuda:0"))
    print(json.dumps({k: r[k] for k in ("train_ms_per_step", "engine_loop_ms_per_step", "train_over_engine_loop")}))
PY
