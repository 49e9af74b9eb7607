#!/bin/bash
# round 6, third GPU call: the sweep-group record and train() against the bare loop after the permutation prefetch
export PYTHONPATH=$PWD
mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6c_group_e2e.txt
import sys, json, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
for n in (2, 4):
    print(json.dumps(bench.sweep_group_record(dev, n_saes=n)))
for rep in range(2):
    r = bench.train_e2e_record(dev)
    print(json.dumps({k: r[k] for k in ("train_ms_per_step", "engine_loop_ms_per_step", "train_over_engine_loop")}))
PY
