#!/bin/bash
# configs[0] (B 4 096: 16 batch blocks x 24 latent tiles = 384 tile jobs on 256 CUs): does another split of the jobs help?
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for rep in 1 2; do for wgs in 0 192 384 128; do
SAEV_AMD_ENC_WGS=$wgs python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, torch
sys.path.insert(0, ".")
import bench
r = bench.other_config_record(torch.device("cuda:0"), name="c0", d=768, s=6144, k=32, b=4096, encoder=None, steps=200, warmup=100)
print("enc_wgs", os.environ["SAEV_AMD_ENC_WGS"], "ms_per_step", round(r["ms_per_step"], 4), "enc_ms", round(r["encoder_kernel_ms"], 4), "dense", r["dense_route"], "cand_max", r["cand_max"])
PY
done; done | tee gpurun_out/r6k_c0_wgs.txt
