#!/bin/bash
# round 6, second GPU call: new tests (ownership checks, one-gather decode), A/B of the switches added this round, train() host profile
export PYTHONPATH=$PWD
mkdir -p gpurun_out
T=r6b
timeout 900 python -m pytest tests/test_gpu_stream.py tests/test_gpu_register_layout.py tests/test_gpu_configs.py tests/test_gpu_api.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -25 > gpurun_out/${T}_tests.txt
tail -6 gpurun_out/${T}_tests.txt
ab() {  # ab <label> <env...>: steady-state ms per step of the default bench loop under an environment
    local label=$1; shift
    env "$@" timeout 600 python bench.py --steps 60 --warmup 10 --sustained-steps 0 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras > /tmp/ab.log 2>&1
    python - "$label" <<'PY'
import json, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line:
    print(sys.argv[1], "FAILED", open("/tmp/ab.log").read()[-400:])
else:
    d = json.loads(line[-1])
    print(f"{sys.argv[1]:28s} ms {d['ms_per_step']:.4f}  early {d['from_random_init']['ms_per_step']:.4f}  enc {d['roofline']['kernel_ms']:.4f}")
PY
}
{
for rep in 1 2; do
ab "default"
ab "own_check off" SAEV_AMD_OWN_CHECK=1
ab "enc lock step" SAEV_AMD_ENC_ROT=1
ab "enc 512 wgs" SAEV_AMD_ENC_WGS=512
ab "enc 512 wgs lock step" SAEV_AMD_ENC_WGS=512 SAEV_AMD_ENC_ROT=1
done
} 2>&1 | tee gpurun_out/${T}_ab.txt
python - <<'PY' 2>&1 | tee gpurun_out/r6b_c3_decode_ab.txt
# configs[3] (bf16, d 1280, 81 920 latents, k 64): the one-gather decode against the two-half decode
import os, sys, time, torch
sys.path.insert(0, ".")
import bench
for route in ("0", "1", "0", "1"):
    os.environ["SAEV_AMD_DEC_ROUTE"] = route
    r = bench.other_config_record(torch.device("cuda:0"), name="c3", d=1280, s=81920, k=64, b=16384, encoder="bf16", steps=20, warmup=5)
    print("dec_route", route, "ms_per_step", round(r["ms_per_step"], 4), "enc", round(r["encoder_kernel_ms"], 4))
PY
timeout 600 python tools/train_host_profile.py --steps 300 > gpurun_out/${T}_train_host_profile.txt 2>&1; head -40 gpurun_out/${T}_train_host_profile.txt | cut -c1-200
