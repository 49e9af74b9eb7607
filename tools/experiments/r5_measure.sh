#!/bin/bash
# round 5: the measurements behind VERDICT r4 items 3, 4, 5, 8 on the current build
export PYTHONPATH=$PWD
TAG=${1:-r5h}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_register_layout.py tests/test_gpu_stream.py tests/test_gpu_api.py -m gpu -q 2>&1 | grep -v "^$" | tail -30 > gpurun_out/${TAG}_tests.txt
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_tests.txt | tail -12
timeout 900 python bench.py > /tmp/bench.log 2>&1; grep '^{' /tmp/bench.log | tail -1 > gpurun_out/${TAG}_bench_line.json; tail -3 /tmp/bench.log | cut -c1-300 | grep -v '^{'
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_line.json").read())
print({k: d[k] for k in ("value", "ms_per_step", "sustained_ms_per_step")}, "early", d["from_random_init"]["ms_per_step"])
print("roofline", {k: d["roofline"].get(k) for k in ("achieved", "frac", "kernel_ms", "vendor_gemm_tflops")})
print("vendor", {k: v for k, v in d["roofline"].get("vendor_gemm", {}).items() if isinstance(v, dict)})
print("auxk", [(r["n_dead_forced"], round(r["ms_per_step"], 3)) for r in d.get("auxk_active", [])])
print("other", [(r["config"][:12], round(r["ms_per_step"], 3), round(r["encoder_kernel_ms"], 3)) for r in d.get("other_configs", [])])
print("regimes", [(r["data"], round(r["ms_per_step"], 3), r["n_dead_last"], r["dense_route"]) for r in d.get("data_regimes", [])])
print("train_e2e", d.get("train_e2e"))
print("mse", d.get("mse_rel_err_vs_oracle"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
timeout 600 python tools/ddp_host_timing.py > gpurun_out/${TAG}_ddp_host_timing.json 2> /tmp/ddp.err; tail -2 /tmp/ddp.err | cut -c1-300
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_ddp_host_timing.json").read().strip().splitlines()[-1])
    for r in d["records"]:
        print(r)
except Exception as e:
    print("ddp timing:", e)
PY
rm -f gpurun_out/r5_eight_rank_rehearsal.txt; bash tools/experiments/r5_eight_rank_rehearsal.sh 2>&1 | tail -8
timeout 900 python tools/bench_train_e2e.py --gb 4 --root /dev/shm --epochs 2 > gpurun_out/${TAG}_train_e2e_tool.txt 2>&1; tail -8 gpurun_out/${TAG}_train_e2e_tool.txt | cut -c1-250
