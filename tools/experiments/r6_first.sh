#!/bin/bash
# round 6, first GPU call: the whole -m gpu suite with the skip reasons listed (-rs), the default bench line (new cpu_baseline
# protocol), the vendor GEMM anchor
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -rs 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r6a_gputests.txt
tail -8 gpurun_out/r6a_gputests.txt
timeout 900 python bench.py > gpurun_out/r6a_bench.log 2> gpurun_out/r6a_bench.err
grep '^{' gpurun_out/r6a_bench.log | tail -1 > gpurun_out/r6a_bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6a_bench_line.json"))
print("ms", d["ms_per_step"], "sustained", d.get("sustained_ms_per_step"), "enc", d["roofline"]["kernel_ms"], "vendor", d["roofline"].get("vendor_gemm_tflops"))
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "seconds_per_step")}, d["cpu_baseline"]["threads_8"])
print("aux", [(a["n_dead"], round(a["ms_per_step"], 3)) for a in d.get("auxk_active", [])])
print("other", [(o["config"][:12], round(o["ms_per_step"], 3)) for o in d.get("other_configs", [])])
print("e2e", d.get("train_e2e", {}).get("train_over_engine_loop"))
PY
