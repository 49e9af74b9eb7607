#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r04_gpu_all.log 2>&1; echo "all rc=$?"; tail -4 gpurun_out/r04_gpu_all.log
python bench.py --no-cpu-baseline 2>gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04b_bench_line.json; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04b_bench_line.json"))
print("value", d["value"], "ms", d["ms_per_step"], "enc", d["roofline"]["kernel_ms"], d["roofline"]["frac"])
print("early", d.get("from_random_init"))
print("sustained", d["sustained"]["ms_per_step"], d["timed_region"])
print("aux", d["auxk_active"])
for o in d["other_configs"]: print(o["config"][:30], o["ms_per_step"], o["encoder_frac_of_peak"])
PY
bash tools/experiments/rehearse_two_ranks.sh > gpurun_out/r04_two_rank_rehearsal.txt 2>&1
python - <<'PY'
import json
for line in open("gpurun_out/r04_two_rank_rehearsal.txt"):
    if line.startswith("#"): print(line.strip()); continue
    try: d=json.loads(line)
    except Exception as e: print("  unparsable:", line[:200]); continue
    print("  ", d["ms_per_step"], d["config"]["grad_exchange"][:60], d.get("exchange_selection"), list((d.get("collectives") or {}).keys())[:3])
PY
