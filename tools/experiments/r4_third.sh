#!/bin/bash
# round 4, re-entry: the whole gpu suite on HEAD, the bench line, steady-state kernel stats and queue gaps
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04c}
python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_all.log 2>&1; echo "all rc=$?"; tail -6 gpurun_out/${TAG}_gpu_all.log
python bench.py --no-cpu-baseline 2>gpurun_out/${TAG}_bench.err | tail -1 > gpurun_out/${TAG}_bench_line.json; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_line.json"))
print("value", d["value"], "ms", d["ms_per_step"], "enc", d["roofline"]["kernel_ms"], d["roofline"]["frac"])
print("early", d.get("from_random_init"))
print("sustained", d["sustained"]["ms_per_step"], d["timed_region"])
print("aux", d["auxk_active"])
for o in d["other_configs"]: print(o["config"][:30], o["ms_per_step"], o["encoder_frac_of_peak"])
PY
rm -rf /tmp/prof_st
rocprofv3 --kernel-trace -d /tmp/prof_st -o run -- python bench.py --steps 100 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-auxk-probe --no-other-configs > /tmp/prof_st.log 2>&1
DB=$(find /tmp/prof_st -name '*.db' | head -1)
python tools/rocpd_stats.py "$DB" --last 100 > gpurun_out/${TAG}_kernel_stats_steady.txt
python tools/rocpd_gaps.py "$DB" --last 100 > gpurun_out/${TAG}_gaps_steady.txt
tail -1 /tmp/prof_st.log > gpurun_out/${TAG}_steady_bench_line.json
head -30 gpurun_out/${TAG}_kernel_stats_steady.txt; cat gpurun_out/${TAG}_gaps_steady.txt | head -20
