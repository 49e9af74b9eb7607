export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
( time timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04g_gpu_tests.log 2>&1 ) 2>&1 | grep real
grep -E "passed|failed|error" gpurun_out/r04g_gpu_tests.log | tail -2
timeout 600 bash tools/experiments/r4_steady_profile.sh r04g > /dev/null 2>&1
python tools/rocpd_per_step.py "$(find /tmp/prof_st -name '*.db' | head -1)" --steps 100 > gpurun_out/r04g_per_step_steady.txt 2>&1
rm -rf /tmp/prof_kt; timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_kt -o run -- python bench.py --pretrain-steps 0 --steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 > /tmp/kt.log 2>&1
python tools/rocpd_stats.py "$(find /tmp/prof_kt -name '*.db' | head -1)" > gpurun_out/r04g_kernel_stats.txt
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04g_bench_line.json
python -c "
import json; d=json.load(open('gpurun_out/r04g_bench_line.json'))
print('steady', d['ms_per_step'], 'value', d['value'], 'frac', d['roofline']['frac'], 'enc', d['roofline']['kernel_ms'])
print('early', d['from_random_init']['ms_per_step'], 'sustained', d['sustained']['ms_per_step'])
print([ (a['n_dead'], round(a['ms_per_step'],3)) for a in d['auxk_active']], [round(o['ms_per_step'],3) for o in d['other_configs']])"
head -4 gpurun_out/r04g_per_step_steady.txt; head -5 gpurun_out/r04g_gaps_steady.txt
