export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/prof_a
SAEV_AMD_AUX_SMALL_MAX=-1 rocprofv3 --kernel-trace -d /tmp/prof_a -o run -- python tools/experiments/r4_aux_nd.py 30 > /dev/null 2>&1
python tools/rocpd_stats.py "$(find /tmp/prof_a -name '*.db' | head -1)" --last 20 > gpurun_out/r04_aux_nd30_dense_kernel_stats.txt
python tools/rocpd_gaps.py "$(find /tmp/prof_a -name '*.db' | head -1)" --last 15 | head -8
cat gpurun_out/r04_aux_nd30_dense_kernel_stats.txt | head -60
