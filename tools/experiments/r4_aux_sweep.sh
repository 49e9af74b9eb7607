#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${1:-r04_aux}
{
echo "# few-dead-latents route (count <= 64) / dense route beyond"
for nd in 0 1 8 16 30 48 64 100 1000; do python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
echo "# dense route forced (SAEV_AMD_AUX_SMALL_MAX=-1)"
for nd in 8 30 64; do SAEV_AMD_AUX_SMALL_MAX=-1 python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
} | tee gpurun_out/${TAG}_sweep.txt
for nd in 30 1000; do
  rm -rf /tmp/prof_aux$nd
  rocprofv3 --kernel-trace -d /tmp/prof_aux$nd -o run -- python tools/experiments/r4_aux_nd.py $nd > /dev/null 2>&1
  python tools/rocpd_stats.py "$(find /tmp/prof_aux$nd -name '*.db' | head -1)" --last 20 > gpurun_out/${TAG}_nd${nd}_kernel_stats.txt
done
grep -i "aux\|dead\|colsum\|encode_f16x3\|split\|absmax\|pow2\|select_dense\|sum_parts\|mask\|scale_pair\|fill" gpurun_out/${TAG}_nd30_kernel_stats.txt | head -20
echo ----
grep -i "aux\|dead\|colsum\|encode_f16x3\|split\|absmax\|pow2\|select_dense\|sum_parts\|mask\|scale_pair\|fill" gpurun_out/${TAG}_nd1000_kernel_stats.txt | head -30
