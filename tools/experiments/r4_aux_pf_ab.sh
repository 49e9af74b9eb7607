#!/bin/bash
# one-pass AuxK with several trips of rows in flight (auxk.hip: aux_small_fused_kernel<NW, ND, PF>), builds side by side on ONE box:
#   tools/experiments/r4_aux_pf_ab.sh lib1.so lib2.so ...   (the LAST one is the tree's own build: parity of the AuxK tests on it)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py tests/test_gpu_configs.py tests/test_gpu_api.py -x -q -m gpu -k "aux or dead or Aux or known" 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 300 python tools/experiments/r4_fuzz_register_layout.py 2>&1 | tail -3
for rep in 1 2; do
for L in "$@"; do
  echo "== $L"
  for nd in 0 1 3 4 5 8; do SAEV_AMD_LIB=$L timeout 120 python tools/experiments/r4_aux_nd.py $nd 40 2>/dev/null; done
done
done
for L in "$@"; do
  rm -rf /tmp/prof_ab
  SAEV_AMD_LIB=$L timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_ab -o run -- python tools/experiments/r4_aux_nd.py 3 40 > /tmp/prof_ab.log 2>&1
  echo "== $L (3 dead latents)"
  python tools/rocpd_stats.py "$(find /tmp/prof_ab -name '*.db' | head -1)" --last 40 | grep -E "aux_|dead_|stats_reduce|gather_dead" | cut -c1-60,75-130
done
} 2>&1 | tee gpurun_out/r04_aux_pf_ab.txt
