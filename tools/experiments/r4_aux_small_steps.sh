#!/bin/bash
# per-step kernel tables of the few-dead-latents AuxK routes (shipped defaults) at several dead counts: only the AuxK kernels
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for nd in "$@"; do
  rm -rf /tmp/prof_nd
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_nd -o run -- python tools/experiments/r4_aux_nd.py $nd 24 > /tmp/prof_nd.log 2>&1
  echo "== n_dead $nd: $(tail -1 /tmp/prof_nd.log)"
  python tools/rocpd_per_step.py "$(find /tmp/prof_nd -name '*.db' | head -1)" --steps 20 | grep -E "^steps|aux_|dead_|colsum|stats_reduce|sum_parts|fillBuffer|scatter_add|gather_dead"
done 2>&1 | tee gpurun_out/r04_aux_small_steps.txt
