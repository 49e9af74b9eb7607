#!/bin/bash
# per-step kernel tables of the dense AuxK route (every dead set sent there: SAEV_AMD_AUX_SMALL_MAX=-1) at several dead counts
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for nd in "$@"; do
  rm -rf /tmp/prof_nd
  SAEV_AMD_AUX_SMALL_MAX=-1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_nd -o run -- python tools/experiments/r4_aux_nd.py $nd 24 > /tmp/prof_nd.log 2>&1
  echo "== dense route, n_dead $nd: $(tail -1 /tmp/prof_nd.log)"
  python tools/rocpd_per_step.py "$(find /tmp/prof_nd -name '*.db' | head -1)" --steps 20
done 2>&1 | tee gpurun_out/r04_aux_dense_steps.txt
