#!/bin/bash
# per-step kernel tables at forced dead counts on the round-5 build (the natural route of each count)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
tag=$1; shift
for nd in "$@"; do
  rm -rf /tmp/prof_nd
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_nd -o run -- python tools/experiments/r4_aux_nd.py $nd 24 > /tmp/prof_nd.log 2>&1
  echo "== n_dead $nd: $(grep n_dead /tmp/prof_nd.log | tail -1)"
  python tools/rocpd_per_step.py "$(find /tmp/prof_nd -name '*.db' | head -1)" --steps 20
done 2>&1 | tee gpurun_out/${tag}_aux_steps.txt
