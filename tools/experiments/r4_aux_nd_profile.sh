#!/bin/bash
# per-kernel table of the step with N forced dead latents: tools/experiments/r4_aux_nd_profile.sh <lib> <n_dead> [<n_dead> ...]
export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
L=$1; shift
for nd in "$@"; do
  rm -rf /tmp/prof_nd
  SAEV_AMD_LIB=$L rocprofv3 --kernel-trace -d /tmp/prof_nd -o run -- python tools/experiments/r4_aux_nd.py $nd > /tmp/prof_nd.log 2>&1
  echo "== $L n_dead $nd"; tail -1 /tmp/prof_nd.log
  python tools/rocpd_stats.py "$(find /tmp/prof_nd -name '*.db' | head -1)" --last 20 | cut -c1-62,75-120 | head -48
done
