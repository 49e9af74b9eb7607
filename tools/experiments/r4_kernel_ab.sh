#!/bin/bash
# average duration of kernels matching a pattern under several builds on ONE box (rocprofv3 kernel trace of the steady bench loop):
#   tools/experiments/r4_kernel_ab.sh <pattern> lib1.so lib2.so ...
export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
PAT=$1; shift
for L in "$@"; do
  rm -rf /tmp/prof_ab
  SAEV_AMD_LIB=$L rocprofv3 --kernel-trace -d /tmp/prof_ab -o run -- python bench.py --steps 60 --warmup 5 --sustained-steps 0 --no-cpu-baseline --no-auxk-probe --no-other-configs > /tmp/prof_ab.log 2>&1
  echo "== $L"
  python tools/rocpd_stats.py "$(find /tmp/prof_ab -name '*.db' | head -1)" --last 60 | grep -E "$PAT" | cut -c1-60,75-130
done
