#!/bin/bash
# rocprofv3 counters of a microbenchmark binary, one --pmc group per pass (run on the GPU box from the repo root):
#   tools/experiments/ubench_pmc.sh build/ubench/dw_cols out_tag
BIN=$1; TAG=$2
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for grp in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/up_$name
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/up_$name -o run -- $BIN > /tmp/up_$name.log 2>&1
  python3 - "$(find /tmp/up_$name -name '*counter_collection.csv' | head -1)" >> $OUT/${TAG}_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'][:60], r['Counter_Name'])
    a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, v) in acc.items():
    print(f"{k:60s} {c:14s} launches {n:4d}  mean {v / n:14.1f}")
PY
done
