#!/bin/bash
# L2 hit / miss counts per kernel of the train step (one --pmc pass, kernel-trace only)
export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rm -rf /tmp/l2p
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/l2p -o run -- python bench.py --pretrain-steps 300 --steps 3 --warmup 1 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 > /tmp/l2p.log 2>&1
python3 - "$(find /tmp/l2p -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:70]
    a = acc.setdefault(k, {}); c = a.setdefault(r['Counter_Name'], [0, 0.0]); c[0] += 1; c[1] += float(r['Counter_Value'])
rows = []
for k, a in acc.items():
    h = a.get('TCC_HIT_sum', [1, 0])[1] / max(1, a.get('TCC_HIT_sum', [1, 0])[0]); m = a.get('TCC_MISS_sum', [1, 0])[1] / max(1, a.get('TCC_MISS_sum', [1, 0])[0])
    rows.append((h + m, k, h, m))
for t, k, h, m in sorted(rows, reverse=True)[:16]:
    print(f"{k:70s} hits {h:12.0f} misses {m:12.0f}  hit rate {h / max(1.0, h + m):.3f}  (requests x 128 B = {(h + m) * 128 / 1e9:.2f} GB)")
PY
