#!/bin/bash
# encoder time and step time for bound-refresh cadences and group counts (bench loop, 30 steps)
for ng in 32 64; do for first in 2 4 8; do for every in 1 2 4 8; do
  r=$(SAEV_AMD_NGROUPS=$ng SAEV_AMD_REFRESH_FIRST=$first SAEV_AMD_REFRESH_EVERY=$every python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --sustained-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4), d['cand_max'])")
  echo "ng=$ng first=$first every=$every: $r"
done; done; done
