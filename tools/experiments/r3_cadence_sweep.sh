for first in 4 8 16; do for every in 1 2 4; do
  r=$(SAEV_AMD_REFRESH_FIRST=$first SAEV_AMD_REFRESH_EVERY=$every python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['kernel_ms'],4), round(d['ms_per_step'],4), d['cand_max'])")
  echo "first=$first every=$every: $r"
done; done
