#!/bin/bash
# encoder variants built as separate libraries (SAEV_AMD_LIB): steady-state step and encoder kernel time, alternating rounds
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for rep in 1 2 3; do for lib in "$@"; do
  SAEV_AMD_LIB=$PWD/saev_amd/$lib timeout 600 python bench.py --steps 60 --warmup 10 --sustained-steps 0 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras > /tmp/ab.log 2>&1
  python - "$lib" <<'PY'
import json, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line: print(sys.argv[1], "FAILED", open("/tmp/ab.log").read()[-300:])
else:
    d = json.loads(line[-1]); print(f"{sys.argv[1]:28s} ms {d['ms_per_step']:.4f}  early {d['from_random_init']['ms_per_step']:.4f}  enc {d['roofline']['kernel_ms']:.4f}")
PY
done; done | tee gpurun_out/r6m_enc_variants.txt
