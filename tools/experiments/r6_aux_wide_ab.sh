#!/bin/bash
# sustained segment (2 000 steps past the dead-latent threshold) with the matrix-core AuxK kernels up to 128 dead latents (default) and up
# to 64 as in round 5 (SAEV_AMD_AUX_WIDE=1), alternating
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for rep in 1 2; do for wide in 0 1; do
  SAEV_AMD_AUX_WIDE=$wide timeout 900 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras > /tmp/ab.log 2>&1
  python - "$wide" <<'PY'
import json, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line: print(sys.argv[1], "FAILED", open("/tmp/ab.log").read()[-300:])
else:
    d = json.loads(line[-1]); print(f"aux_wide_route {sys.argv[1]}: steady {d['ms_per_step']:.4f}  sustained {d['sustained_ms_per_step']:.4f}  (n_dead_last {d['sustained']['n_dead_last']}, route_last {d['sustained']['aux_route_last']}, readbacks {d['sustained']['n_dead_readbacks_in_segment']})")
PY
done; done | tee gpurun_out/r6o_aux_wide_ab.txt
