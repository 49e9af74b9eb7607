#!/bin/bash
# the sustained segment is ONE chaotic trajectory of the dead count: compare the two AuxK rules over several (other pretrain lengths)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
for pre in 1400 1700 2000; do for wide in 0 1; do
  SAEV_AMD_AUX_WIDE=$wide timeout 900 python bench.py --steps 40 --warmup 10 --pretrain-steps $pre --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras > /tmp/ab.log 2>&1
  python - "$wide" "$pre" <<'PY'
import json, sys
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line: print(sys.argv[1], "FAILED", open("/tmp/ab.log").read()[-300:])
else:
    d = json.loads(line[-1]); print(f"pretrain {sys.argv[2]} aux_wide_route {sys.argv[1]}: steady {d['ms_per_step']:.4f}  sustained {d['sustained_ms_per_step']:.4f}  (n_dead_last {d['sustained']['n_dead_last']}, route_last {d['sustained']['aux_route_last']})")
PY
done; done | tee gpurun_out/r6q_aux_wide_ab2.txt
