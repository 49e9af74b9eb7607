export PYTHONPATH=$PWD
for m in f16x3 f32 f16r; do echo "== $m"; SAEV_AMD_ENCODER=$m python tools/experiments/r4_diag_growing.py 2>&1 | tail -40; done
