#!/bin/bash
# Encoder staging loads under other cache policies / rasterisations (the 8.6x fabric traffic of the first pass).
#   libraries: the shipped one; W tiles loaded non-temporal (they stream: an XCD's eight batch blocks use a tile within a quarter of a
#   tile time and never again), x blocks non-temporal, W tiles sc1;  grids: 256 workgroups (per XCD 8 batch blocks x 4 latent ranges at a
#   time) and 512 (4 batch blocks x 8 latent ranges: 2 MB of x per XCD, which an L2 that did not keep the W stream could hold)
# per variant: encoder kernel time (HIP events, median of 40) and FETCH_SIZE / WRITE_SIZE of the kernel (rocprofv3 --pmc, own passes)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/${1:-r05b}_enc_policy.txt
: > $OUT
for round in 1 2; do
for wgs in 0 512; do
for lib in libsaev_amd.so libsaev_amd_wnt.so libsaev_amd_xnt.so libsaev_amd_wsc1.so; do
  t=$(SAEV_AMD_ENC_WGS=$wgs SAEV_AMD_LIB=$PWD/saev_amd/$lib timeout 300 python tools/time_encoder.py f16r 44 2>&1 | tail -1 | sed 's/.*median/median/')
  echo "round $round wgs ${wgs} $lib: $t" | tee -a $OUT
done; done; done
for wgs in 0 512; do
for lib in libsaev_amd.so libsaev_amd_wnt.so libsaev_amd_xnt.so; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pol
    SAEV_AMD_ENC_WGS=$wgs SAEV_AMD_LIB=$PWD/saev_amd/$lib timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pol -o run -- python tools/time_encoder.py f16r 8 > /tmp/pol.log 2>&1
    python - "$c" "$wgs" "$lib" <<'PY' | tee -a $OUT
import csv, glob, sys
c, wgs, lib = sys.argv[1:4]
f = glob.glob('/tmp/pol/**/*counter_collection.csv', recursive=True)
vals = []
for fn in f:
    for r in csv.DictReader(open(fn)):
        if 'encode_m16' in r.get('Kernel_Name', '') and r.get('Counter_Name') == c:
            vals.append(float(r['Counter_Value']))
if vals:
    vals = vals[len(vals) // 2:]
    print(f"wgs {wgs} {lib}: {c} per launch (raw counter units) {sum(vals) / len(vals):.4g} over {len(vals)} launches")
else:
    print(f"wgs {wgs} {lib}: {c}: no rows ({len(f)} files)")
PY
  done
done; done
