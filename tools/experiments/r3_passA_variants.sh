#!/bin/bash
# pass A of dw_slices with sub-blocks of 2 pairs at five waves per SIMD against the shipped 4 pairs at four waves: rebuilds
# sparse.o on the box, same box for both.   tools/experiments/r3_passA_variants.sh <tag>
TAG=${1:-r3_passA}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export PYTHONPATH=$PWD
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-pass-failed -Wno-unused-value -Iinclude"
OPTS="--steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
run() {
  echo "== $1" >> $OUT/${TAG}.txt
  python bench.py $OPTS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])" >> $OUT/${TAG}.txt
  rm -rf /tmp/pv; rocprofv3 --kernel-trace -d /tmp/pv -o run -- python bench.py $OPTS > /dev/null 2>&1
  python tools/rocpd_stats.py "$(find /tmp/pv -name '*.db' | head -1)" | grep -E "dw_slices" >> $OUT/${TAG}.txt
}
run "shipped (PB 4, 4 waves)"
for v in "2 5" "4 4"; do
  set -- $v
  /opt/rocm/bin/hipcc $FL -DDWS_PB_A=$1 -DDWS_WAVES_A=$2 -c saev_amd/csrc/sparse.hip -o build/sparse.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build/*.o -o saev_amd/libsaev_amd.so
  run "PB $1, $2 waves"
done
