#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline block (run on the GPU box from the repo root):
#   tools/collect_profiles.sh <tag>      ->  gpurun_out/<tag>_{kernel_stats.txt,pmc.txt,encoder_traffic.json,bench_line.json}
# Counters are collected in their own passes (one --pmc group per run, kernel-trace only), as MI355X_MICROARCH.md asks.
set -u
TAG=${1:-r01_x}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
W=/tmp/prof_$TAG
rm -rf "$W"
rocprofv3 --kernel-trace -d $W/kt -o run -- python bench.py --pretrain-steps 0 --steps 30 --warmup 5 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 > $W.kt.log 2>&1
python tools/rocpd_stats.py "$(find $W/kt -name '*.db' | head -1)" > "$OUT/${TAG}_kernel_stats.txt"
for grp in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  name=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $W/pmc_$name -o run -- python bench.py --pretrain-steps 0 --steps 3 --warmup 1 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 > $W.$name.log 2>&1
done
python tools/pmc_summary.py $W "$OUT/${TAG}_pmc.txt" "$OUT/${TAG}_encoder_traffic.json" "$TAG"
# the bench line last, so that its roofline block can cite the traffic file just written (bench.py reads profiles/)
mkdir -p profiles && cp "$OUT/${TAG}_encoder_traffic.json" profiles/r06_encoder_traffic.json 2>/dev/null
python bench.py 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_line.json"
