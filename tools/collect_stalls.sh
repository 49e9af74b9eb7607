#!/bin/bash
# Stall attribution for the hot kernels (run on the GPU box from the repo root):
#   tools/collect_stalls.sh <tag> [bench args]   ->  gpurun_out/<tag>_stalls.txt
# Each --pmc group is its own run with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 section: 8 SQ slots per pass).
set -u
TAG=${1:-r02_x}
shift || true
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export PYTHONPATH=$PWD
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
W=/tmp/stalls_$TAG
rm -rf "$W"
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $W/pmc_g$i -o run -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-auxk-probe --no-other-configs --no-extras --sustained-steps 0 "$@" > $W.g$i.log 2>&1 || tail -5 $W.g$i.log
done
python tools/stall_summary.py $W "$OUT/${TAG}_stalls.txt" "$TAG"
