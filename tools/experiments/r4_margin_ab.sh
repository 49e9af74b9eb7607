#!/bin/bash
# measured rounding-error norms in the f16r margin (select.hip: f16r_margin) against the a-priori 2^-11 per operand:
# the GPU suite first, then step time and the refinement chain's kernel times, alternating builds on ONE box
#   tools/experiments/r4_margin_ab.sh saev_amd/lib_base.so saev_amd/libsaev_amd.so
export PYTHONPATH=$PWD
mkdir -p gpurun_out
{
( time timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) 2>&1
timeout 600 bash tools/experiments/r4_lib_ab.sh 2 "$@"
timeout 600 bash tools/experiments/r4_kernel_ab.sh "refine_slices|refine_sum|select_cand|encode_m16|center_stats|split_f16r|bias_finish" "$@"
} 2>&1 | tee gpurun_out/r04_margin_ab.txt
