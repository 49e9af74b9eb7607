#!/bin/bash
# the fp32-MFMA few-dead-latents kernels: tests that reach them, then step times / per-step tables at 9 ... 32 dead latents on both routes
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py tests/test_gpu_configs.py tests/test_gpu_ddp.py -m gpu -q -x -k "aux or dead or matrix_cores" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/${1}_aux_mfma_tests.txt
tail -12 gpurun_out/${1}_aux_mfma_tests.txt
{
for nd in 0 9 16 24 32 33 40; do timeout 120 python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
echo "# SAEV_AMD_AUX_SMALL_MAX=64 (the few-dead-latents route up to its capacity)"
for nd in 33 40 48 64; do SAEV_AMD_AUX_SMALL_MAX=64 timeout 120 python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
echo "# dense route (SAEV_AMD_AUX_SMALL_MAX=-1)"
for nd in 40 64; do SAEV_AMD_AUX_SMALL_MAX=-1 timeout 120 python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
echo "# SAEV_AMD_AUX_SMALL=1 (vector-ALU kernels)"
for nd in 9 16 24 32; do SAEV_AMD_AUX_SMALL=1 timeout 120 python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
} | tee gpurun_out/${1}_aux_mfma_sweep.txt
bash tools/experiments/r5_aux_dense_steps.sh ${1}_mfma 24 > /dev/null 2>&1
grep -v "^$" gpurun_out/${1}_mfma_aux_steps.txt | head -40
