#!/bin/bash
# AuxK cost against the dead count with the round-5 dense route: where do the few-dead-latents kernels and the dense algebra meet now?
export PYTHONPATH=$PWD
mkdir -p gpurun_out
{
echo "# shipped defaults"
for nd in 0 8 9 16 24 32 40 41 64 128 256 512 1000 2000; do python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
echo "# dense route forced (SAEV_AMD_AUX_SMALL_MAX=-1)"
for nd in 9 16 20 24 28 32 36 40; do SAEV_AMD_AUX_SMALL_MAX=-1 python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
} | tee gpurun_out/${1:-r05b}_aux_route_sweep.txt
