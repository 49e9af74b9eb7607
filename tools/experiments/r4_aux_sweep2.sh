#!/bin/bash
# AuxK cost against the dead count on both routes with the current build (few-dead-latents kernels forced up to 64; dense forced)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
{
echo "# shipped defaults (one-pass kernel up to 8 dead latents, five-pass kernels up to 40, dense algebra beyond)"
for nd in 0 1 2 4 8 9 16 32 40 41 64; do python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
echo "# five-pass few-dead-latents kernels forced up to 64 dead latents, no one-pass kernel (SAEV_AMD_AUX_SMALL_MAX=64)"
for nd in 0 1 4 8 16 24 32 48 64; do SAEV_AMD_AUX_SMALL_MAX=64 python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
echo "# dense route forced (SAEV_AMD_AUX_SMALL_MAX=-1)"
for nd in 8 16 32 64 128 256 1000; do SAEV_AMD_AUX_SMALL_MAX=-1 python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
} | tee gpurun_out/${1:-r04}_aux_route_sweep.txt
