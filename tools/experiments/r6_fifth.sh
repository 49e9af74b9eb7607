#!/bin/bash
# round 6, fifth GPU call: the dense AuxK route with both operand forms from one pass (split_both_kernel): AuxK tests, A/B at forced
# dead counts, per-step kernel table at 1 000 dead latents
export PYTHONPATH=$PWD
mkdir -p gpurun_out
T=r6g
timeout 1500 python -m pytest tests/test_gpu_known_answers.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ddp.py tests/test_gpu_register_layout.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -20 > gpurun_out/${T}_tests.txt
tail -5 gpurun_out/${T}_tests.txt
for rep in 1 2; do for route in 0 1; do for nd in 100 1000 3000; do
  echo "aux_split_route $route: $(SAEV_AMD_AUX_SPLIT=$route timeout 300 python tools/experiments/r4_aux_nd.py $nd 30 2>&1 | grep n_dead | tail -1)"
done; done; done 2>&1 | tee gpurun_out/${T}_aux_split_ab.txt
bash tools/experiments/r5_aux_dense_steps.sh r6g 1000 > /dev/null 2>&1; head -45 gpurun_out/r6g_aux_steps.txt
