#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py tests/test_gpu_api.py tests/test_gpu_configs.py tests/test_gpu_ddp.py -x -q -m gpu > gpurun_out/r04_auxf_tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/r04_auxf_tests.log | tail -2
for L in build/abl/lib_RS8.so build/abl/lib_FUSED.so; do
  echo "== $L"
  for nd in 0 1 4 8 9 16 32; do SAEV_AMD_LIB=$L python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
done
