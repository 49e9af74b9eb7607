export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py tests/test_gpu_known_answers.py -x -q -m gpu -k "aux or dead or Aux" 2>&1 | grep -E "passed|failed"
for L in build/abl/lib_RS8.so build/abl/lib_FUSED.so; do
  echo "== $L"
  for nd in 0 1 4 8; do SAEV_AMD_LIB=$L python tools/experiments/r4_aux_nd.py $nd 2>/dev/null; done
done
bash tools/experiments/r4_aux_nd_profile.sh build/abl/lib_FUSED.so 4 | grep -i "aux\|dead\|colsum\|stats_red"
