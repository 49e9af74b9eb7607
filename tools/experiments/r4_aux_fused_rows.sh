export PYTHONPATH=$PWD
for L in build/abl/lib_AF32.so build/abl/lib_AFP.so; do
  echo "== $L"; for nd in 0 2 8; do SAEV_AMD_LIB=$L python tools/experiments/r4_aux_nd.py $nd 40 2>/dev/null; done
  bash tools/experiments/r4_aux_nd_profile.sh $L 4 | grep -i "aux_small_fused\|aux_fused_wsum"
done
