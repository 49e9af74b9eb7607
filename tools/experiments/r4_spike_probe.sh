export PYTHONPATH=$PWD
echo "== current build (one-pass AuxK up to 8, few-dead kernels up to 40)"; python tools/experiments/r4_spike_probe.py 512 25 2>/dev/null
echo "== current build, five-pass kernels up to 64 (no one-pass form)"; SAEV_AMD_AUX_SMALL_MAX=64 python tools/experiments/r4_spike_probe.py 512 25 2>/dev/null
echo "== current build, dense route always"; SAEV_AMD_AUX_SMALL_MAX=-1 python tools/experiments/r4_spike_probe.py 512 25 2>/dev/null
echo "== no AuxK (k_aux 0)"; python tools/experiments/r4_spike_probe.py 0 25 2>/dev/null
