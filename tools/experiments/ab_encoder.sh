#!/bin/bash
# A/B the fused encoder kernel on ONE box: tools/experiments/ab_encoder.sh <lib A> <lib B> [mode] [rounds]
# (alternates the two libraries so that clock / temperature drift of the box hits both alike)
A=$1; B=$2; MODE=${3:-f16r}; R=${4:-3}
export PYTHONPATH=$PWD
for i in $(seq $R); do
  for L in $A $B; do echo -n "$L: "; SAEV_AMD_LIB=$L python tools/time_encoder.py $MODE 44 2>&1 | tail -1 | sed 's/.*median/median/'; done
done
