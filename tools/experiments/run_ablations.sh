for v in "" NOEPI NOCAND NOSTAGE+NOEPI NOLDS+NOEPI NOSTAGE+NOLDS+NOEPI NOMFMA+NOEPI NOBAR+NOSTAGE+NOLDS+NOEPI NOMFMA+NOLDS+NOEPI NOMFMA+NOSTAGE+NOEPI; do
  if [ -z "$v" ]; then echo -n "FULL: "; python tools/time_encoder.py f16r 14 2>&1 | tail -1; else echo -n "$v: "; SAEV_AMD_LIB=build/abl/lib_$v.so python tools/time_encoder.py f16r 14 2>&1 | tail -1; fi
done
