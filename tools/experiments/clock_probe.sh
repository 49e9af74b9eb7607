#!/bin/bash
# effective clock and MFMA-pipe occupancy of encoder ablation variants (tools/ablate_encoder.py must have been run)
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
export PYTHONPATH=$PWD
for v in FULL NOEPI NOSTAGE+NOLDS+NOEPI NOBAR+NOSTAGE+NOLDS+NOEPI NOMFMA+NOEPI; do
  lib=build/abl/lib_$v.so; [ "$v" = FULL ] && lib=saev_amd/libsaev_amd.so
  rm -rf /tmp/clk_$v
  SAEV_AMD_LIB=$lib rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA --output-format csv -d /tmp/clk_$v -o run -- python tools/time_encoder.py f16r 10 > /tmp/clk_$v.log 2>&1
  python - "$v" <<'PY'
import csv, glob, sys, collections
v = sys.argv[1]
c = collections.defaultdict(list); d = []
for f in glob.glob(f"/tmp/clk_{v}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "encode_f16x3_kernel<1" in r["Kernel_Name"]: c[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"/tmp/clk_{v}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "encode_f16x3_kernel<1" in r["Kernel_Name"]: d.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
m = {k: sum(x[2:]) / len(x[2:]) for k, x in c.items()}
dur = sum(d[2:]) / len(d[2:])
gui = m.get("GRBM_GUI_ACTIVE", 0)
print(f"{v:28s} {dur/1e3:8.1f} us  clock {gui/8/dur:5.2f} GHz  mfma_busy {100*m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(gui/8*1024):5.1f}%  "
      f"wait_any {100*m.get('SQ_WAIT_ANY',0)/m.get('SQ_WAVE_CYCLES',1):5.1f}%  wait_inst {100*m.get('SQ_WAIT_INST_ANY',0)/m.get('SQ_WAVE_CYCLES',1):5.1f}%  "
      f"active {100*m.get('SQ_ACTIVE_INST_ANY',0)/m.get('SQ_WAVE_CYCLES',1):5.1f}%  wave_cycles {m.get('SQ_WAVE_CYCLES',0):.3e}")
PY
done
