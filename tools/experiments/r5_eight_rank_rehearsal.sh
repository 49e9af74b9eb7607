#!/bin/bash
# The 8-rank bench line, rehearsed on ONE GPU at a toy shape over gloo (RCCL refuses several ranks on one device): does the line parse,
# which exchange does the start-up self-check select, do `collectives` / `exchange_selection` appear.  Not a performance number.
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
for ex in auto dense; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 2954${#ex} bench.py \
    --gpus 8 --steps 5 --warmup 2 --pretrain-steps 0 --sustained-steps 0 --shape 256,2048,16 --batch 256 --backend gloo --same-device \
    --exchange $ex --tail auto > /tmp/r8_$ex.log 2>&1
  echo "exchange=$ex rc=$?" >> gpurun_out/r5_eight_rank_rehearsal.txt
  grep '^{' /tmp/r8_$ex.log | tail -1 >> gpurun_out/r5_eight_rank_rehearsal.txt
  grep -iE "error|Traceback" /tmp/r8_$ex.log | head -5 >> gpurun_out/r5_eight_rank_rehearsal.txt
done
python - <<'PY'
import json
for line in open("gpurun_out/r5_eight_rank_rehearsal.txt"):
    if line.startswith("{"):
        d = json.loads(line)
        print({k: d.get(k) for k in ("n_gpus", "value", "ms_per_step", "scaling")}, d["config"]["grad_exchange"][:60], d.get("exchange_selection"))
    else:
        print(line.strip())
PY
