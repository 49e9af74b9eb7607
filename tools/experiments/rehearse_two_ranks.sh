#!/bin/bash
# The N > 1 code path of bench.py on a ONE-GPU box: two ranks on cuda:0 over gloo (RCCL refuses two ranks on one device;
# gloo stages device tensors through the host).  Not a performance number -- both ranks share one GPU and every collective
# goes through host memory -- but the schema of the multi-rank JSON line and the three exchanges end to end.
#   tools/experiments/rehearse_two_ranks.sh > gpurun_out/rehearsal.txt
run() {
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 \
    --backend gloo --same-device --steps 6 --warmup 2 --sustained-steps 0 --no-auxk-probe --no-cpu-baseline "${@:2}" 2>/dev/null | tail -1
}
echo "# replicated tail, dense exchange, 2048 rows per rank"; run 29541 --batch 2048 --tail replicated
echo "# sharded tail, dense exchange, 2048 rows per rank";    run 29542 --batch 2048 --tail sharded
echo "# sparse-state exchange, 2048 rows per rank";           run 29543 --batch 2048 --exchange sparse
echo "# sparse-state exchange, global batch 16384 (strong scaling)"; run 29544 --global-batch 16384 --exchange sparse
