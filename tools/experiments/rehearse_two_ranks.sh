#!/bin/bash
# The N > 1 code path of bench.py on a ONE-GPU box: two ranks on cuda:0 over gloo (RCCL refuses two ranks on one device;
# gloo stages device tensors through the host).  Not a performance number -- both ranks share one GPU and every collective
# goes through host memory -- but the schema of the multi-rank JSON line, the start-up selection of the exchange
# (framework/ddp.py: choose_exchange) and the three exchanges end to end.  Last: ONE rank over RCCL (--force-dist), which runs
# the same in-place reduce-scatter / all-gather calls and the busbw probe on the real backend.
#   tools/experiments/rehearse_two_ranks.sh > gpurun_out/rehearsal.txt
run() {
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 \
    --backend gloo --same-device --pretrain-steps 30 --steps 6 --warmup 2 --sustained-steps 0 --no-auxk-probe --no-cpu-baseline "${@:2}" 2>/dev/null | tail -1
}
echo "# replicated tail, dense exchange, 2048 rows per rank"; run 29541 --batch 2048 --tail replicated
echo "# sharded tail, dense exchange, 2048 rows per rank";    run 29542 --batch 2048 --tail sharded
echo "# --tail auto (self-check, then sharded), 2048 rows per rank"; run 29545 --batch 2048 --tail auto
echo "# sparse-state exchange, 2048 rows per rank";           run 29543 --batch 2048 --exchange sparse
echo "# --exchange auto (2048 rows per rank <= 4096: the sparse exchange after its self-check)"; run 29546 --batch 2048 --exchange auto
echo "# sparse-state exchange, global batch 16384 (strong scaling)"; run 29544 --global-batch 16384 --exchange sparse
echo "# one rank over RCCL (--force-dist): sharded tail with the in-place collectives, busbw probe (n = 1: factors are 0)"
python bench.py --force-dist --tail sharded --pretrain-steps 30 --steps 6 --warmup 2 --sustained-steps 0 --no-auxk-probe --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1
