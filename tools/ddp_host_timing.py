"""Host cost of ONE data-parallel step against its device time: the sharded tail and the sparse-state exchange
each issue ~12 ctypes calls and 4-5 collectives per step from Python; at 2 048 rows per rank (configs[2]'s share of a 16 384-row
global batch on 8 GPUs) the device step is ~1 ms, so the enqueue must stay below that or the ranks run host-bound.

One rank over RCCL (the collectives are real launches into the communicator, with nobody to wait for), per mode and batch:
`enqueue_ms` = wall time of the Python loop that enqueues N steps without synchronising, per step; `device_ms` = the same N steps
to completion, per step.  enqueue < device means the GPU never waits for the host.

    python tools/ddp_host_timing.py [--steps 60]  -> one JSON line
"""
import argparse
import json
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    a = ap.parse_args()
    from saev_amd.engine import EngineConfig, SaeEngine
    from saev_amd.framework.ddp import DataParallelStepper, init_distributed

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_PORT", "29533")
    dist = init_distributed("nccl", rank=0, world_size=1, device=dev)
    out = {"steps": a.steps, "world": 1, "backend": "nccl (RCCL), one rank",
           "note": "the host may run at most four steps ahead of the device (saev_step_dead waits for the tracker record of four steps "
                   "ago), so at the real shape enqueue_ms tracks device_ms; the toy shape (64 x 512, 128 rows: ~0.1 ms of device work, the "
                   "same launches and collectives) shows what the host itself needs per step"}
    recs = []
    for D, S, K, row_list in ((64, 512, 8, (128,)), (1024, 32768, 32, (2048, 16384))):
      for tail, exchange in (("none", "single-process"), ("replicated", "dense"), ("sharded", "dense"), ("replicated", "sparse"), ("replicated", "c-abi")):
        for rows in row_list:
            single = exchange == "single-process"
            eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=rows, max_backward_rows=rows if exchange == "sparse" else 0,
                                         shard_world=1), dev)
            g = torch.Generator(device=dev).manual_seed(1)
            W = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
            W /= W.norm(dim=1, keepdim=True)
            eng.view("W_dec").copy_(W)
            eng.view("W_enc").copy_(W.t())
            del W
            if exchange == "c-abi":  # the same dense exchange issued by the library itself: ONE ctypes call per step (saev_train_step_dp)
                class _CAbi:
                    def train_step(self, x, lr, mn):
                        eng.train_step_dp(x, lr, mn)

                    def close(self):
                        pass

                eng.comm_init(eng.comm_unique_id(), 0, 1)
                st = _CAbi()
            else:
                st = DataParallelStepper(eng, None if single else dist, 1, force=not single, tail="replicated" if single else tail,
                                         exchange="dense" if single else exchange)
            x = torch.randn(rows, D, device=dev, generator=g) + torch.randn(D, device=dev, generator=g)
            for i in range(20):
                st.train_step(x, 1e-4, 1.0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                st.train_step(x, 1e-4, 1.0)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            st.close()
            eng.close()
            recs.append({"shape": f"{D} x {S}", "mode": f"{exchange}" if single else ("dense exchange, replicated tail, collectives inside the library (saev_train_step_dp)" if exchange == "c-abi" else f"{exchange} exchange, {tail} tail"), "rows_per_rank": rows,
                         "enqueue_ms": (t1 - t0) / a.steps * 1e3, "device_ms": (t2 - t0) / a.steps * 1e3,
                         })
            del eng, x
            torch.cuda.empty_cache()
    out["records"] = recs
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
