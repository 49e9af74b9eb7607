"""The COMPUTE side of strong scaling at 8 ranks, measured on one GPU (no multi-GPU node here: the collectives are the only
part of the prediction in DESIGN.md section 5 that stays a prediction).  configs[2]'s global batch of 16 384 rows on 8 ranks
= 2 048 rows per rank:

  sparse-state exchange: forward + tracker on 2 048 rows, copy of the step state, backward over 16 384 gathered rows (the
                         local rows repeated eight times stand in for the other ranks'), fused tail;
  dense exchange:        forward + tracker + backward on 2 048 rows, then the whole tail (replicated) or 1/8 of it (sharded).

    python tools/experiments/r3_strong_scaling_compute.py
"""
import json
import math
import pathlib
import sys
import time

import torch

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
from saev_amd.engine import EngineConfig, SaeEngine  # noqa: E402

D, S, K, W, NL = 1024, 32768, 32, 8, 2048


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    Wd = (torch.rand(S, D, device=dev, generator=g) * 2 - 1) * math.sqrt(6.0 / D)
    Wd /= Wd.norm(dim=1, keepdim=True)
    pool = torch.randn(16 * NL, D, device=dev, generator=g) + torch.randn(D, device=dev, generator=g)
    out = {}
    # ---- sparse-state exchange, compute only
    eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=W * NL, aux_dead_cap=4096), dev)
    eng.view("W_dec").copy_(Wd); eng.view("W_enc").copy_(Wd.t())
    x_all, g_all, idx_all, val_all = eng.gather_buffers(W, NL)
    it = [0]

    def sparse_step():
        i = it[0] = it[0] + 1
        x = pool[(i % 16) * NL:(i % 16 + 1) * NL]
        eng.step_forward(x, training=True, n_rows_global=W * NL)
        eng.step_dead(W * NL)
        for r in range(W):  # (stands in for the all-gather: every slice filled with this rank's rows)
            sl = slice(r * NL, (r + 1) * NL)
            x_all[sl].copy_(x)
            eng.copy_step_state(NL, g_all[sl], idx_all[sl], val_all[sl])
        eng.backward_begin_gathered(x_all, g_all, idx_all, val_all)
        eng.backward_rows(0, S)
        eng.backward_end()
        eng.step_tail(1e-4, 1.0, grad_scale=1.0 / W, trusted=True)

    out["sparse_total_ms"] = timed(sparse_step)

    def fwd_only():
        i = it[0] = it[0] + 1
        eng.step_forward(pool[(i % 16) * NL:(i % 16 + 1) * NL], training=True, n_rows_global=W * NL)
        eng.step_dead(W * NL)

    out["forward_2048_ms"] = timed(fwd_only)
    eng.close()
    # ---- dense exchange, compute only
    eng = SaeEngine(EngineConfig(d_model=D, d_sae=S, top_k=K, max_batch=NL, aux_dead_cap=4096), dev)
    eng.view("W_dec").copy_(Wd); eng.view("W_enc").copy_(Wd.t())

    def dense_step():
        i = it[0] = it[0] + 1
        eng.step_forward(pool[(i % 16) * NL:(i % 16 + 1) * NL], training=True, n_rows_global=W * NL)
        eng.step_dead(W * NL)
        eng.step_backward()
        eng.step_tail(1e-4, 1.0, grad_scale=1.0 / W)

    out["dense_replicated_tail_total_ms"] = timed(dense_step)

    def dense_no_tail():
        i = it[0] = it[0] + 1
        eng.step_forward(pool[(i % 16) * NL:(i % 16 + 1) * NL], training=True, n_rows_global=W * NL)
        eng.step_dead(W * NL)
        eng.step_backward()

    out["dense_forward_backward_ms"] = timed(dense_no_tail)
    out["note"] = ("one MI355X; sparse_total includes the 8 local copies that stand in for the all-gather (x: 8 x 8 MB, state: 8 x 8.4 MB "
                   "device-to-device); the collectives themselves are not in any of these numbers")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
