"""Data-parallel driver of the train step: one process per GPU, RCCL collectives via torch.distributed.

The reference has no distributed training; the step shards by batch rows:
parameters are replicated, each rank runs phases 1-3 on its rows, and exactly three quantities
cross ranks per step:

  * the per-latent "fired" flags (int32, d_sae)        -> all-reduce MAX, before the tracker update
  * the gradients (fp32, 2*D*S + S + D)                -> all-reduce SUM, scaled by 1/world in the tail
  * nothing else: the clip norm is computed on the reduced gradient, so replicas stay bit-identical.

Gradient exchange, two ways:

  * ``overlap=False`` (default): one all-reduce of the flat gradient buffer after the backward;
  * ``overlap=True`` (``SAEV_AMD_DDP_OVERLAP=1``): the backward runs in ``n_buckets`` latent ranges
    (saev_backward_rows); as soon as a range is done, the matching rows of dW_dec and of the transposed
    W_enc gradient (both contiguous) are all-reduced asynchronously while the next range is computed; the
    transposed gradient is turned into the (D, S) layout after it has been reduced (a transpose is linear).
    xGMI is point-to-point, so the exchange is long (hundreds of MB per step) and worth hiding -- but the backward is
    only ~0.85 ms of the step and the extra collectives cost latency; on one rank the bucketed path is 0.4 ms slower.
    It is parity-tested (gloo with two ranks, RCCL with one) and left opt-in until it can be measured on a multi-GPU
    node.

`dist` may be any object with torch.distributed's all_reduce/ReduceOp API (gloo on CPU in tests).
"""

from __future__ import annotations

import os

import torch


class DataParallelStepper:
    def __init__(self, engine, dist=None, world_size: int = 1, force: bool = False, overlap: bool | None = None,
                 n_buckets: int = 2):
        """``force`` keeps the collective path even for one rank (exercises RCCL on a single-GPU box)."""
        self.engine = engine
        self.dist = dist if (world_size > 1 or force) else None
        self.world = world_size
        if overlap is None:
            overlap = os.environ.get("SAEV_AMD_DDP_OVERLAP", "0") == "1"
        self.overlap = overlap and hasattr(engine, "backward_rows")
        self.n_buckets = max(1, n_buckets)

    def _exchange_overlapped(self) -> None:
        eng, dist = self.engine, self.dist
        S = eng.cfg.d_sae
        D = eng.cfg.d_model
        g_dec = eng.view("W_dec", eng.grads)          # (S, D) rows
        g_enc_t = eng.grad_w_enc_t()                  # (S, D) rows, transposed W_enc gradient
        eng.backward_begin()
        works = []
        bounds = [S * i // self.n_buckets for i in range(self.n_buckets + 1)]
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            if hi <= lo:
                continue
            eng.backward_rows(lo, hi)
            works.append(dist.all_reduce(g_dec[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            works.append(dist.all_reduce(g_enc_t[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        works.append(dist.all_reduce(eng.view("b_dec", eng.grads), op=dist.ReduceOp.SUM, async_op=True))
        works.append(dist.all_reduce(eng.view("b_enc", eng.grads), op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            w.wait()
        eng.backward_end()  # reduced transposed gradient -> W_enc segment of the flat buffer

    def train_step(self, x_local: torch.Tensor, lr: float, max_norm: float = 1.0, pre_tail=None) -> None:
        """One optimizer step.  ``pre_tail`` (log steps) is called after the backward and before rpg / clip / Adam --
        the point where the reference's log block looks at the parameters (train.py:365-442 sits between
        ``clip_grad_norm_`` and ``opt.step()``)."""
        eng = self.engine
        if self.dist is None:
            if pre_tail is None:
                eng.train_step(x_local, lr, max_norm)
                return
            n = x_local.shape[0]
            eng.step_forward(x_local, training=True, n_rows_global=n)
            eng.step_dead(n)
            eng.step_backward()
            pre_tail()
            eng.step_tail(lr, max_norm)
            return
        n_global = x_local.shape[0] * self.world  # equal shards by construction (data.ShuffledDataLoader.n_epoch)
        eng.step_forward(x_local, training=True, n_rows_global=n_global)
        self.dist.all_reduce(eng.fired, op=self.dist.ReduceOp.MAX)
        eng.step_dead(n_global)
        if self.overlap:
            self._exchange_overlapped()
        else:
            eng.step_backward()
            self.dist.all_reduce(eng.grads, op=self.dist.ReduceOp.SUM)
        if pre_tail is not None:
            pre_tail()
        eng.step_tail(lr, max_norm, grad_scale=1.0 / self.world)
