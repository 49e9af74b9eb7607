"""Data-parallel driver of the train step: one process per GPU, RCCL collectives via torch.distributed.

The reference has no distributed training; the step shards by batch rows:
parameters are replicated, each rank runs phases 1-3 on its rows, and exactly three quantities
cross ranks per step:

  * the per-latent "fired" flags (int32, d_sae)        -> all-reduce MAX, before the tracker update
  * the flat gradient buffer (fp32, 2*D*S + S + D)     -> all-reduce SUM, scaled by 1/world in the tail
  * nothing else: the clip norm is computed on the reduced gradient, so replicas stay bit-identical.

`dist` may be any object with torch.distributed's all_reduce/ReduceOp API (gloo on CPU in tests).
"""

from __future__ import annotations

import torch


class DataParallelStepper:
    def __init__(self, engine, dist=None, world_size: int = 1, force: bool = False):
        """``force`` keeps the collective path even for one rank (exercises RCCL on a single-GPU box)."""
        self.engine = engine
        self.dist = dist if (world_size > 1 or force) else None
        self.world = world_size

    def train_step(self, x_local: torch.Tensor, lr: float, max_norm: float = 1.0) -> None:
        eng = self.engine
        if self.dist is None:
            eng.train_step(x_local, lr, max_norm)
            return
        n_global = x_local.shape[0] * self.world  # equal shards by construction
        eng.step_forward(x_local, training=True, n_rows_global=n_global)
        self.dist.all_reduce(eng.fired, op=self.dist.ReduceOp.MAX)
        eng.step_dead(n_global)
        eng.step_backward()
        self.dist.all_reduce(eng.grads, op=self.dist.ReduceOp.SUM)
        eng.step_tail(lr, max_norm, grad_scale=1.0 / self.world)
