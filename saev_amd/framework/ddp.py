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

Tail, two ways (``tail=``, ``SAEV_AMD_DDP_TAIL``):

  * ``"replicated"`` (default): the gradient is all-reduced and every rank runs the whole tail (rpg, clip norm, Adam:
    1.88 GB of streaming at configs[1]) -- replicas stay bit-identical by construction;
  * ``"sharded"``: the engine's flat buffers are laid out as two halves of ``world`` equal chunks
    (``EngineConfig.shard_world``).  The gradient halves are reduce-scattered (half the bytes of an all-reduce each
    way; the backward runs in two passes, decoder gradient first, so that the decoder half's reduce-scatter travels
    while the encoder gradient is still being formed), every rank projects / squares / Adam-updates only its own chunk of each half (1/world of the streaming), one
    double -- the sum of squares -- is all-reduced so that all ranks clip with the same global norm, and the parameter
    halves are all-gathered: the encoder half first, on the compute stream (the next forward starts with it), the decoder
    half on a side stream, waited for only right before the next step's decode (``saev_wdec_ready_event``), so the
    encoder hides it.  Same bytes on the wire as the all-reduce, less tail, part of the gather off the critical path.

Strong scaling (``exchange="sparse"``, ``SAEV_AMD_DDP_EXCHANGE=sparse``): when the GLOBAL batch is fixed and the ranks
split it, a rank's compute shrinks with 1/world while the 268 MB gradient exchange does not -- at configs[2]'s 2 048 rows
per rank the all-reduce (>= 0.66 ms over seven xGMI links) is longer than the rank's forward.  This mode exchanges what the
backward CONSUMES instead of what it produces: every rank all-gathers x, dL/dx_hat and the codes of its rows
((8 D + 8 k) bytes per row: 17 MB per rank at D = 1024, k = 32, 2 048 rows) and forms the full gradient of the global
batch itself -- redundantly, with deterministic kernels, hence bit-identically on every rank.  The auxiliary loss stays
local to a rank's rows; its gradient is a few compact rows (the dead latents'), summed with one small all-reduce.  No
gradient buffer crosses ranks, every rank runs the whole tail (and may use the fused one: nothing touches the gradient
between backward and tail).  What it costs: the sparse backward over the global batch on every rank (0.8 ms at 16 384
rows) -- worth it when that is less than the dense exchange, i.e. for small per-rank batches; weak scaling keeps "dense".

`dist` may be any object with torch.distributed's collectives / ReduceOp API (gloo on CPU in tests).
"""

from __future__ import annotations

import os

import torch


class DataParallelStepper:
    def __init__(self, engine, dist=None, world_size: int = 1, force: bool = False, overlap: bool | None = None,
                 n_buckets: int = 2, tail: str | None = None, rank: int | None = None, exchange: str | None = None):
        """``force`` keeps the collective path even for one rank (exercises RCCL on a single-GPU box)."""
        if exchange is None:
            exchange = os.environ.get("SAEV_AMD_DDP_EXCHANGE", "dense")
        if exchange not in ("dense", "sparse"):
            raise ValueError(f"exchange must be 'dense' or 'sparse', got {exchange!r}")
        self.engine = engine
        self.dist = dist if (world_size > 1 or force) else None
        self.world = world_size
        if overlap is None:
            overlap = os.environ.get("SAEV_AMD_DDP_OVERLAP", "0") == "1"
        self.overlap = overlap and hasattr(engine, "backward_rows")
        self.n_buckets = max(1, n_buckets)
        if tail is None:
            tail = os.environ.get("SAEV_AMD_DDP_TAIL", "replicated")
        if tail not in ("replicated", "sharded"):
            raise ValueError(f"tail must be 'replicated' or 'sharded', got {tail!r}")
        self.tail = tail if self.dist is not None else "replicated"
        self.rank = rank if rank is not None else (self.dist.get_rank() if self.dist is not None else 0)
        self._side = None
        self.exchange = exchange if self.dist is not None else "dense"
        if self.exchange == "sparse":
            if self.overlap or self.tail == "sharded":
                raise ValueError("exchange='sparse' moves no gradient between ranks: it goes with the replicated tail and no overlap")
            if not hasattr(engine, "backward_begin_gathered"):
                raise ValueError("exchange='sparse' needs an engine with the gathered backward")
        self.two_pass = os.environ.get("SAEV_AMD_DDP_TWO_PASS", "1") != "0"
        if self.tail == "sharded":
            if self.overlap:
                raise ValueError("the sharded tail reduce-scatters whole halves after the backward; overlap=True is the all-reduce variant")
            got = getattr(engine, "shard_world", None)
            if got != self.world:
                raise ValueError(f"tail='sharded' needs an engine laid out for {self.world} ranks (shard_world), got {got}")

    def _backward_sharded(self) -> None:
        """Backward in two passes with the exchange of the decoder half behind the second one.  The decoder pass leaves
        [W_dec | b_dec] final; its reduce-scatter (half of the gradient bytes) is issued at once and travels while the
        encoder pass (the other half of the backward's gather traffic) and the transpose run; the encoder half follows."""
        eng, dist, r = self.engine, self.dist, self.rank
        S = eng.cfg.d_sae
        g_a, g_b = eng.halves(eng.grads)  # [W_dec | b_dec | pad], [W_enc | b_enc | pad]: `world` equal chunks each
        ca, cb = g_a.numel() // self.world, g_b.numel() // self.world
        if not self.two_pass:  # SAEV_AMD_DDP_TWO_PASS=0: one-pass backward, then both halves (0.09 ms less compute, nothing hidden)
            eng.step_backward()
            dist.reduce_scatter_tensor(g_a[r * ca : (r + 1) * ca], g_a, op=dist.ReduceOp.SUM)
            dist.reduce_scatter_tensor(g_b[r * cb : (r + 1) * cb], g_b, op=dist.ReduceOp.SUM)
            return
        eng.backward_begin()
        eng.backward_rows(0, S, 1)
        w = dist.reduce_scatter_tensor(g_a[r * ca : (r + 1) * ca], g_a, op=dist.ReduceOp.SUM, async_op=True)
        eng.backward_rows(0, S, 2)
        eng.backward_end()
        dist.reduce_scatter_tensor(g_b[r * cb : (r + 1) * cb], g_b, op=dist.ReduceOp.SUM)
        w.wait()

    def _tail_sharded(self, lr: float, max_norm: float, pre_tail=None) -> None:
        eng, dist, r = self.engine, self.dist, self.rank
        if pre_tail is not None:
            pre_tail()
        eng.tail_prepare(r)                                   # rpg on my decoder rows, sum of squares of my chunks
        dist.all_reduce(eng.sumsq, op=dist.ReduceOp.SUM)      # one double: every rank clips with the global norm
        eng.tail_apply(lr, max_norm, 1.0 / self.world, r)     # Adam on my chunks
        p_a, p_b = eng.halves(eng.params)
        cb = p_b.numel() // self.world
        ca = p_a.numel() // self.world
        if p_a.is_cuda:
            import torch

            # both gathers on a side stream: the next forward prepares its batch meanwhile and waits for the encoder half
            # only before it reads W_enc, for the decoder half only before its decode
            if self._side is None:
                self._side = torch.cuda.Stream(device=p_a.device)
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                dist.all_gather_into_tensor(p_b, p_b[r * cb : (r + 1) * cb])
                ev_b = torch.cuda.Event()
                ev_b.record(self._side)
                dist.all_gather_into_tensor(p_a, p_a[r * ca : (r + 1) * ca])
                ev_a = torch.cuda.Event()
                ev_a.record(self._side)
            eng.wenc_ready_after(ev_b)
            eng.wdec_ready_after(ev_a)
        else:
            dist.all_gather_into_tensor(p_b, p_b[r * cb : (r + 1) * cb])
            dist.all_gather_into_tensor(p_a, p_a[r * ca : (r + 1) * ca])

    def sync_params(self) -> None:
        """Make the current stream wait for parameter halves still arriving on the side stream (sharded tail): call before
        reading the parameters outside the train step -- checkpoints, evaluation, tests."""
        if self._side is not None:
            import torch

            torch.cuda.current_stream().wait_stream(self._side)

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

    def _step_sparse(self, x_local: torch.Tensor, lr: float, max_norm: float, pre_tail) -> None:
        """The sparse-state exchange (module docstring): forward on this rank's rows, all-gather of x / dL/dx_hat / codes,
        backward over every rank's rows, the auxiliary term's compact rows summed, replicated tail."""
        eng, dist, r, w = self.engine, self.dist, self.rank, self.world
        n = x_local.shape[0]
        n_global = n * w
        eng.step_forward(x_local, training=True, n_rows_global=n_global)
        dist.all_reduce(eng.fired, op=dist.ReduceOp.MAX)
        eng.step_dead(n_global)
        x_all, g_all, idx_all, val_all = eng.gather_buffers(w, n)
        sl = slice(r * n, (r + 1) * n)
        x_all[sl].copy_(x_local)
        eng.copy_step_state(n, g_all[sl], idx_all[sl], val_all[sl])
        # (RCCL gathers in place; other backends -- gloo in the tests -- get an input that does not alias the output)
        in_place = getattr(dist, "get_backend", lambda: "nccl")() == "nccl"
        for buf in (x_all, g_all, idx_all, val_all):  # rank-major row order on every rank: identical pair lists, identical sums
            dist.all_gather_into_tensor(buf, buf[sl] if in_place else buf[sl].clone())
        eng.backward_begin_gathered(x_all, g_all, idx_all, val_all)
        aux = eng.aux_compact_export()
        if aux is not None:
            dist.all_reduce(aux, op=dist.ReduceOp.SUM)
            eng.aux_compact_import(aux)
        eng.backward_rows(0, eng.cfg.d_sae)
        eng.backward_end()
        if pre_tail is not None:
            pre_tail()
        # (with a log-step callback in between the caller may look at -- not write -- the gradient: still trusted)
        eng.step_tail(lr, max_norm, grad_scale=1.0 / w, trusted=True)

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
        if self.exchange == "sparse":
            self._step_sparse(x_local, lr, max_norm, pre_tail)
            return
        n_global = x_local.shape[0] * self.world  # equal shards by construction (data.ShuffledDataLoader.n_epoch)
        eng.step_forward(x_local, training=True, n_rows_global=n_global)
        self.dist.all_reduce(eng.fired, op=self.dist.ReduceOp.MAX)
        eng.step_dead(n_global)
        if self.overlap:
            self._exchange_overlapped()
        elif self.tail == "sharded":
            self._backward_sharded()
        else:
            eng.step_backward()
            self.dist.all_reduce(eng.grads, op=self.dist.ReduceOp.SUM)
        if self.tail == "sharded":
            self._tail_sharded(lr, max_norm, pre_tail)
            return
        if pre_tail is not None:
            pre_tail()
        eng.step_tail(lr, max_norm, grad_scale=1.0 / self.world)
