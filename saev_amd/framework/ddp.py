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

import datetime
import logging
import os
import sys
import threading
import time

import torch

logger = logging.getLogger("saev_amd.ddp")

# A collective that has not returned after this many seconds is a failed rank ("rank-failure = abort").
DEFAULT_TIMEOUT_S = float(os.environ.get("SAEV_AMD_DDP_TIMEOUT_S", "600"))


def init_distributed(backend: str = "nccl", *, rank: int | None = None, world_size: int | None = None, device=None,
                     timeout_s: float = DEFAULT_TIMEOUT_S):
    """``torch.distributed.init_process_group`` with this package's failure policy: every collective carries a timeout,
    RCCL errors and timeouts tear the process down instead of leaving the other ranks spinning in a kernel
    (``TORCH_NCCL_ASYNC_ERROR_HANDLING=1``), and a timed-out collective dumps the communicator state
    (``TORCH_NCCL_DUMP_ON_TIMEOUT``, ``TORCH_NCCL_DESYNC_DEBUG``) so that the log names the rank that never arrived.
    Environment set by the launcher wins (``setdefault``).  Returns ``torch.distributed``."""
    import torch.distributed as dist

    for k, v in (("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1"), ("TORCH_NCCL_DUMP_ON_TIMEOUT", "1"), ("TORCH_NCCL_DESYNC_DEBUG", "1"),
                 ("HSA_ENABLE_IPC_MODE_LEGACY", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29531")):
        os.environ.setdefault(k, v)
    if dist.is_initialized():
        return dist
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
    kw = {"timeout": datetime.timedelta(seconds=timeout_s)}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world_size, **kw)
    return dist


class CollectiveWatchdog:
    """Names the collective a rank is stuck behind and ends the process.

    RCCL's own watchdog aborts a communicator after the process-group timeout, but collectives are enqueued
    asynchronously: a rank whose peer died blocks later -- in the next step's event wait, in a read-back, or inside the
    enqueue of a later collective -- and what the log then shows is a HIP stream that never drains.  The stepper marks the
    start and end of every step and the name of every collective it enqueues (attribute writes, no locks); a daemon thread
    looks every few seconds and, when a step has made no such progress for ``timeout_s``, logs rank, step and the last
    collective enqueued and leaves with ``os._exit(13)`` -- torchrun then takes the other ranks down."""

    def __init__(self, rank: int, timeout_s: float = DEFAULT_TIMEOUT_S, poll_s: float = 2.0, on_stall=None):
        self.rank, self.timeout_s, self.poll_s = rank, timeout_s, min(poll_s, max(0.05, timeout_s / 4))
        self.last: str = "(none yet)"
        self.since = time.monotonic()
        self.step = 0
        self.active = False
        self.on_stall = on_stall  # tests replace the exit
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="saev-ddp-watchdog", daemon=True)
        self._thread.start()

    def begin_step(self, step: int) -> None:
        self.step, self.since, self.active = step, time.monotonic(), True

    def end_step(self) -> None:
        self.active = False

    def enter(self, what: str) -> None:
        self.since = time.monotonic()
        self.last = what

    def leave(self) -> None:
        self.since = time.monotonic()

    def close(self) -> None:
        self._stop.set()

    def _run(self) -> None:
        while not self._stop.wait(self.poll_s):
            if self.active and time.monotonic() - self.since > self.timeout_s:
                msg = (f"[saev_amd.ddp] rank {self.rank}: step {self.step} has made no progress for {self.timeout_s:.0f} s; last "
                       f"collective enqueued: '{self.last}' -- a peer rank failed or never reached it; aborting this rank")
                logger.error(msg)
                print(msg, file=sys.stderr, flush=True)
                if self.on_stall is not None:
                    self.on_stall(self.last)
                    self.active = False
                    continue
                os._exit(13)


class _SelfCheck:
    """The start-up self-check of choose_exchange in two phases, so that the ranks can agree between them: ``__init__`` only
    allocates (engines of a 64 x 512 SAE pair that shares its batches, as train() links every SAE after the first to the
    first; AuxK active from the third step) -- the one place a single rank can fail on its own (out of memory, a HIP error)
    -- and ``run`` steps them through DataParallelStepper(**stepper_kw), which is where the collectives are."""

    N_ROWS = 128

    def __init__(self, stepper_kw: dict, dist, world: int, rank: int, device, shard_world: int, global_rows: bool, n_saes: int = 2):
        from ..engine import EngineConfig, SaeEngine

        self.kw, self.dist, self.world, self.rank, self.device = stepper_kw, dist, world, rank, device
        self.engines = []
        try:
            n = self.N_ROWS
            gg = torch.Generator(device=device).manual_seed(3)
            W0 = torch.randn(512, 64, device=device, generator=gg)
            W0 /= W0.norm(dim=1, keepdim=True)
            for j in range(n_saes):
                e = SaeEngine(EngineConfig(d_model=64, d_sae=512, top_k=8, k_aux=16, dead_threshold_tokens=256, max_batch=n,
                                           max_backward_rows=n * world if global_rows else 0, shard_world=shard_world), device)
                self.engines.append(e)
                Wj = W0.roll(5 * j, dims=0)
                e.view("W_dec").copy_(Wj)
                e.view("W_enc").copy_(Wj.t())
                if j > 0:
                    e.share_x(self.engines[0])
        except Exception:
            self.close()
            raise

    def run(self) -> dict:
        """Four steps; returns the parameters of every SAE.  Every rank draws its own rows.  Errors propagate: a rank that
        fails in here fails inside a sequence of collectives, and the only safe outcome is the process group's abort."""
        steppers = [DataParallelStepper(e, self.dist, self.world, force=True, rank=self.rank, **self.kw) for e in self.engines]
        try:
            gx = torch.Generator(device=self.device).manual_seed(100 + self.rank)
            for i in range(4):
                x = torch.randn(self.N_ROWS, 64, device=self.device, generator=gx)
                for st in steppers:
                    st.train_step(x, 1e-3 * i, 0.05)
            for st in steppers:
                st.sync_params()
            torch.cuda.synchronize(self.device)
            return {f"{j}.{k}": v.clone() for j, e in enumerate(self.engines) for k, v in e.param_views().items()}
        finally:
            for st in steppers:
                st.close()

    def close(self) -> None:
        for e in self.engines:
            e.close()
        self.engines = []


def choose_exchange(dist, world: int, rank: int, device, local_batch: int, *, tail: str = "auto", exchange: str = "auto",
                    sparse_max_rows: int = 4096) -> tuple[str, str, dict]:
    """Resolve ("auto" | explicit) tail / exchange settings into what a run uses, identically on every rank.

      exchange  "auto": the sparse step state when a rank holds at most ``sparse_max_rows`` rows (strong scaling: the
                268 MB gradient exchange would outlast the rank's compute), else the gradient ("dense"); with the tail pinned
                to "sharded" it resolves to "dense" (the sparse exchange moves no gradient, there is nothing to shard);
      tail      "auto" (dense exchange only): "sharded" -- reduce-scatter, 1/world of the tail, all-gather -- when a
                start-up self-check on a small pair of SAEs reproduces the all-reduce path's parameters on every rank and
                leaves all ranks with identical parameters; "replicated" if anything differs.  An automatically chosen sparse
                exchange gets the same check against the all-reduce path.  Explicit settings are honoured unchecked.

    The self-check is a fixed protocol every rank walks in step: (1) allocate -- errors caught locally, then one MIN
    all-reduce of a status flag; (2) step the all-reduce path and the candidate -- errors in here are not caught: a rank that
    fails between two collectives cannot be waited for, the process group's timeout / abort handling ends the job; (3) compare
    -- again a local verdict and one MIN all-reduce.  No rank ever enters a collective its peers may skip.
    Returns (tail, exchange, report); the report says what was checked and why a fallback was taken."""
    report: dict = {"requested": {"tail": tail, "exchange": exchange}}
    if dist is None:
        return "replicated", "dense", report
    if world <= 1:  # one rank over a real backend (bench.py --force-dist): explicit choices run as they are, "auto" has nothing to check
        return (tail if tail != "auto" else "replicated"), (exchange if exchange != "auto" else "dense"), report
    if exchange == "sparse" and tail not in ("auto", "replicated"):
        raise ValueError("exchange='sparse' goes with the replicated tail")
    if exchange == "sparse":  # pinned by the caller: honoured as it is
        return "replicated", "sparse", report
    if exchange == "auto":
        exchange = "sparse" if (local_batch <= sparse_max_rows and tail != "sharded") else "dense"
    if exchange == "dense" and tail != "auto":
        return tail, exchange, report

    def agree(ok: bool) -> bool:
        flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    candidate = {"tail": "sharded"} if exchange == "dense" else {"tail": "replicated", "exchange": "sparse"}
    checks, why = [], None
    try:  # phase 1: allocation only, no collective inside
        checks.append(_SelfCheck({"tail": "replicated", "exchange": "dense"}, dist, world, rank, device, 1, False))
        checks.append(_SelfCheck(dict({"exchange": "dense"}, **candidate), dist, world, rank, device,
                                 world if candidate["tail"] == "sharded" else 1, exchange == "sparse"))
        built = True
    except Exception as exc:
        built, why = False, f"allocation failed on rank {rank}: {type(exc).__name__}: {exc}"
        print(f"[saev_amd.ddp] self-check of {candidate}: {why}", file=sys.stderr, flush=True)
    try:
        if not agree(built):
            report["self_check"] = {"candidate": candidate, "passed": False, "why_not": why or "allocation failed on another rank"}
            return "replicated", "dense", report
        ref, got = checks[0].run(), checks[1].run()  # phase 2: every rank runs the same collectives or the job aborts
    finally:
        for ch in checks:
            ch.close()

    # phase 3.  Against the all-reduce path a tolerance, not equality: the routes add in different orders, and Adam's
    # m / sqrt(v) turns a noise-level gradient of either sign into a step of size lr -- so: nearly all elements within 1e-4
    # relative, none further apart than a few learning rates
    def close(a, b):
        d = (a - b).abs()
        return bool(((d > 1e-6 + 1e-4 * b.abs()).float().mean() < 1e-3) and d.max() < 0.02)

    ok = all(close(got[k], ref[k]) for k in ref)
    flat = torch.cat([v.reshape(-1) for v in got.values()]).clone()
    mine = flat.clone()
    dist.broadcast(flat, src=0)
    ok = ok and torch.equal(flat, mine)  # every rank must hold rank 0's parameters, bit for bit
    why = None if ok else "parameters differ from the all-reduce path or between ranks"
    ok = agree(ok)
    report["self_check"] = {"candidate": candidate, "passed": ok, "why_not": why}
    if ok:
        return candidate["tail"], exchange, report
    return "replicated", "dense", report


def collective_busbw(dist, world: int, device, n_params: int, chunk_a: int, chunk_b: int, sparse_bytes_per_rank: int = 0,
                     iters: int = 5) -> dict:
    """Wall time and RCCL bus bandwidth of a step's own collectives at their real sizes, on the live communicator: the flat
    all-reduce (replicated tail), the two reduce-scatters and all-gathers of the sharded tail (same in-place aliasing as the
    step), and the all-gather of the sparse step state.  busbw: algbw x 2 (n - 1) / n for the all-reduce, x (n - 1) / n for
    reduce-scatter / all-gather (the nccl-tests convention).  Every rank must call it."""
    out = {}
    r = dist.get_rank()
    flat = torch.zeros(max(n_params, world * (chunk_a + chunk_b)), device=device)

    def run(name, nbytes, factor, fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(device)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize(device)
        t = torch.tensor([(time.perf_counter() - t0) / iters], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        sec = t.item()
        out[name] = {"bytes": nbytes, "ms": sec * 1e3, "algbw_GBps": nbytes / sec / 1e9, "busbw_GBps": nbytes / sec / 1e9 * factor}

    n = world
    run("all_reduce_flat_gradient", 4 * n_params, 2 * (n - 1) / n, lambda: dist.all_reduce(flat[:n_params], op=dist.ReduceOp.SUM))
    a, b = flat[: n * chunk_a], flat[n * chunk_a : n * chunk_a + n * chunk_b]
    for name, buf, c in (("decoder_half", a, chunk_a), ("encoder_half", b, chunk_b)):
        mine = buf[r * c : (r + 1) * c]
        run(f"reduce_scatter_{name}", 4 * buf.numel(), (n - 1) / n, lambda buf=buf, mine=mine: dist.reduce_scatter_tensor(mine, buf, op=dist.ReduceOp.SUM))
        run(f"all_gather_{name}", 4 * buf.numel(), (n - 1) / n, lambda buf=buf, mine=mine: dist.all_gather_into_tensor(buf, mine))
    if sparse_bytes_per_rank > 0:
        m = sparse_bytes_per_rank // 4
        g = torch.zeros(n * m, device=device)
        run("all_gather_sparse_step_state", 4 * g.numel(), (n - 1) / n, lambda: dist.all_gather_into_tensor(g, g[r * m : (r + 1) * m]))
    return out


class DataParallelStepper:
    def __init__(self, engine, dist=None, world_size: int = 1, force: bool = False, overlap: bool | None = None,
                 n_buckets: int = 2, tail: str | None = None, rank: int | None = None, exchange: str | None = None,
                 timeout_s: float | None = None):
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
        self.check = os.environ.get("SAEV_AMD_DDP_CHECK", "0") == "1"  # debug: ranks compare their host-side route decisions
        self.watchdog = CollectiveWatchdog(self.rank, timeout_s if timeout_s is not None else DEFAULT_TIMEOUT_S) if self.dist is not None else None
        self.steps = 0
        if self.tail == "sharded":
            if self.overlap:
                raise ValueError("the sharded tail reduce-scatters whole halves after the backward; overlap=True is the all-reduce variant")
            got = getattr(engine, "shard_world", None)
            if got != self.world:
                raise ValueError(f"tail='sharded' needs an engine laid out for {self.world} ranks (shard_world), got {got}")

    def close(self) -> None:
        if self.watchdog is not None:
            self.watchdog.close()

    def _coll(self, what: str, fn, *args, **kw):
        """Run one collective under the watchdog's bracket (async ones: the bracket covers the enqueue; ``_wait`` the wait)."""
        wd = self.watchdog
        if wd is None:
            return fn(*args, **kw)
        wd.enter(what)
        try:
            return fn(*args, **kw)
        finally:
            wd.leave()

    def _wait(self, what: str, work) -> None:
        if work is not None:
            self._coll(f"wait({what})", work.wait)

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
            self._coll("reduce_scatter(decoder half of the gradient)", dist.reduce_scatter_tensor, g_a[r * ca : (r + 1) * ca], g_a, op=dist.ReduceOp.SUM)
            self._coll("reduce_scatter(encoder half of the gradient)", dist.reduce_scatter_tensor, g_b[r * cb : (r + 1) * cb], g_b, op=dist.ReduceOp.SUM)
            return
        eng.backward_begin()
        eng.backward_rows(0, S, 1)
        w = self._coll("reduce_scatter(decoder half of the gradient)", dist.reduce_scatter_tensor, g_a[r * ca : (r + 1) * ca], g_a,
                       op=dist.ReduceOp.SUM, async_op=True)
        eng.backward_rows(0, S, 2)
        eng.backward_end()
        self._coll("reduce_scatter(encoder half of the gradient)", dist.reduce_scatter_tensor, g_b[r * cb : (r + 1) * cb], g_b, op=dist.ReduceOp.SUM)
        self._wait("reduce_scatter(decoder half of the gradient)", w)

    def _tail_sharded(self, lr: float, max_norm: float, pre_tail=None) -> None:
        eng, dist, r = self.engine, self.dist, self.rank
        if pre_tail is not None:
            pre_tail()
        eng.tail_prepare(r)                                   # rpg on my decoder rows, sum of squares of my chunks
        self._coll("all_reduce(sum of squares for the clip norm)", dist.all_reduce, eng.sumsq, op=dist.ReduceOp.SUM)  # one double: every rank clips with the global norm
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
                self._coll("all_gather(encoder half of the parameters)", dist.all_gather_into_tensor, p_b, p_b[r * cb : (r + 1) * cb])
                ev_b = torch.cuda.Event()
                ev_b.record(self._side)
                self._coll("all_gather(decoder half of the parameters)", dist.all_gather_into_tensor, p_a, p_a[r * ca : (r + 1) * ca])
                ev_a = torch.cuda.Event()
                ev_a.record(self._side)
            eng.wenc_ready_after(ev_b)
            eng.wdec_ready_after(ev_a)
        else:
            self._coll("all_gather(encoder half of the parameters)", dist.all_gather_into_tensor, p_b, p_b[r * cb : (r + 1) * cb])
            self._coll("all_gather(decoder half of the parameters)", dist.all_gather_into_tensor, p_a, p_a[r * ca : (r + 1) * ca])

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
            works.append(self._coll(f"all_reduce(dW_dec rows {lo}:{hi})", dist.all_reduce, g_dec[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            works.append(self._coll(f"all_reduce(dW_enc^T rows {lo}:{hi})", dist.all_reduce, g_enc_t[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        works.append(self._coll("all_reduce(db_dec)", dist.all_reduce, eng.view("b_dec", eng.grads), op=dist.ReduceOp.SUM, async_op=True))
        works.append(self._coll("all_reduce(db_enc)", dist.all_reduce, eng.view("b_enc", eng.grads), op=dist.ReduceOp.SUM, async_op=True))
        for w in works:
            self._wait("bucketed gradient all-reduce", w)
        eng.backward_end()  # reduced transposed gradient -> W_enc segment of the flat buffer

    def _step_sparse(self, x_local: torch.Tensor, lr: float, max_norm: float, pre_tail) -> None:
        """The sparse-state exchange (module docstring): forward on this rank's rows, all-gather of x / dL/dx_hat / codes,
        backward over every rank's rows, the auxiliary term's compact rows summed, replicated tail."""
        eng, dist, r, w = self.engine, self.dist, self.rank, self.world
        n = x_local.shape[0]
        n_global = n * w
        eng.step_forward(x_local, training=True, n_rows_global=n_global)
        self._coll("all_reduce(fired flags, MAX)", dist.all_reduce, eng.fired, op=dist.ReduceOp.MAX)
        eng.step_dead(n_global)
        if self.check:
            self._check_same_route()
        x_all, g_all, idx_all, val_all = eng.gather_buffers(w, n)
        sl = slice(r * n, (r + 1) * n)
        x_all[sl].copy_(x_local)
        eng.copy_step_state(n, g_all[sl], idx_all[sl], val_all[sl])
        # (RCCL gathers in place; other backends -- gloo in the tests -- get an input that does not alias the output)
        in_place = getattr(dist, "get_backend", lambda: "nccl")() == "nccl"
        for name, buf in (("x", x_all), ("dL/dx_hat", g_all), ("code indices", idx_all), ("code values", val_all)):
            # rank-major row order on every rank: identical pair lists, identical sums
            self._coll(f"all_gather({name} of the step state)", dist.all_gather_into_tensor, buf, buf[sl] if in_place else buf[sl].clone())
        eng.backward_begin_gathered(x_all, g_all, idx_all, val_all)
        aux = eng.aux_compact_export()
        if aux is not None:
            # (every rank must take this branch with the same size: the route is a host-side decision from the tracker record,
            # which is identical on all ranks -- the fired flags were all-reduced; SAEV_AMD_DDP_CHECK=1 verifies it per step)
            self._coll("all_reduce(AuxK compact gradient rows)", dist.all_reduce, aux, op=dist.ReduceOp.SUM)
            eng.aux_compact_import(aux)
        eng.backward_rows(0, eng.cfg.d_sae)
        eng.backward_end()
        if pre_tail is not None:
            pre_tail()
        # The fused tail takes the clip norm and the projection from row statistics the backward left behind: only sound when
        # nothing wrote the gradient since.  A pre_tail hook is caller code (it may project or rescale eng.grads): with one,
        # the generic tail re-reads the gradient buffer as it now stands.
        eng.step_tail(lr, max_norm, grad_scale=1.0 / w, trusted=pre_tail is None)

    def _check_same_route(self) -> None:
        """Debug (SAEV_AMD_DDP_CHECK=1): every rank must have made the same host-side AuxK decision -- route and compact-row
        count size the auxiliary all-reduce.  One tiny MIN / MAX all-reduce per step; raises on the first disagreement."""
        eng = self.engine
        rows = int(eng.lib.saev_aux_compact_rows(eng.ctx)) if hasattr(eng, "lib") else 0
        v = torch.tensor([eng.aux_route() if hasattr(eng, "aux_route") else 0, rows], device=eng.fired.device, dtype=torch.int32)
        t = torch.stack([v, -v])
        self._coll("all_reduce(route check)", self.dist.all_reduce, t, op=self.dist.ReduceOp.MAX)
        if not torch.equal(t[0], -t[1]):
            raise RuntimeError(f"rank {self.rank}: data-parallel ranks disagree on the AuxK route / compact rows of step {self.steps}: "
                               f"mine {v.tolist()}, max {t[0].tolist()}, min {(-t[1]).tolist()}")

    def train_step(self, x_local: torch.Tensor, lr: float, max_norm: float = 1.0, pre_tail=None) -> None:
        """One optimizer step.  ``pre_tail`` (log steps) is called after the backward and before rpg / clip / Adam --
        the point where the reference's log block looks at the parameters (train.py:365-442 sits between
        ``clip_grad_norm_`` and ``opt.step()``).  It may READ the gradient; under ``exchange="sparse"`` a hook makes the step
        take the generic tail (which re-reads the gradient buffer), so a hook that writes the gradient is honoured too."""
        eng = self.engine
        self.steps += 1
        if self.watchdog is None:
            return self._train_step(x_local, lr, max_norm, pre_tail)
        self.watchdog.begin_step(self.steps)
        try:
            return self._train_step(x_local, lr, max_norm, pre_tail)
        finally:
            self.watchdog.end_step()

    def _train_step(self, x_local: torch.Tensor, lr: float, max_norm: float, pre_tail) -> None:
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
        self._coll("all_reduce(fired flags, MAX)", self.dist.all_reduce, eng.fired, op=self.dist.ReduceOp.MAX)
        eng.step_dead(n_global)
        if self.check:
            self._check_same_route()
        if self.overlap:
            self._exchange_overlapped()
        elif self.tail == "sharded":
            self._backward_sharded()
        else:
            eng.step_backward()
            self._coll("all_reduce(flat gradient)", self.dist.all_reduce, eng.grads, op=self.dist.ReduceOp.SUM)
        if self.tail == "sharded":
            self._tail_sharded(lr, max_norm, pre_tail)
            return
        if pre_tail is not None:
            pre_tail()
        eng.step_tail(lr, max_norm, grad_scale=1.0 / self.world)
