"""World-size-2 and -4 checks of the data-parallel step protocol (saev_amd/framework/ddp.py) on CPU with gloo.

The HIP engine needs a GPU, so each rank drives a test-local stand-in engine that implements the same
phase interface (step_forward / step_dead / step_backward / step_tail, .fired, .grads) with the CPU
oracle.  What is verified is the protocol itself: which buffers cross ranks, with which reduction, in
which order, and the 1/world gradient scale -- i.e. that 2 ranks x B/2 rows reproduce the single-process
step on B rows, including the dead-latent tracker and AuxK."""

import math
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import sae_ref as R
from saev_amd.framework.ddp import DataParallelStepper


class OracleEngine:
    """Phase-split restatement of R.train_step for ONE rank's rows.  Parameters, Adam moments and gradients live in flat
    buffers laid out by saev_layout (the product's own layout function; with shard_world > 1 it pads the two halves to
    equal per-rank chunks), and state.params / m / v are views into them -- so the sharded tail's collectives run on the
    same memory pattern as on the GPU."""

    def __init__(self, params, cfg: R.RefConfig, shard_world: int = 1):
        from saev_amd.engine import EngineConfig, flat_layout

        self.cfg = cfg
        self.ecfg = EngineConfig(d_model=cfg.d_model, d_sae=cfg.d_sae, shard_world=shard_world)
        lay = flat_layout(self.ecfg)
        self.chunk_a, self.chunk_b, self.n_total = lay.chunk_a, lay.chunk_b, lay.n_total
        self._off = {"W_dec": lay.off_W_dec, "b_dec": lay.off_b_dec, "W_enc": lay.off_W_enc, "b_enc": lay.off_b_enc}
        self.params, self.adam_m, self.adam_v = (torch.zeros(self.n_total) for _ in range(3))
        self.grads = torch.zeros(self.n_total)

        def views(flat):
            return {k: flat[self._off[k] : self._off[k] + params[k].numel()].view(params[k].shape) for k in R.PARAM_ORDER}

        self.state = R.TrainState(params=views(self.params), m=views(self.adam_m), v=views(self.adam_v),
                                  toks_since_active=torch.zeros(cfg.d_sae, dtype=torch.int64))
        for k in R.PARAM_ORDER:
            self.state.params[k].copy_(params[k])
        self.fired = torch.zeros(cfg.d_sae, dtype=torch.int32)
        self.sumsq = torch.zeros(1, dtype=torch.float64)
        self.shard_world = shard_world
        self.calls = []

    def step_forward(self, x, *, training=True, n_rows_global=None):
        self.calls.append("forward")
        P = self.state.params
        if self.cfg.normalize_w_dec:
            P["W_dec"].copy_(R.normalize_w_dec(P["W_dec"]))
        self.leaves = {k: P[k].detach().clone().requires_grad_(True) for k in R.PARAM_ORDER}
        self.x = x
        self.h = R.encode_pre(x, self.leaves["W_enc"], self.leaves["b_enc"])
        self.f = R.topk_activation(self.h, self.cfg.top_k)
        self.fired.copy_((self.f.detach().abs() > 0).any(dim=0).to(torch.int32))

    def step_dead(self, n_rows_global):
        self.calls.append("dead")
        t = self.state.toks_since_active
        t += n_rows_global
        t[self.fired > 0] = 0
        self.fired.zero_()
        dead = t >= self.cfg.dead_threshold_tokens
        x_hats = R.decode(self.f, self.leaves["W_dec"], self.leaves["b_dec"])
        x_hats.retain_grad()
        mse = R.mean_squared_err(x_hats, self.x[:, None, :]).mean()
        aux = R.auxk_loss(x=self.x, h=self.h, x_hat_last=x_hats[:, -1], dead_mask=dead, W_dec=self.leaves["W_dec"],
                          b_dec=self.leaves["b_dec"], k_aux=self.cfg.k_aux, alpha=self.cfg.alpha)
        self.loss, self.mse, self.n_dead = mse + aux, mse.item(), int(dead.sum())
        self._mse_t, self._aux_t, self._x_hats, self._dead = mse, aux, x_hats, dead

    # ---- gathered backward (exchange="sparse"): what crosses ranks is x, dL/dx_hat and the codes; every rank forms the main
    # path's gradient of ALL rows from them with plain algebra, the auxiliary term's gradient stays local and travels as
    # the dead latents' compact rows ----
    def gather_buffers(self, world, n_local):
        n, D, K = world * n_local, self.cfg.d_model, self.cfg.top_k
        return torch.empty(n, D), torch.empty(n, D), torch.empty(n, K, dtype=torch.int32), torch.empty(n, K)

    def copy_step_state(self, n_rows, g_out, idx_out, val_out):
        self.calls.append("copy_state")
        (g,) = torch.autograd.grad(self._mse_t, self._x_hats, retain_graph=True)
        idx = torch.topk(self.h.detach(), self.cfg.top_k, dim=-1).indices.sort(dim=-1).values
        g_out.copy_(g[:, -1])
        idx_out.copy_(idx.to(torch.int32))
        val_out.copy_(self.h.detach().gather(1, idx))

    def backward_begin_gathered(self, x_all, g_all, idx_all, val_all):
        self.calls.append("backward_begin_gathered")
        S = self.cfg.d_sae
        W_dec = self.leaves["W_dec"].detach()
        idx = idx_all.long()
        F = torch.zeros(x_all.shape[0], S).scatter_(1, idx, val_all)
        mask = torch.zeros(x_all.shape[0], S).scatter_(1, idx, 1.0)
        dF = (g_all @ W_dec.T) * mask  # straight through the kept entries
        self._g = {"W_dec": F.T @ g_all, "b_dec": g_all.sum(dim=0), "W_enc": x_all.T @ dF, "b_enc": dF.sum(dim=0)}
        self._aux_local = None
        if self.n_dead > 0:
            ga = torch.autograd.grad(self._aux_t, [self.leaves[k] for k in R.PARAM_ORDER], allow_unused=True)
            ga = {k: (torch.zeros_like(self.leaves[k]) if v is None else v) for k, v in zip(R.PARAM_ORDER, ga)}
            dl = self._dead.nonzero().flatten()
            # (only the dead latents' rows / columns and b_dec receive anything from the auxiliary term)
            live = torch.ones(S, dtype=torch.bool); live[dl] = False
            assert ga["W_dec"][live].abs().max() == 0 and ga["W_enc"][:, live].abs().max() == 0 and ga["b_enc"][live].abs().max() == 0
            self._aux_local = (dl, torch.cat([ga["W_dec"][dl].reshape(-1), ga["W_enc"][:, dl].T.reshape(-1), ga["b_enc"][dl], ga["b_dec"]]))
        self.view("b_dec").copy_(self._g["b_dec"])

    def aux_compact_export(self):
        return None if self._aux_local is None else self._aux_local[1].clone()

    def aux_compact_import(self, buf):
        self.calls.append("aux_import")
        dl, D = self._aux_local[0], self.cfg.d_model
        nd = len(dl)
        self._g["W_dec"][dl] += buf[: nd * D].view(nd, D)
        self._g["W_enc"][:, dl] += buf[nd * D : 2 * nd * D].view(nd, D).T
        self._g["b_enc"][dl] += buf[2 * nd * D : 2 * nd * D + nd]
        self._g["b_dec"] += buf[2 * nd * D + nd :]
        self.view("b_dec").copy_(self._g["b_dec"])

    def step_backward(self):
        self.calls.append("backward")
        self.loss.backward()
        self.grads.zero_()
        for k in R.PARAM_ORDER:
            g = self.leaves[k].grad
            if g is not None:
                self.view(k).copy_(g)

    # ---- backward in latent ranges (what the overlapped exchange drives) ----
    @property
    def offsets(self):
        return self._off

    def view(self, name, flat=None):
        flat = self.grads if flat is None else flat
        p = self.state.params[name]
        return flat[self.offsets[name] : self.offsets[name] + p.numel()].view(p.shape)

    def grad_w_enc_t(self):
        if not hasattr(self, "_w_enc_t"):
            self._w_enc_t = torch.zeros(self.cfg.d_sae, self.cfg.d_model)
        return self._w_enc_t

    def backward_begin(self):
        self.calls.append("backward_begin")
        self.loss.backward()
        self._g = {k: (torch.zeros_like(self.leaves[k]) if self.leaves[k].grad is None else self.leaves[k].grad)
                   for k in R.PARAM_ORDER}
        self.view("b_dec").copy_(self._g["b_dec"])

    def backward_rows(self, lo, hi, part=0):
        self.calls.append(f"rows[{lo}:{hi}]" + (f"/{part}" if part else ""))
        if part != 2:
            self.view("W_dec")[lo:hi] = self._g["W_dec"][lo:hi]
        if part != 1:
            self.grad_w_enc_t()[lo:hi] = self._g["W_enc"].T[lo:hi]
            self.view("b_enc")[lo:hi] = self._g["b_enc"][lo:hi]

    def backward_end(self):
        self.calls.append("backward_end")
        self.view("W_enc").copy_(self.grad_w_enc_t().T)

    # ---- tail: whole (shard_rank < 0) or this rank's chunk of each half, as saev_tail_prepare / saev_tail_apply ----
    def halves(self, flat):
        a = self.chunk_a * self.ecfg.shard_world
        return flat[:a], flat[a:]

    def _ranges(self, r):
        if r < 0:
            a = self.chunk_a * self.ecfg.shard_world
            return (0, a), (a, self.n_total)
        a0 = self.chunk_a * self.ecfg.shard_world
        return (r * self.chunk_a, (r + 1) * self.chunk_a), (a0 + r * self.chunk_b, a0 + (r + 1) * self.chunk_b)

    def tail_prepare(self, shard_rank=-1):
        self.calls.append(f"prepare[{shard_rank}]")
        (a_lo, a_hi), (b_lo, b_hi) = self._ranges(shard_rank)
        S, D = self.cfg.d_sae, self.cfg.d_model
        r0, r1 = min(a_lo // D, S), min(a_hi // D, S)
        if self.cfg.remove_parallel_grads and r1 > r0:
            g = self.view("W_dec")[r0:r1]
            g.copy_(R.remove_parallel_grads(g, self.state.params["W_dec"][r0:r1]))
        sq = self.grads[a_lo:a_hi].double().pow(2).sum() + self.grads[b_lo:b_hi].double().pow(2).sum()
        self.sumsq[0] = sq

    def tail_apply(self, lr, max_norm=1.0, grad_scale=1.0, shard_rank=-1):
        self.calls.append(f"apply[{shard_rank}]")
        total = grad_scale * float(self.sumsq[0].sqrt())
        self.grad_norm = total
        coef = min(max_norm / (total + 1e-6), 1.0)
        self.state.adam_steps += 1
        for lo, hi in self._ranges(shard_rank):
            R.adam_update(self.params[lo:hi], self.grads[lo:hi] * (grad_scale * coef), self.adam_m[lo:hi], self.adam_v[lo:hi],
                          self.state.adam_steps, lr)

    def wenc_ready_after(self, event):
        self.calls.append("wenc_ready_after")

    def wdec_ready_after(self, event):
        self.calls.append("wdec_ready_after")

    def step_tail(self, lr, max_norm=1.0, grad_scale=1.0, trusted=False):
        self.calls.append("tail")
        self.tail_prepare(-1)
        self.tail_apply(lr, max_norm, grad_scale, -1)

    def train_step(self, x, lr, max_norm=1.0):
        self.step_forward(x)
        self.step_dead(x.shape[0])
        self.step_backward()
        self.step_tail(lr, max_norm, 1.0)


def _problem():
    cfg = R.RefConfig(d_model=32, d_sae=256, top_k=8, k_aux=16, dead_threshold_tokens=192, lr=2e-3, n_lr_warmup=2)
    g = torch.Generator().manual_seed(5)
    params = R.init_params(cfg, g)
    params["b_enc"] = 0.02 * torch.randn(cfg.d_sae, generator=g)
    A = torch.randn(32, 64, generator=g)
    batches = []
    for _ in range(6):
        s = torch.zeros(64, 64)
        for i in range(64):
            s[i, torch.randperm(64, generator=g)[:6]] = torch.rand(6, generator=g) + 0.5
        batches.append(s @ A.T / 4 + 0.05 * torch.randn(64, 32, generator=g))
    return cfg, params, batches


def _worker(rank, world, port, out, overlap=False, tail="replicated", exchange="dense"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg, params, batches = _problem()
    eng = OracleEngine(params, cfg, shard_world=world if tail == "sharded" else 1)
    stepper = DataParallelStepper(eng, dist, world, overlap=overlap, n_buckets=3, tail=tail, exchange=exchange)
    sched = R.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, len(batches), 0.0)
    lr, dead_counts = 0.0, []
    for x in batches:
        local = x[rank::world].contiguous()
        stepper.train_step(local, lr, cfg.grad_clip)
        dead_counts.append(eng.n_dead)
        lr = sched.step()
    if rank == 0:
        if exchange == "sparse":
            # (no gradient crosses ranks: state gathered, backward over all rows, compact auxiliary rows summed, whole tail)
            first = eng.calls[: eng.calls.index("tail") + 1]
            assert first in (["forward", "dead", "copy_state", "backward_begin_gathered", "rows[0:256]", "backward_end", "tail"],
                             ["forward", "dead", "copy_state", "backward_begin_gathered", "aux_import", "rows[0:256]", "backward_end", "tail"])
            assert "aux_import" in eng.calls and "backward" not in eng.calls
        elif overlap:
            assert eng.calls[:7] == ["forward", "dead", "backward_begin", "rows[0:85]", "rows[85:170]", "rows[170:256]",
                                     "backward_end"] and eng.calls[7] == "tail"
        elif tail == "sharded":
            # (backward in two passes: the decoder half is exchanged while the encoder pass runs)
            assert eng.calls[:8] == ["forward", "dead", "backward_begin", "rows[0:256]/1", "rows[0:256]/2", "backward_end",
                                     "prepare[0]", "apply[0]"] and "tail" not in eng.calls
        else:
            assert eng.calls[:4] == ["forward", "dead", "backward", "tail"]
    # padding of a sharded layout stays zero
    pad = torch.ones(eng.n_total, dtype=torch.bool)
    for k in R.PARAM_ORDER:
        pad[eng.offsets[k] : eng.offsets[k] + eng.state.params[k].numel()] = False
    assert (eng.params[pad] == 0).all() and (eng.adam_v[pad] == 0).all()
    torch.save({"params": {k: v.clone() for k, v in eng.state.params.items()}, "toks": eng.state.toks_since_active,
                "n_dead": dead_counts}, out.format(rank=rank))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,overlap,tail,exchange", [(2, False, "replicated", "dense"), (2, True, "replicated", "dense"),
                                                         (2, False, "sharded", "dense"), (4, False, "sharded", "dense"),
                                                         (4, False, "replicated", "dense"), (2, False, "replicated", "sparse"),
                                                         (4, False, "replicated", "sparse"), (8, False, "sharded", "dense"),
                                                         (8, False, "replicated", "dense"), (8, False, "replicated", "sparse")])
def test_ranks_reproduce_single_process_step(tmp_path, world, overlap, tail, exchange):
    """2 (4, 8 -- the node size north_star names) ranks x B/2 (B/4, B/8) rows == one process on B rows, for the all-reduce exchange (flat and bucketed / overlapped),
    for the sharded tail (reduce-scatter -> tail on 1/world of the elements -> all-gather) and for the sparse-state exchange
    (all-gather of x / dL/dx_hat / codes, full backward on every rank, the auxiliary term's compact rows summed)."""
    out = str(tmp_path / "rank{rank}.pt")
    mp.spawn(_worker, args=(world, _free_port(), out, overlap, tail, exchange), nprocs=world, join=True)
    rs = [torch.load(out.format(rank=r)) for r in range(world)]
    r0 = rs[0]
    # replicas stay bit-identical
    for other in rs[1:]:
        for k in R.PARAM_ORDER:
            assert torch.equal(r0["params"][k], other["params"][k]), k
        assert torch.equal(r0["toks"], other["toks"]) and r0["n_dead"] == other["n_dead"]
    # and match the single-process step on the full batches
    cfg, params, batches = _problem()
    single = OracleEngine(params, cfg)
    sched = R.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, len(batches), 0.0)
    lr, dead_counts = 0.0, []
    for x in batches:
        single.train_step(x, lr, cfg.grad_clip)
        dead_counts.append(single.n_dead)
        lr = sched.step()
    assert dead_counts == r0["n_dead"] and max(dead_counts) > 0, dead_counts
    assert torch.equal(single.state.toks_since_active, r0["toks"])
    for k in R.PARAM_ORDER:
        torch.testing.assert_close(r0["params"][k], single.state.params[k], rtol=1e-4, atol=1e-6)
    # the stand-in itself equals the oracle's monolithic train_step
    ref_state = R.TrainState.create(params)
    sched = R.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, len(batches), 0.0)
    for x in batches:
        R.train_step(ref_state, x, cfg, sched)
    for k in R.PARAM_ORDER:
        torch.testing.assert_close(single.state.params[k], ref_state.params[k], rtol=1e-5, atol=1e-7)


# ---- failure policy ("rank-failure = abort") -------------------------------------------------------------


def test_watchdog_names_the_last_collective_of_a_stalled_step():
    """A step that makes no progress for `timeout_s` is reported with rank, step and the last collective enqueued (the
    product ends the process there; the test swaps the exit for a callback)."""
    import time

    from saev_amd.framework.ddp import CollectiveWatchdog

    seen = []
    wd = CollectiveWatchdog(rank=3, timeout_s=0.2, poll_s=0.05, on_stall=seen.append)
    try:
        wd.begin_step(7)
        wd.enter("reduce_scatter(decoder half of the gradient)")
        wd.leave()
        time.sleep(0.1)
        assert seen == [], "inside the budget: nothing reported"
        wd.enter("all_reduce(sum of squares for the clip norm)")
        time.sleep(0.6)  # ... and the step never ends
        assert seen == ["all_reduce(sum of squares for the clip norm)"]
        # a finished step is never reported, however long the pause between steps
        wd.begin_step(8)
        wd.end_step()
        time.sleep(0.5)
        assert len(seen) == 1
    finally:
        wd.close()


def _failing_worker(rank, world, port, out):
    """Rank 1 dies before its third step; rank 0 must not hang: the stepper's watchdog ends it with exit code 13."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    torch.set_num_threads(1)
    cfg, params, batches = _problem()
    eng = OracleEngine(params, cfg)
    stepper = DataParallelStepper(eng, dist, world, timeout_s=2.0)
    for i, x in enumerate(batches):
        if rank == 1 and i == 2:
            os._exit(0)  # a crashed peer: no destroy_process_group, no goodbye
        stepper.train_step(x[rank::world].contiguous(), 1e-3, cfg.grad_clip)
        if rank == 0:
            with open(out, "a") as f:
                f.write(f"{i}\n")


@pytest.mark.timeout(120)
def test_a_dead_rank_aborts_the_survivor_instead_of_hanging_it(tmp_path):
    import time

    out = tmp_path / "progress.txt"
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, str(out))) for r in range(2)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(90)
    assert all(not p.is_alive() for p in procs), "the surviving rank hung"
    assert procs[1].exitcode == 0
    # rank 0 finished two steps, then either its watchdog (13) or gloo's own error on the broken pipe ended it -- never a clean 0
    assert out.read_text().split() == ["0", "1"]
    assert procs[0].exitcode not in (0, None), procs[0].exitcode
    assert time.time() - t0 < 90
