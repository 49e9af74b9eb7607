"""World-size-2 check of the data-parallel step protocol (saev_amd/framework/ddp.py) on CPU with gloo.

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
    """Phase-split restatement of R.train_step for ONE rank's rows."""

    def __init__(self, params, cfg: R.RefConfig):
        self.cfg = cfg
        self.state = R.TrainState.create(params)
        S = cfg.d_sae
        self.fired = torch.zeros(S, dtype=torch.int32)
        n = sum(p.numel() for p in params.values())
        self.grads = torch.zeros(n)
        self.calls = []

    def step_forward(self, x, *, training=True, n_rows_global=None):
        self.calls.append("forward")
        P = self.state.params
        if self.cfg.normalize_w_dec:
            P["W_dec"] = R.normalize_w_dec(P["W_dec"])
        self.leaves = {k: P[k].detach().requires_grad_(True) for k in R.PARAM_ORDER}
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
        mse = R.mean_squared_err(x_hats, self.x[:, None, :]).mean()
        aux = R.auxk_loss(x=self.x, h=self.h, x_hat_last=x_hats[:, -1], dead_mask=dead, W_dec=self.leaves["W_dec"],
                          b_dec=self.leaves["b_dec"], k_aux=self.cfg.k_aux, alpha=self.cfg.alpha)
        self.loss, self.mse, self.n_dead = mse + aux, mse.item(), int(dead.sum())

    def step_backward(self):
        self.calls.append("backward")
        self.loss.backward()
        off = 0
        for k in R.PARAM_ORDER:
            g = self.leaves[k].grad
            g = torch.zeros_like(self.leaves[k]) if g is None else g
            self.grads[off : off + g.numel()] = g.reshape(-1)
            off += g.numel()

    # ---- backward in latent ranges (what the overlapped exchange drives) ----
    @property
    def offsets(self):
        off, out = 0, {}
        for k in R.PARAM_ORDER:
            out[k] = off
            off += self.state.params[k].numel()
        return out

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

    def backward_rows(self, lo, hi):
        self.calls.append(f"rows[{lo}:{hi}]")
        self.view("W_dec")[lo:hi] = self._g["W_dec"][lo:hi]
        self.grad_w_enc_t()[lo:hi] = self._g["W_enc"].T[lo:hi]
        self.view("b_enc")[lo:hi] = self._g["b_enc"][lo:hi]

    def backward_end(self):
        self.calls.append("backward_end")
        self.view("W_enc").copy_(self.grad_w_enc_t().T)

    def step_tail(self, lr, max_norm=1.0, grad_scale=1.0):
        self.calls.append("tail")
        P = self.state.params
        grads, off = {}, 0
        for k in R.PARAM_ORDER:
            n = P[k].numel()
            grads[k] = (self.grads[off : off + n] * grad_scale).view_as(P[k]).clone()
            off += n
        if self.cfg.remove_parallel_grads:
            grads["W_dec"] = R.remove_parallel_grads(grads["W_dec"], P["W_dec"])
        clipped, self.grad_norm = R.clip_grad_norm([grads[k] for k in R.PARAM_ORDER], max_norm)
        self.state.adam_steps += 1
        for k, g in zip(R.PARAM_ORDER, clipped):
            R.adam_update(P[k], g, self.state.m[k], self.state.v[k], self.state.adam_steps, lr)

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


def _worker(rank, world, port, out, overlap=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    cfg, params, batches = _problem()
    eng = OracleEngine(params, cfg)
    stepper = DataParallelStepper(eng, dist, world, overlap=overlap, n_buckets=3)
    sched = R.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, len(batches), 0.0)
    lr, dead_counts = 0.0, []
    for x in batches:
        local = x[rank::world].contiguous()
        stepper.train_step(local, lr, cfg.grad_clip)
        dead_counts.append(eng.n_dead)
        lr = sched.step()
    if rank == 0:
        if overlap:
            assert eng.calls[:7] == ["forward", "dead", "backward_begin", "rows[0:85]", "rows[85:170]", "rows[170:256]",
                                     "backward_end"] and eng.calls[7] == "tail"
        else:
            assert eng.calls[:4] == ["forward", "dead", "backward", "tail"]
    torch.save({"params": eng.state.params, "toks": eng.state.toks_since_active, "n_dead": dead_counts}, out.format(rank=rank))
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(180)
@pytest.mark.parametrize("overlap", [False, True])
def test_two_ranks_reproduce_single_process_step(tmp_path, overlap):
    world = 2
    out = str(tmp_path / "rank{rank}.pt")
    mp.spawn(_worker, args=(world, _free_port(), out, overlap), nprocs=world, join=True)
    r0, r1 = (torch.load(out.format(rank=r)) for r in range(world))
    # replicas stay bit-identical
    for k in R.PARAM_ORDER:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
    assert torch.equal(r0["toks"], r1["toks"]) and r0["n_dead"] == r1["n_dead"]
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
