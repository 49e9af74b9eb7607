"""CPU oracle for the TopK-SAE train step (TEST INFRASTRUCTURE — not the product).

A plain PyTorch-CPU restatement of the algorithm the reference runs on its hot path
(``/root/reference/src/saev/nn/modeling.py``, ``nn/objectives.py``, ``framework/train.py``,
``utils/scheduling.py``).  It exists only so that

  * ``tests/``  can check the HIP path against it,
  * ``__graft_entry__.smoke()`` can check one small invocation, and
  * ``bench.py``'s ``cpu_baseline`` leg can time it on the host cores.

Nothing in ``saev_amd/`` may import this file.  The oracle is *pinned*: ``oracle/gen_golden.py``
imports the real reference (in the build container only) and writes ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against those vectors, and
``tests/test_oracle_known_answers.py`` restates the reference's own known-answer tests.

The arithmetic deliberately follows the reference's *dense* formulation (dense decode GEMM, autograd
backward) so that the CPU baseline it provides costs what the reference costs.

All tensors are fp32 on CPU unless stated.  Shapes: B batch, D d_model, S d_sae, P n_prefixes.
"""

from __future__ import annotations

import dataclasses
import math
from typing import Iterable, Iterator

import torch
from torch import Tensor

# --------------------------------------------------------------------------------------------
# Configuration (mirrors the fields the hot path reads; reference modeling.py:66-73,119-130,
# 259-284; objectives.py:13-25; train.py:50-105).
# --------------------------------------------------------------------------------------------


@dataclasses.dataclass(frozen=True)
class RefConfig:
    d_model: int = 1024
    d_sae: int = 16384
    top_k: int = 32
    k_aux: int = 512
    alpha: float = 1.0 / 32.0
    use_aux: bool = True
    normalize_w_dec: bool = True
    remove_parallel_grads: bool = True
    n_prefixes: int = 1
    dead_threshold_tokens: int = 10_000_000
    lr: float = 4e-4
    n_lr_warmup: int = 500
    grad_clip: float = 1.0
    # Not in the reference (it trains in fp32 / TF32 only): models BASELINE.json configs[3] "bf16" -- the encoder
    # contraction that feeds TopK sees bf16-rounded x and W_enc (fp32 accumulate); everything else -- the AuxK branch's
    # pre-activations of dead latents, decode, losses, all gradients, Adam -- is fp32.
    encoder_bf16: bool = False


PARAM_ORDER = ("W_dec", "b_dec", "W_enc", "b_enc")  # state_dict order, modeling.py:312-327


# --------------------------------------------------------------------------------------------
# Parameter init (modeling.py:306-329)
# --------------------------------------------------------------------------------------------


def init_params(cfg: RefConfig, generator: torch.Generator | None = None) -> dict[str, Tensor]:
    """Kaiming-uniform decoder rows, unit-normalised; encoder = decoder transposed (own storage);
    zero biases.  modeling.py:312-327.  (Bit-parity with the reference's init is *not* claimed —
    parity runs feed the reference's initial parameters in as a fixture.)"""
    S, D = cfg.d_sae, cfg.d_model
    # kaiming_uniform_(a=0, fan_in=D): gain = sqrt(2), bound = gain*sqrt(3/fan_in) = sqrt(6/D)
    bound = math.sqrt(6.0 / D)
    W_dec = (torch.rand(S, D, generator=generator) * 2 - 1) * bound
    if cfg.normalize_w_dec:
        W_dec = normalize_w_dec(W_dec)
    return {
        "W_dec": W_dec.contiguous(),
        "b_dec": torch.zeros(D),
        "W_enc": W_dec.t().contiguous().clone(),
        "b_enc": torch.zeros(S),
    }


# --------------------------------------------------------------------------------------------
# Per-op restatements
# --------------------------------------------------------------------------------------------


def normalize_w_dec(W_dec: Tensor) -> Tensor:
    """Each decoder row divided by its L2 norm, no epsilon.  modeling.py:411-417."""
    return W_dec / torch.norm(W_dec, dim=1, keepdim=True)


def encode_pre(x: Tensor, W_enc: Tensor, b_enc: Tensor) -> Tensor:
    """Pre-activations h = x @ W_enc + b_enc.  modeling.py:343-347."""
    return torch.einsum("bd,ds->bs", x, W_enc) + b_enc


def encode_pre_bf16(x: Tensor, W_enc: Tensor, b_enc: Tensor) -> Tensor:
    """Pre-activations with bf16-rounded operands (round to nearest even) and fp32 accumulation.  The value is the
    bf16 product; the gradient is that of the fp32 formula (straight-through), which is what the HIP path computes:
    its backward uses the fp32 x and the fp32 master weights."""
    h = encode_pre(x, W_enc, b_enc)
    with torch.no_grad():
        xb = x.to(torch.bfloat16).to(torch.float32)
        wb = W_enc.to(torch.bfloat16).to(torch.float32)
        delta = torch.einsum("bd,ds->bs", xb, wb) + b_enc - h
    return h + delta


def topk_mask(h: Tensor, k: int) -> Tensor:
    """0/1 mask with ones at the k largest *signed* pre-activations of each row (no ReLU; ties: any
    k).  modeling.py:174-177."""
    k = min(k, h.shape[-1])
    _, idx = torch.topk(h, k, dim=-1, sorted=False)
    return torch.zeros_like(h).scatter(-1, idx, 1.0)


def topk_activation(h: Tensor, k: int) -> Tensor:
    """f = mask * h; gradient reaches only selected entries.  modeling.py:169-179."""
    return topk_mask(h, k) * h


def decode(f: Tensor, W_dec: Tensor, b_dec: Tensor, prefixes: Tensor | None = None) -> Tensor:
    """Matryoshka cumulative decode -> (B, P, D).  Block i uses latents [p_{i-1}, p_i); the bias is
    added to block 0 only; prefix reconstructions are the running sum over blocks.
    modeling.py:351-409."""
    S = f.shape[1]
    if prefixes is None:
        prefixes = torch.tensor([S], dtype=torch.int64)
    cuts = [0] + [int(p) for p in prefixes]
    assert all(b > a for a, b in zip(cuts[1:-1], cuts[2:])), "prefixes must be strictly increasing"
    assert cuts[1] >= 1 and cuts[-1] == S
    parts = []
    for i, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
        part = torch.einsum("bs,sd->bd", f[:, lo:hi], W_dec[lo:hi, :])
        if i == 0:
            part = part + b_dec
        parts.append(part)
    return torch.cumsum(torch.stack(parts, dim=-2), dim=-2)


def mean_squared_err(x_hat: Tensor, x: Tensor) -> Tensor:
    """Elementwise squared error computed after dividing both operands by max|x| (clamped at
    1e-12) and multiplied back by that scale twice.  objectives.py:223-237 (norm=False branch)."""
    upper = x.abs().max().clamp(min=1e-12)
    diff = x_hat / upper - x / upper
    return diff**2 * upper * upper


def auxk_loss(
    *, x: Tensor, h: Tensor, x_hat_last: Tensor, dead_mask: Tensor, W_dec: Tensor, b_dec: Tensor,
    k_aux: int, alpha: float,
) -> Tensor:
    """AuxK dead-latent loss, training branch.  modeling.py:89-103.

    Target is the *detached* main residual x - x_hat; candidates are the pre-activations of dead
    latents only; k_use = min(k_aux, n_dead); the auxiliary reconstruction goes through the normal
    decoder and therefore includes b_dec; mean over B*D, scaled by alpha."""
    residual = (x - x_hat_last).detach()
    n_dead = int(dead_mask.sum().item())
    k_use = min(k_aux, n_dead)
    if k_use == 0:
        return residual.new_zeros(())
    masked = h.masked_fill(~dead_mask, float("-inf"))
    _, top_i = masked.topk(k_use, dim=-1)
    aux_acts = torch.zeros_like(h)
    aux_acts = aux_acts.scatter(-1, top_i, h.gather(-1, top_i))
    aux_recon = decode(aux_acts, W_dec, b_dec)[:, -1, :]
    return alpha * (aux_recon - residual).pow(2).mean()


def update_dead_tracker(toks_since_active: Tensor, f: Tensor, threshold: int) -> Tensor:
    """In-place tracker update; returns this step's dead mask.  objectives.py:114-120:
    every latent ages by B tokens, latents with any |f|>0 in the batch reset to 0, dead iff
    age >= threshold."""
    fired = (f.abs() > 0).any(dim=0)
    toks_since_active += f.shape[0]
    toks_since_active[fired] = 0
    return toks_since_active >= threshold


def sample_prefixes(d_sae: int, n_prefixes: int, min_prefix_length: int = 1, pareto_power: float = 0.5) -> Tensor:
    """n_prefixes <= 1 -> [d_sae] (objectives.py:177-178); otherwise n_prefixes - 1 lengths drawn without replacement
    from the discretised Pareto law with torch's global RNG, plus d_sae, sorted (objectives.py:183-201).  Pinned by
    fixture G15 (the reference's draws under fixed seeds)."""
    if n_prefixes <= 1:
        return torch.tensor([d_sae], dtype=torch.int64)
    lengths = torch.arange(1, d_sae)
    cdf = 1 - (min_prefix_length / lengths.float()) ** pareto_power
    pdf = torch.cat([cdf[:1], cdf[1:] - cdf[:-1]])
    pdf = pdf / pdf.sum()
    picks = torch.multinomial(pdf, num_samples=n_prefixes - 1, replacement=False)
    out = torch.cat((lengths[picks], torch.tensor([d_sae])))
    return torch.sort(out).values.to(torch.int64)


@dataclasses.dataclass
class StepOut:
    mse: Tensor
    aux: Tensor
    l0: Tensor
    l1: Tensor
    n_dead: int
    h: Tensor
    f: Tensor
    x_hats: Tensor

    @property
    def loss(self) -> Tensor:  # objectives.py:75-78 (sparsity term is NoSparsity -> 0)
        return self.mse + self.aux


def objective_forward(
    params: dict[str, Tensor], x: Tensor, cfg: RefConfig, *, toks_since_active: Tensor | None,
    training: bool = True, prefixes: Tensor | None = None,
) -> StepOut:
    """One objective evaluation.  objectives.py:101-156.

    ``toks_since_active`` (S,) int64 is updated in place in training mode.  In eval mode there is no
    dead tracking and the auxiliary term is zero (modeling.py:83-87, objectives.py:121-122)."""
    enc = encode_pre_bf16 if cfg.encoder_bf16 else encode_pre
    h = enc(x, params["W_enc"], params["b_enc"])
    f = topk_activation(h, cfg.top_k)
    dead_mask = None
    if training:
        assert toks_since_active is not None
        with torch.no_grad():
            dead_mask = update_dead_tracker(toks_since_active, f, cfg.dead_threshold_tokens)
    if prefixes is None:
        prefixes = sample_prefixes(cfg.d_sae, cfg.n_prefixes)
    x_hats = decode(f, params["W_dec"], params["b_dec"], prefixes)
    P = x_hats.shape[1]
    mse = mean_squared_err(x_hats, x[:, None, :].expand(-1, P, -1)).mean()
    if training and cfg.use_aux:
        h_aux = encode_pre(x, params["W_enc"], params["b_enc"]) if cfg.encoder_bf16 else h
        aux = auxk_loss(
            x=x, h=h_aux, x_hat_last=x_hats[:, -1, :], dead_mask=dead_mask,
            W_dec=params["W_dec"], b_dec=params["b_dec"], k_aux=cfg.k_aux, alpha=cfg.alpha,
        )
    else:
        aux = x.new_zeros(())
    return StepOut(
        mse=mse, aux=aux,
        l0=(f != 0).float().sum(dim=1).mean(dim=0),
        l1=f.abs().sum(dim=1).mean(dim=0),
        n_dead=int(dead_mask.sum().item()) if dead_mask is not None else 0,
        h=h, f=f, x_hats=x_hats,
    )


def remove_parallel_grads(g_W_dec: Tensor, W_dec: Tensor) -> Tensor:
    """Project each decoder-row gradient orthogonal to the row itself; rows of zero norm are left
    alone.  modeling.py:419-445."""
    dots = (g_W_dec * W_dec).sum(dim=1)
    nsq = (W_dec * W_dec).sum(dim=1)
    scale = torch.zeros_like(dots)
    nz = nsq > 0
    scale[nz] = dots[nz] / nsq[nz]
    return g_W_dec - scale[:, None] * W_dec


def clip_grad_norm(grads: list[Tensor], max_norm: float) -> tuple[list[Tensor], Tensor]:
    """torch.nn.utils.clip_grad_norm_ semantics (train.py:356-362): one L2 norm over all tensors,
    coef = min(1, max_norm / (total + 1e-6)); returns (scaled grads, pre-clip total)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adam_update(
    p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
    beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8,
) -> None:
    """torch.optim.Adam defaults (train.py:294), in place.  ``step`` is the 1-based step count."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1**step
    bc2 = 1 - beta2**step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)


class WarmupCosine:
    """scheduling.py:43-71: linear warm-up to ``peak`` over ``n_warmup`` calls, cosine to ``final``
    until ``n_steps``, ``final`` afterwards.  The counter increments *before* evaluation."""

    def __init__(self, init: float, n_warmup: int, peak: float, n_steps: int, final: float):
        self.init, self.n_warmup, self.peak, self.n_steps, self.final = init, n_warmup, peak, n_steps, final
        self.count = 0

    def step(self) -> float:
        self.count += 1
        if self.count < self.n_warmup:
            return self.init + (self.peak - self.init) * (self.count / self.n_warmup)
        if self.count < self.n_steps:
            frac = (self.count - self.n_warmup) / (self.n_steps - self.n_warmup)
            return self.final + (self.peak - self.final) * (1 + math.cos(math.pi * frac)) / 2
        return self.final


def limited_batches(epoch_batches: list[Tensor], n_samples: int, batch_size: int, drop_last: bool) -> Iterator[Tensor]:
    """BatchLimiter.__iter__ (scheduling.py:109-122): cycle over the loader until n_samples rows
    were yielded; after every exhausted epoch, when drop_last is False, the seen-counter is reduced
    by one nominal batch (so the loop can run more steps than ceil(n_samples/batch_size))."""
    seen = 0
    while True:
        for b in epoch_batches:
            yield b
            seen += len(b)
            if seen >= n_samples:
                return
        if not drop_last:
            seen -= batch_size


@dataclasses.dataclass
class TrainState:
    params: dict[str, Tensor]
    m: dict[str, Tensor]
    v: dict[str, Tensor]
    toks_since_active: Tensor
    adam_steps: int = 0
    lr: float = 0.0  # first optimizer step runs with lr = 0 (train.py:118)

    @classmethod
    def create(cls, params: dict[str, Tensor]) -> "TrainState":
        return cls(
            params={k: params[k].clone() for k in PARAM_ORDER},
            m={k: torch.zeros_like(params[k]) for k in PARAM_ORDER},
            v={k: torch.zeros_like(params[k]) for k in PARAM_ORDER},
            toks_since_active=torch.zeros(params["b_enc"].shape[0], dtype=torch.int64),
        )


def train_step(state: TrainState, x: Tensor, cfg: RefConfig, sched: WarmupCosine | None = None) -> dict[str, float]:
    """One iteration of the loop body at train.py:332-460 for a single SAE:
    renormalise decoder rows -> objective forward -> backward -> project out parallel gradients ->
    global-norm clip -> Adam with the *previous* iteration's lr -> scheduler step."""
    P = state.params
    if cfg.normalize_w_dec:
        P["W_dec"] = normalize_w_dec(P["W_dec"])
    leaves = {k: P[k].detach().requires_grad_(True) for k in PARAM_ORDER}
    out = objective_forward(leaves, x, cfg, toks_since_active=state.toks_since_active, training=True)
    out.loss.backward()
    grads = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(P[k])) for k in PARAM_ORDER}
    if cfg.remove_parallel_grads:
        grads["W_dec"] = remove_parallel_grads(grads["W_dec"], P["W_dec"])
    clipped, total = clip_grad_norm([grads[k] for k in PARAM_ORDER], cfg.grad_clip)
    state.adam_steps += 1
    lr_used = state.lr
    for k, g in zip(PARAM_ORDER, clipped):
        adam_update(P[k], g, state.m[k], state.v[k], state.adam_steps, lr_used)
    if sched is not None:
        state.lr = sched.step()
    return {
        "mse": out.mse.item(), "aux": out.aux.item(), "loss": out.loss.item(),
        "l0": out.l0.item(), "l1": out.l1.item(), "n_dead": out.n_dead,
        "grad_norm": total.item(), "lr": lr_used,
        "grads": {k: grads[k].detach() for k in PARAM_ORDER},
    }


def train_loop(
    params0: dict[str, Tensor], epoch_batches: list[Tensor], cfg: RefConfig, *, n_train: int,
    batch_size: int, drop_last: bool = False,
) -> tuple[TrainState, list[dict[str, float]]]:
    """train.py:238-462 for one SAE on an in-memory epoch of batches."""
    state = TrainState.create(params0)
    sched = WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, math.ceil(n_train / batch_size), 0.0)
    log = []
    for x in limited_batches(epoch_batches, n_train, batch_size, drop_last):
        rec = train_step(state, x, cfg, sched)
        rec.pop("grads")
        log.append(rec)
    return state, log


@torch.no_grad()
def evaluate(params: dict[str, Tensor], batches: Iterable[Tensor], cfg: RefConfig) -> dict[str, object]:
    """train.py:510-618 for one SAE: eval-mode objective per batch, fp64 accumulators for the
    baseline/SAE sums of squares, per-latent firing counts (f > 0) and value sums."""
    S, D = cfg.d_sae, cfg.d_model
    n_fired = torch.zeros(S)
    values = torch.zeros(S)
    l0 = l1 = mse = 0.0
    sse = torch.zeros((), dtype=torch.float64)
    sum_sq = torch.zeros((), dtype=torch.float64)
    sum_vec = torch.zeros(D, dtype=torch.float64)
    n = 0
    for x in batches:
        b = x.shape[0]
        x64 = x.double()
        sum_sq += (x64 * x64).sum()
        sum_vec += x64.sum(dim=0)
        n += b
        out = objective_forward(params, x, cfg, toks_since_active=None, training=False)
        sse += ((x - out.x_hats[:, -1, :]).double() ** 2).sum()
        n_fired += (out.f > 0).sum(dim=0)
        values += out.f.sum(dim=0)
        l0 += out.l0.item() * b
        l1 += out.l1.item() * b
        mse += out.mse.item() * b
    assert n > 0
    sse_baseline = (sum_sq - torch.dot(sum_vec, sum_vec) / n).item()
    assert sse_baseline > 0
    freqs = n_fired / n
    return {
        "l0": l0 / n, "l1": l1 / n, "mse": mse / n,
        "normalized_mse": sse.item() / sse_baseline, "sse_sae": sse.item(), "sse_baseline": sse_baseline,
        "n_dead": int((freqs == 0).sum()), "n_almost_dead": int((freqs < 1e-7).sum()),
        "n_dense": int((freqs > 1e-2).sum()), "freqs": freqs, "mean_values": values / n_fired,
    }
