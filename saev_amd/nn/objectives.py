"""saev.nn.objectives' public surface (reference: src/saev/nn/objectives.py) over the HIP engine.

``get_objective(Matryoshka(...))`` returns a module whose ``forward(sae, x)`` gives
``(MatryoshkaLoss, Output)`` like the reference (objectives.py:92-156): encode + TopK, dead-latent
tracking, decode, MSE with the max|x| rescale (objectives.py:223-237), AuxK (modeling.py:75-103).
All of it runs in libsaev_amd.so; ``loss.loss.backward()`` runs the HIP sparse backward and leaves the
four parameter gradients in ``param.grad`` (views of the engine's flat gradient buffer).

Matryoshka prefixes (``n_prefixes > 1``, the reference default of 10): the cut points are sampled on the
host every call exactly as the reference does (Pareto law, torch global RNG, objectives.py:159-201) and
handed to the kernels, which form all nested reconstructions from the (latent-ordered) codes in one sweep.
"""

from __future__ import annotations

import dataclasses
import functools
import typing as tp

import torch
from torch import Tensor

from . import modeling


@dataclasses.dataclass(frozen=True, slots=True)
class Matryoshka:
    """objectives.py:13-25."""

    n_prefixes: int = 10
    dead_threshold_tokens: int = 10_000_000


ObjectiveConfig = Matryoshka


class Loss:
    @property
    def loss(self) -> Tensor:
        raise NotImplementedError()

    def metrics(self) -> dict[str, object]:
        raise NotImplementedError()


@dataclasses.dataclass(frozen=True, slots=True)
class MatryoshkaLoss(Loss):
    """objectives.py:57-89.  ``total`` carries the autograd edge into the HIP backward."""

    mse: Tensor
    sparsity: Tensor
    l0: Tensor
    l1: Tensor
    aux: Tensor
    n_dead: tp.Any
    total: Tensor | None = None

    @property
    def loss(self) -> Tensor:
        if self.total is not None:
            return self.total
        return self.mse + self.sparsity + self.aux

    def metrics(self) -> dict[str, object]:
        return {
            "loss": self.loss.item(), "mse": self.mse.item(), "l0": self.l0.item(), "l1": self.l1.item(),
            "sparsity": self.sparsity.item(), "aux": self.aux.item(), "n_dead": self.n_dead,
        }


class Objective(torch.nn.Module):
    def forward(self, sae: modeling.SparseAutoencoder, x: Tensor):
        raise NotImplementedError()


class _HipStep(torch.autograd.Function):
    """loss = mse + aux as an autograd node whose backward is saev_step_backward."""

    @staticmethod
    def forward(ctx, total: Tensor, eng, W_dec, b_dec, W_enc, b_enc):
        ctx.eng = eng
        return total.clone()

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.eng
        eng.step_backward()
        g = eng.grad_views()
        scale = grad_out.to(torch.float32)
        outs = []
        for name in ("W_dec", "b_dec", "W_enc", "b_enc"):
            outs.append(g[name] * scale)
        return (None, None, *outs)


class MatryoshkaObjective(Objective):
    def __init__(self, cfg: Matryoshka):
        super().__init__()
        self.cfg = cfg
        self.__dict__["_eng_ref"] = None
        self.__dict__["_pending_toks"] = None

    # the tracker lives in the engine (device int64, d_sae); the reference keeps it on the objective
    @property
    def toks_since_active(self) -> Tensor | None:
        eng = self.__dict__["_eng_ref"]
        if eng is None:
            return self.__dict__["_pending_toks"]
        return eng.toks_since_active

    @toks_since_active.setter
    def toks_since_active(self, value: Tensor | None):
        eng = self.__dict__["_eng_ref"]
        if eng is None:
            self.__dict__["_pending_toks"] = value
        else:
            eng.set_tracker(value)

    def _bind(self, sae: modeling.SparseAutoencoder, n_rows: int):
        sae.__dict__["_dead_threshold_tokens"] = self.cfg.dead_threshold_tokens
        eng = sae._eng(n_rows)
        if self.__dict__["_eng_ref"] is not eng:
            self.__dict__["_eng_ref"] = eng
            pending = self.__dict__["_pending_toks"]
            if pending is not None:
                eng.set_tracker(pending)
                self.__dict__["_pending_toks"] = None
        return eng

    def forward(self, sae: modeling.SparseAutoencoder, x: Tensor):
        n = x.shape[0]
        eng = self._bind(sae, n)
        x = x.detach()
        prefixes = sample_prefixes(sae.cfg.d_sae, self.cfg.n_prefixes)
        eng.set_prefixes(prefixes if self.cfg.n_prefixes > 1 else None)
        if self.training:
            assert sae.training, "objective.train() with sae.eval(): AuxK needs a dead mask only in training"
            eng.step_forward(x, training=True)
            eng.step_dead(n)
        else:
            eng.step_forward(x, training=False)
        st = eng.read_stats()
        dev = eng.device
        t = lambda v: torch.tensor(v, device=dev, dtype=torch.float32)  # noqa: E731
        total = t(st.mse + st.aux)
        if self.training and torch.is_grad_enabled():
            total = _HipStep.apply(total, eng, sae.W_dec, sae.b_dec, sae.W_enc, sae.b_enc)
        loss = MatryoshkaLoss(
            mse=t(st.mse), sparsity=torch.tensor(0.0), l0=t(st.l0), l1=t(st.l1), aux=t(st.aux),
            n_dead=torch.tensor(st.n_dead, device=dev) if self.training else torch.tensor(0), total=total,
        )
        idx, val, x_hat = eng.last_codes(n)
        if self.cfg.n_prefixes > 1:
            return loss, modeling.Output(sae, x, idx, val, None, prefixes=prefixes)
        return loss, modeling.Output(sae, x, idx, val, x_hat[:, None, :])


@functools.lru_cache(maxsize=8)
def _prefix_law(d_sae: int, min_prefix_length: int, pareto_power: float) -> tuple[Tensor, Tensor]:
    """Lengths 1..d_sae-1 and their probabilities under the discretised Pareto law P(len <= L) = 1 - (min/L)^power
    (objectives.py:183-186).  Deterministic, so it is computed once per d_sae; only the draw is per step."""
    lengths = torch.arange(1, d_sae)
    cdf = 1 - (min_prefix_length / lengths.float()) ** pareto_power
    pdf = torch.cat([cdf[:1], cdf[1:] - cdf[:-1]])
    return lengths, pdf / pdf.sum()


@torch.no_grad()
def sample_prefixes(d_sae: int, n_prefixes: int, min_prefix_length: int = 1, pareto_power: float = 0.5) -> Tensor:
    """Sorted prefix lengths ending in d_sae; n_prefixes-1 lengths drawn without replacement from the Pareto law
    with torch's global CPU RNG -- the same draw, consuming the same random numbers, as objectives.py:159-201."""
    if n_prefixes <= 1:
        return torch.tensor([d_sae], dtype=torch.int64)
    assert n_prefixes <= d_sae
    lengths, pdf = _prefix_law(d_sae, min_prefix_length, pareto_power)
    picks = torch.multinomial(pdf, num_samples=n_prefixes - 1, replacement=False)
    out = torch.cat((lengths[picks], torch.tensor([d_sae])))
    return torch.sort(out).values.to(torch.int64)


def get_objective(cfg: ObjectiveConfig) -> Objective:
    if isinstance(cfg, Matryoshka):
        return MatryoshkaObjective(cfg)
    raise TypeError(f"unknown objective config {cfg!r}")
