"""saev.nn.modeling's public surface (reference: src/saev/nn/modeling.py) over the HIP engine.

Same names, fields, defaults and checkpoint format as the reference so configs, sweeps and ``sae.pt``
files move between the two unchanged:

* config dataclasses ``NoSparsity / L1Sparsity / NoAux / AuxK / Relu / TopK / BatchTopK /
  SparseAutoencoderConfig`` (modeling.py:23-146, 259-284);
* ``SparseAutoencoder`` — a ``torch.nn.Module`` with Parameters ``W_dec (S,D), b_dec (D), W_enc (D,S),
  b_enc (S)`` in that state_dict order (modeling.py:306-329) and methods ``encode / decode / forward /
  normalize_w_dec / remove_parallel_grads`` (modeling.py:331-445);
* ``dump`` / ``load`` — one JSON header line + ``torch.save(state_dict)`` (modeling.py:548-658).

What differs: every compute method runs hand-written HIP kernels through libsaev_amd.so on the
module's parameters, which are views into one flat device buffer owned by ``saev_amd.engine.SaeEngine``.
There is no CPU implementation: calling a compute method on CPU tensors raises.  Only the TopK
activation is on the accelerated path (BASELINE.json north_star); ``Relu`` and ``BatchTopK`` configs
still parse and round-trip through checkpoints but raise ``NotImplementedError`` when run.
"""

from __future__ import annotations

import dataclasses
import io
import json
import logging
import pathlib
import typing as tp

import torch
from torch import Tensor

from .. import __version__
from ..engine import EngineConfig, SaeEngine

SCHEMA_VERSION = 5


# ------------------------------------------------------------------------------------------------
# configs
# ------------------------------------------------------------------------------------------------


@dataclasses.dataclass(frozen=True)
class NoSparsity:
    """No explicit sparsity penalty (modeling.py:25-31)."""

    key: tp.Literal["no-sparsity"] = "no-sparsity"


@dataclasses.dataclass(frozen=True)
class L1Sparsity:
    key: tp.Literal["l1-sparsity"] = "l1-sparsity"
    coeff: float = 1e-4


Sparsity = tp.Union[NoSparsity, L1Sparsity]


@dataclasses.dataclass(frozen=True)
class NoAux:
    key: tp.Literal["no-aux"] = "no-aux"


@dataclasses.dataclass(frozen=True)
class AuxK:
    """AuxK dead-latent reconstruction loss (modeling.py:66-103)."""

    key: tp.Literal["auxk"] = "auxk"
    k_aux: int = 512
    alpha: float = 1 / 32


Aux = tp.Union[AuxK, NoAux]


@dataclasses.dataclass(frozen=True)
class Relu:
    key: tp.Literal["relu"] = "relu"
    sparsity: Sparsity = L1Sparsity(coeff=4e-4)
    aux: Aux = NoAux()


@dataclasses.dataclass(frozen=True)
class TopK:
    key: tp.Literal["top-k"] = "top-k"
    top_k: int = 32
    sparsity: Sparsity = NoSparsity()
    aux: Aux = AuxK()

    def __post_init__(self):
        assert self.top_k > 0, f"top_k = {self.top_k}: at least one latent must be kept"


@dataclasses.dataclass(frozen=True)
class BatchTopK:
    key: tp.Literal["batch-top-k"] = "batch-top-k"
    top_k: int = 32
    sparsity: Sparsity = NoSparsity()
    momentum: float = 0.1
    aux: AuxK = AuxK()

    def __post_init__(self):
        assert self.top_k > 0, f"top_k = {self.top_k}: at least one latent must be kept"


ActivationConfig = tp.Union[Relu, TopK, BatchTopK]


@dataclasses.dataclass(frozen=True)
class SparseAutoencoderConfig:
    d_model: int = 1024
    d_sae: int = 1024 * 16
    activation: ActivationConfig = TopK()
    reinit_blend: float = 0.8
    reinit_enc_dec_tranpose: bool = True  # (sic) spelling kept: it is a checkpoint/config key
    remove_parallel_grads: bool = True
    normalize_w_dec: bool = True


_CONFIG_CLASSES = {c.__name__: c for c in (NoSparsity, L1Sparsity, NoAux, AuxK, Relu, TopK, BatchTopK)}


# ------------------------------------------------------------------------------------------------
# outputs
# ------------------------------------------------------------------------------------------------


class EncodeOut(tp.NamedTuple):
    """Dense pre-activations and activated latents (modeling.py:292-296)."""

    h_x: Tensor
    f_x: Tensor


class Output:
    """Forward outputs with the reference's field names (modeling.py:299-304).

    The HIP path keeps the k-sparse codes ``idx`` / ``val`` (batch, top_k); the dense ``h_x`` / ``f_x``
    (batch, d_sae) matrices are materialised only when read."""

    def __init__(self, sae: "SparseAutoencoder", x: Tensor, idx: Tensor, val: Tensor, x_hats: Tensor | None,
                 h_x: Tensor | None = None, prefixes: Tensor | None = None):
        self._sae, self._x, self.idx, self.val, self._x_hats = sae, x, idx, val, x_hats
        self._h_x, self._f_x, self.prefixes = h_x, None, prefixes

    @property
    def x_hats(self) -> Tensor:
        """(batch, n_prefixes, d_model); with Matryoshka prefixes the nested reconstructions are decoded on demand."""
        if self._x_hats is None:
            self._x_hats = self._sae._eng().decode_sparse(self.idx, self.val, prefixes=self.prefixes)
        return self._x_hats

    @property
    def h_x(self) -> Tensor:
        if self._h_x is None:
            self._h_x = self._sae._eng().encode_dense(self._x)
        return self._h_x

    @property
    def f_x(self) -> Tensor:
        if self._f_x is None:
            self._f_x = self._sae._eng().scatter_dense(self.idx, self.val)
        return self._f_x


# ------------------------------------------------------------------------------------------------
# module
# ------------------------------------------------------------------------------------------------


class TopKActivation(torch.nn.Module):
    """Per-row top-k of signed pre-activations (modeling.py:160-179), on the HIP select kernel."""

    def __init__(self, cfg: TopK, sae: "SparseAutoencoder | None" = None):
        super().__init__()
        self.cfg = cfg
        self.__dict__["_sae"] = sae  # not a submodule

    def forward(self, x: Tensor) -> Tensor:
        sae = self.__dict__["_sae"]
        if sae is None:
            raise RuntimeError("TopKActivation needs its owning SparseAutoencoder to reach the HIP engine")
        eng = sae._eng()
        k = min(self.cfg.top_k, x.shape[-1])
        idx, val = eng.topk_dense(x, k)
        return eng.scatter_dense(idx, val)


class _Unsupported(torch.nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg

    def forward(self, x):
        raise NotImplementedError(
            f"{type(self.cfg).__name__} is outside the MI355X hot path (TopK only); the config is kept so "
            "sweeps/checkpoints parse.")


class SparseAutoencoder(torch.nn.Module):
    """Sparse auto-encoder (modeling.py:287-445)."""

    EncodeOut = EncodeOut
    Output = Output

    def __init__(self, cfg: SparseAutoencoderConfig):
        super().__init__()
        self.cfg = cfg
        self.logger = logging.getLogger("sae")
        # Kaiming-uniform decoder rows, unit-normalised; the encoder is an independent copy of its transpose.
        W_dec = torch.nn.init.kaiming_uniform_(torch.empty(cfg.d_sae, cfg.d_model))
        if cfg.normalize_w_dec:
            W_dec /= torch.norm(W_dec, dim=1, keepdim=True)
        self.W_dec = torch.nn.Parameter(W_dec)
        self.b_dec = torch.nn.Parameter(torch.zeros(cfg.d_model))
        self.W_enc = torch.nn.Parameter(W_dec.T.clone())
        self.b_enc = torch.nn.Parameter(torch.zeros(cfg.d_sae))
        if isinstance(cfg.activation, TopK):
            self.activation = TopKActivation(cfg.activation, self)
        else:
            self.activation = _Unsupported(cfg.activation)
        self.__dict__["_engine"] = None
        self.__dict__["_engine_max_batch"] = 0

    # ---- engine binding ---------------------------------------------------------------------
    def _engine_cfg(self, max_batch: int, objective_cfg=None) -> EngineConfig:
        act = self.cfg.activation
        if not isinstance(act, TopK):
            raise NotImplementedError(f"{type(act).__name__} activation is not on the HIP path (TopK only)")
        if not isinstance(act.sparsity, NoSparsity):
            raise NotImplementedError("TopK with an explicit sparsity penalty is not on the HIP path")
        aux = act.aux
        thr = getattr(self, "_dead_threshold_tokens", 10_000_000)
        return EngineConfig(
            d_model=self.cfg.d_model, d_sae=self.cfg.d_sae, top_k=act.top_k,
            k_aux=aux.k_aux if isinstance(aux, AuxK) else 0, alpha=aux.alpha if isinstance(aux, AuxK) else 0.0,
            dead_threshold_tokens=thr, normalize_w_dec=self.cfg.normalize_w_dec,
            remove_parallel_grads=self.cfg.remove_parallel_grads, max_batch=max_batch,
            shard_world=getattr(self, "_shard_world", 1), max_backward_rows=getattr(self, "_max_backward_rows", 0),
        )

    def _eng(self, max_batch: int = 0) -> SaeEngine:
        """The engine whose flat buffer backs the four Parameters; (re)built when the module moved
        device, a larger batch arrives or the tracker threshold changed."""
        dev = self.W_dec.device
        if dev.type != "cuda":
            raise RuntimeError("saev_amd runs on a HIP device only: move the module with .to('cuda') first "
                               "(there is no CPU path)")
        eng: SaeEngine | None = self.__dict__["_engine"]
        want_batch = max(max_batch, self.__dict__["_engine_max_batch"], 1)
        stale = (
            eng is None or eng.device != dev or eng.cfg.max_batch < want_batch
            or eng.cfg.dead_threshold_tokens != getattr(self, "_dead_threshold_tokens", eng.cfg.dead_threshold_tokens)
            or eng.cfg.shard_world != getattr(self, "_shard_world", eng.cfg.shard_world)
            or max(eng.cfg.max_batch, eng.cfg.max_backward_rows) < getattr(self, "_max_backward_rows", 0)
            or any(getattr(self, n).data_ptr() != eng.view(n).data_ptr() for n in eng.offsets)
        )
        if stale:
            old = eng
            values = {n: getattr(self, n).detach().clone() for n in ("W_dec", "b_dec", "W_enc", "b_enc")}
            eng = SaeEngine(self._engine_cfg(max(want_batch, 1024)), dev)
            eng.load_params(values)
            if old is not None and old.device == dev:
                eng.set_tracker(old.toks_since_active)
                # per tensor: the padded layout of a sharded tail (shard_world > 1) has other offsets and another length
                for n in eng.offsets:
                    eng.view(n, eng.adam_m).copy_(old.view(n, old.adam_m))
                    eng.view(n, eng.adam_v).copy_(old.view(n, old.adam_v))
                eng.adam_steps = old.adam_steps
                old.close()
            for n in eng.offsets:
                getattr(self, n).data = eng.view(n)
            eng.watch([getattr(self, n) for n in eng.offsets])  # (their version counters are their own: see SaeEngine.watch)
            self.__dict__["_engine"] = eng
            self.__dict__["_engine_max_batch"] = eng.cfg.max_batch
        # The engine keeps what its forward needs of W_enc / W_dec between calls (include/saev_amd.h, PARAMETER OWNERSHIP).  The four
        # Parameters are views of its buffer with version counters of their own: an in-place write through them (load_state_dict,
        # an initialiser, a user's sae.W_enc.mul_()) shows here and drops what the engine kept.  Writes through .data show nowhere.
        try:
            vers = tuple(getattr(self, n)._version for n in ("W_dec", "b_dec", "W_enc", "b_enc"))
        except RuntimeError:  # inference tensors carry no version counter
            vers = None
        if vers is None or vers != self.__dict__.get("_param_versions"):
            eng.params_touched()
            self.__dict__["_param_versions"] = vers
        return eng

    # ---- reference API ----------------------------------------------------------------------
    def forward(self, x: Tensor) -> Output:
        eng = self._eng(x.shape[0])
        eng.step_forward(x, training=False)
        idx, val, x_hat = eng.last_codes(x.shape[0])
        return Output(self, x, idx, val, x_hat[:, None, :])

    def encode(self, x: Tensor) -> EncodeOut:
        eng = self._eng(x.shape[0])
        h_x = eng.encode_dense(x.reshape(-1, self.cfg.d_model))
        f_x = self.activation(h_x)
        shape = (*x.shape[:-1], self.cfg.d_sae)
        return EncodeOut(h_x=h_x.reshape(shape), f_x=f_x.reshape(shape))

    def encode_sparse(self, x: Tensor) -> tuple[Tensor, Tensor]:
        """(idx, val) codes, (batch, top_k) each, without materialising the dense matrices."""
        return self._eng(x.shape[0]).encode_topk(x)

    def decode(self, f_x: Tensor, *, prefixes: Tensor | None = None) -> Tensor:
        """(batch, n_prefixes, d_model) Matryoshka reconstructions of dense latents (modeling.py:351-409)."""
        b, d_sae = f_x.shape
        if prefixes is None:
            prefixes = torch.tensor([d_sae], dtype=torch.int64)
        pre = [int(p) for p in prefixes]
        assert all(b_ > a_ for a_, b_ in zip(pre[:-1], pre[1:]))
        assert 1 <= pre[0] and pre[-1] == d_sae
        eng = self._eng(b)
        nz = f_x != 0
        k = max(int(nz.sum(dim=1).max().item()), 1)
        idx, _ = eng.topk_dense(nz.to(torch.float32), k)  # ties -> lowest index first: all non-zeros, then zeros
        val = f_x.gather(1, idx.long())
        return eng.decode_sparse(idx, val, prefixes=pre)

    @torch.no_grad()
    def normalize_w_dec(self):
        if self.cfg.normalize_w_dec:
            self._eng().normalize_w_dec()

    @torch.no_grad()
    def remove_parallel_grads(self):
        if not self.cfg.remove_parallel_grads or self.W_dec.grad is None:
            return
        eng = self._eng()
        g = eng.view("W_dec", eng.grads)
        if self.W_dec.grad.data_ptr() != g.data_ptr():
            g.copy_(self.W_dec.grad)
            self.W_dec.grad = g
        eng.remove_parallel_grads()


# ------------------------------------------------------------------------------------------------
# checkpoint I/O (modeling.py:448-658)
# ------------------------------------------------------------------------------------------------


def _ser(value: tp.Any) -> tp.Any:
    if dataclasses.is_dataclass(value) and not isinstance(value, type):
        return {"cls": type(value).__name__,
                "params": {f.name: _ser(getattr(value, f.name)) for f in dataclasses.fields(value)}}
    if isinstance(value, (tuple, list)):
        return [_ser(v) for v in value]
    if isinstance(value, dict):
        return {k: _ser(v) for k, v in value.items()}
    return value


def _deser(value: tp.Any) -> tp.Any:
    """Inverse of ``_ser``: ``{"cls", "params"}`` nodes become instances of the named config class."""
    if isinstance(value, list):
        return [_deser(v) for v in value]
    if not isinstance(value, dict):
        return value
    if set(value) == {"cls", "params"}:
        if value["cls"] not in _CONFIG_CLASSES:
            raise ValueError(f"checkpoint names a config class this package does not have: {value['cls']!r}")
        return _CONFIG_CLASSES[value["cls"]](**{k: _deser(v) for k, v in value["params"].items()})
    return {k: _deser(v) for k, v in value.items()}


def _git_commit() -> str:
    import subprocess

    try:
        out = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, timeout=5,
                             cwd=pathlib.Path(__file__).parent)
        return out.stdout.strip() or "unknown"
    except Exception:
        return "unknown"


def dump(fpath: pathlib.Path | str, sae: SparseAutoencoder):
    """Write ``sae.pt``: header line ``{"schema":5,"cfg":...,"commit":...,"lib":...}\\n`` then
    ``torch.save(state_dict)`` with four independent CPU tensors, keys ``W_dec,b_dec,W_enc,b_enc``."""
    cfg_dict = dataclasses.asdict(sae.cfg)
    cfg_dict["activation"] = _ser(sae.cfg.activation)
    header = {"schema": SCHEMA_VERSION, "cfg": cfg_dict, "commit": _git_commit(), "lib": __version__}
    fpath = pathlib.Path(fpath)
    fpath.parent.mkdir(exist_ok=True, parents=True)
    state = {k: v.detach().to("cpu").clone() for k, v in sae.state_dict().items()}
    with open(fpath, "wb") as fd:
        fd.write(json.dumps(header, separators=(",", ":")).encode() + b"\n")
        torch.save(state, fd)


# Header layouts older than schema 5 (what ``load`` accepts besides the current one; the reference reads the same set,
# modeling.py:586-645).  Each layout is reduced to the field dict of ``SparseAutoencoderConfig`` by ``_config_fields``:
#
#   no "schema" key   the header IS the field dict of the first ReLU trainer: ``d_vit`` for d_model, ``exp_factor`` for
#                     the width, loss / seed knobs that no longer exist; the activation is always ReLU
#   schema 1, form A  ``"cls"`` names the activation (Relu / TopK / BatchTopK) and ``cfg`` is flat (``top_k`` beside the widths)
#   schema 1, form B  ``cfg["activation"]`` is a ``{"cls", "params"}`` tree -- as schemas 2-4
#   schemas 2-4       the tree of schema 5 with two older spellings inside it: a field called ``kind`` where the dataclasses
#                     now say ``key``, and ``sparsity`` as a bare dict (``{}`` = none, ``{"coeff": c}`` = L1)
_RETIRED_FIELDS = ("sparsity_coeff", "ghost_grads", "l1_coeff", "use_ghost_grads", "seed", "n_reinit_samples")
_FLAT_ACTIVATIONS = {"Relu": Relu, "TopK": TopK, "BatchTopK": BatchTopK}


def _widths(fields: dict[str, tp.Any]) -> dict[str, tp.Any]:
    """Drop retired knobs; turn ``exp_factor`` into ``d_sae`` (older trainers stored the expansion, not the width)."""
    out = {k: v for k, v in fields.items() if k not in _RETIRED_FIELDS}
    if "exp_factor" in out:
        factor = out.pop("exp_factor")
        if "d_sae" not in out:
            if out.get("d_model") is None:
                raise ValueError("legacy checkpoint stores exp_factor but no d_model: the latent width cannot be derived")
            out["d_sae"] = out["d_model"] * factor
    return out


def _deser_legacy(value: tp.Any, field: str = "") -> tp.Any:
    """``_deser`` for schemas 1-4: ``kind`` reads as ``key``; a bare ``sparsity`` dict becomes its dataclass."""
    if isinstance(value, list):
        return [_deser_legacy(v, field) for v in value]
    if not isinstance(value, dict):
        return value
    if "cls" in value and "params" in value:
        if value["cls"] not in _CONFIG_CLASSES:
            raise ValueError(f"checkpoint names a config class this package does not have: {value['cls']!r}")
        kwargs: dict[str, tp.Any] = {}
        for name, v in value["params"].items():
            name = "key" if name == "kind" else name
            if name in kwargs:
                raise ValueError(f"{value['cls']}: both 'kind' and 'key' present in a legacy checkpoint header")
            kwargs[name] = _deser_legacy(v, name)
        return _CONFIG_CLASSES[value["cls"]](**kwargs)
    if field == "sparsity":
        if not value:
            return NoSparsity()
        if set(value) <= {"coeff"}:
            return L1Sparsity(**value)
    return {k: _deser_legacy(v, field) for k, v in value.items()}


def _config_fields(header: dict[str, tp.Any], where: str) -> dict[str, tp.Any]:
    """Field dict of ``SparseAutoencoderConfig`` from a checkpoint header of any schema the reference reads."""
    if "schema" not in header:
        fields = dict(header)
        fields["d_model"] = fields.pop("d_vit")
        fields = _widths(fields)
        fields["activation"] = Relu()
        return fields
    schema = header["schema"]
    if schema == SCHEMA_VERSION:
        fields = _widths(dict(header["cfg"]))
        fields["activation"] = _deser(fields["activation"])
        return fields
    if schema in (1, 2, 3, 4):
        fields = _widths(dict(header["cfg"]))
        flat = _FLAT_ACTIVATIONS.get(header.get("cls", "")) if schema == 1 else None
        if flat is not None:  # schema 1, form A
            top_k = fields.pop("top_k", 32)
            fields["activation"] = flat() if flat is Relu else flat(top_k=top_k)
        elif "activation" in fields or schema != 1:
            fields["activation"] = _deser_legacy(fields["activation"])
        return fields
    raise ValueError(f"{where}: checkpoint schema {schema!r} is not supported (this loader reads schemas 1-{SCHEMA_VERSION} "
                     "and the pre-schema layout)")


def load(fpath: pathlib.Path | str, *, device="cpu") -> SparseAutoencoder:
    """Read an ``sae.pt``: schema 5 (what ``dump`` here and the reference's ``nn.dump`` write, modeling.py:548-574) and every
    older layout the reference's loader still reads (modeling.py:586-645; table above ``_config_fields``).  Only TopK
    checkpoints run on the HIP path; ReLU / BatchTopK ones load (parameters and config) and raise when run."""
    with open(fpath, "rb") as fd:
        first_line = fd.readline()
        payload = io.BytesIO(fd.read())
    fields = _config_fields(json.loads(first_line), str(fpath))
    model = SparseAutoencoder(SparseAutoencoderConfig(**fields))
    model.load_state_dict(torch.load(payload, weights_only=True, map_location="cpu"))
    return model.to(device)
