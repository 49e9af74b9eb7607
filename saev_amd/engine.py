"""Host-side owner of one SAE's device state and the C-ABI context that runs its train step.

``SaeEngine`` allocates the four parameter-sized flat buffers (params, grads, Adam m, Adam v) as
torch tensors on a HIP device, exposes ``W_dec / b_dec / W_enc / b_enc`` as views into the flat
parameter buffer (state_dict order, reference nn/modeling.py:312-327), and forwards every compute
call to libsaev_amd.so.  PyTorch is used for memory, streams and ``torch.distributed`` only.
"""

from __future__ import annotations

import ctypes as C
import dataclasses
import math
import os

import torch

from . import _lib


DEFAULT_ENCODER = "f16r"


@dataclasses.dataclass(frozen=True)
class EngineConfig:
    d_model: int
    d_sae: int
    top_k: int = 32
    k_aux: int = 512           # 0 disables the auxiliary loss
    alpha: float = 1.0 / 32.0
    dead_threshold_tokens: int = 10_000_000
    normalize_w_dec: bool = True
    remove_parallel_grads: bool = True
    max_batch: int = 16384
    aux_dead_cap: int = 0      # dead set the dense AuxK buffers are sized for at creation; 0 = min(d_sae, max(4096, 8 k_aux)).
                               # A step that meets more dead latents GROWS them (a device-synchronising free + allocate inside
                               # that step, reported on stderr; it fails only if the larger buffers do not fit the device)
    shard_world: int = 1       # > 1: flat buffers padded so that this many data-parallel ranks can each own 1/N of the tail
    max_backward_rows: int = 0  # 0 = max_batch; the GLOBAL batch for the sparse-state exchange (gathered backward over every
                                # rank's rows): sizes the backward's scratch only, the forward's buffers stay at max_batch
    # TopK candidate bounds of the fused encoder: "guaranteed" (default), or "predicted": verified extrapolated bounds
    # with an automatic guaranteed-bound re-run when a prediction fails -- same codes either way; measured no faster over
    # a training run (tools/experiments/README.md), kept as an option.  SAEV_AMD_BOUNDS overrides the default
    bounds: str = dataclasses.field(default_factory=lambda: os.environ.get("SAEV_AMD_BOUNDS", "guaranteed"))
    # "f32": exact fp32 MFMA; "f16x3": split-fp16 MFMA at fp32 accuracy (16/3 of the f32 matrix rate);
    # "bf16": bf16-rounded encoder operands, one MFMA product, fp32 accumulate (everything else stays fp32);
    # "f16r": one fp16 MFMA product as a bounded-error first pass + exact fp32 recomputation of the surviving candidates.
    # The default can be overridden with the SAEV_AMD_ENCODER environment variable.
    encoder: str = dataclasses.field(default_factory=lambda: os.environ.get("SAEV_AMD_ENCODER", DEFAULT_ENCODER))
    # Route switches for A/B runs and for tests that must reach a particular kernel (saev_debug_cfg; same results on every
    # route).  The C library reads no environment variable; these defaults do, so that a test or a shell script can flip a
    # route without touching code:
    #   SAEV_AMD_DW=rows          weight gradients by whole-row gathers instead of column slices
    #   SAEV_AMD_DW=slices_a      column slices, but dval = <dL/dx_hat row, decoder row> formed by their first pass instead of by the decode
    #   SAEV_AMD_DW=slices_s      as the default, but the decode itself gathers 32-column slices of W_dec out of the XCD L2s (decode_s_kernel; a wash)
    #   SAEV_AMD_FWD=rows         exact refinement of the f16r encoder by whole-row gathers instead of 32-column slices
    #   SAEV_AMD_FWD=sum_pass     slices, with the shares of a survivor added by a pass of their own (round 4) instead of by the final select
    #   SAEV_AMD_ENC_MFMA=32      single-product encoders on the 32x32x16 MFMA kernel
    #   SAEV_AMD_FUSED_CHAIN=1    f16r select -> refine -> select as one launch
    #   SAEV_AMD_NGROUPS=64       64-group TopK bound also for top_k <= 32
    #   SAEV_AMD_ENC_WGS, SAEV_AMD_REFRESH_FIRST, SAEV_AMD_REFRESH_EVERY   encoder grid / bound-refresh cadence
    #   SAEV_AMD_AUX_SMALL_MAX    largest dead set of the few-dead-latents AuxK kernels (-1: always the dense algebra)
    #   SAEV_AMD_DEAD_LAG         age in steps of the tracker record that sizes a step's auxiliary work (default 4)
    #   SAEV_AMD_CSC              1: the backward's pair-list build fills its bit map itself (default: the training decode does)
    #   SAEV_AMD_FIN              1: the backward's finalize re-reads the gradient rows for their statistics (round-4 kernels)
    #   SAEV_AMD_PREP             1: every f16r forward prepares its operands from x and W_enc itself (no streamed preparation)
    #   SAEV_AMD_AUX_DENSE        1: the dense AuxK algebra selects with the round-4 kernels (radix select, fills, scatter, absmax)
    #   SAEV_AMD_AUX_SMALL        1: 9-32 dead latents on the vector-ALU kernels of rounds 3-4 instead of the fp32 MFMA ones
    dw_route: str = dataclasses.field(default_factory=lambda: os.environ.get("SAEV_AMD_DW", "slices"))
    fwd_route: str = dataclasses.field(default_factory=lambda: os.environ.get("SAEV_AMD_FWD", "default"))
    enc_mfma: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_ENC_MFMA", "0")))
    fused_chain: bool = dataclasses.field(default_factory=lambda: os.environ.get("SAEV_AMD_FUSED_CHAIN", "0") not in ("", "0"))
    ngroups: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_NGROUPS", "0")))
    enc_wgs: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_ENC_WGS", "0")))
    refresh_first: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_REFRESH_FIRST", "0")))
    refresh_every: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_REFRESH_EVERY", "0")))
    aux_small_max: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_AUX_SMALL_MAX", "0")))
    dead_lag: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_DEAD_LAG", "0")))
    csc_route: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_CSC", "0")))
    fin_route: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_FIN", "0")))
    prep_route: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_PREP", "0")))
    aux_dense_route: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_AUX_DENSE", "0")))
    aux_small_route: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_AUX_SMALL", "0")))
    own_check: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_OWN_CHECK", "0")))
    enc_rot: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_ENC_ROT", "0")))
    group_route: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_GROUP", "0")))
    aux_split_route: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_AUX_SPLIT", "0")))
    aux_wide_route: int = dataclasses.field(default_factory=lambda: int(os.environ.get("SAEV_AMD_AUX_WIDE", "0")))


@dataclasses.dataclass
class StepStats:
    mse: float
    aux: float
    l0: float
    l1: float
    grad_norm: float
    upper: float
    n_dead: int
    n_overflow_rows: int
    cand_max: int
    dense_route: int
    sse: float
    sum_sq: float

    @property
    def loss(self) -> float:
        return self.mse + self.aux


def flat_layout(cfg: EngineConfig) -> "_lib.SaevLayout":
    """Offsets of the four tensors in the flat buffers, their length and the per-rank chunk lengths (saev_layout)."""
    lay = _lib.SaevLayout()
    ccfg = _lib.SaevCfg(d_model=cfg.d_model, d_sae=cfg.d_sae, shard_world=cfg.shard_world)
    rc = _lib.load().saev_layout(C.byref(ccfg), C.byref(lay))
    if rc != 0:
        raise _lib.SaevError(f"saev_layout failed with status {rc} for {cfg}")
    return lay


def _ptr(t: torch.Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SaeEngine:
    def __init__(self, cfg: EngineConfig, device: torch.device | str | int = "cuda", *, with_optim: bool = True):
        if not torch.cuda.is_available():
            raise _lib.SaevError("saev_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device if not isinstance(device, int) else f"cuda:{device}")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        S, D = cfg.d_sae, cfg.d_model
        lay = flat_layout(cfg)
        self.n_params = lay.n_total  # floats per flat buffer (zero padding included when shard_world > 1)
        self.shard_world = cfg.shard_world
        self.chunk_a, self.chunk_b = lay.chunk_a, lay.chunk_b
        self.offsets = {"W_dec": lay.off_W_dec, "b_dec": lay.off_b_dec, "W_enc": lay.off_W_enc, "b_enc": lay.off_b_enc}
        self.shapes = {"W_dec": (S, D), "b_dec": (D,), "W_enc": (D, S), "b_enc": (S,)}
        if cfg.bounds not in ("guaranteed", "predicted"):
            raise ValueError(f"EngineConfig.bounds (SAEV_AMD_BOUNDS) must be 'guaranteed' or 'predicted', got {cfg.bounds!r}")
        if cfg.encoder not in ("f32", "f16x3", "bf16", "f16r"):
            raise ValueError(f"EngineConfig.encoder (SAEV_AMD_ENCODER) must be one of f32, f16x3, bf16, f16r, got {cfg.encoder!r}")
        with torch.cuda.device(self.device):
            self.params = torch.zeros(self.n_params, device=self.device, dtype=torch.float32)
            self.grads = torch.zeros_like(self.params) if with_optim else None
            self.adam_m = torch.zeros_like(self.params) if with_optim else None
            self.adam_v = torch.zeros_like(self.params) if with_optim else None
            self.toks_since_active = torch.zeros(S, device=self.device, dtype=torch.int64)
            self.fired = torch.zeros(S, device=self.device, dtype=torch.int32)
            ccfg = _lib.SaevCfg(
                d_model=D, d_sae=S, top_k=cfg.top_k, k_aux=cfg.k_aux, alpha=cfg.alpha,
                dead_threshold_tokens=cfg.dead_threshold_tokens,
                normalize_w_dec=int(cfg.normalize_w_dec), remove_parallel_grads=int(cfg.remove_parallel_grads),
                max_batch=cfg.max_batch, encoder_mode={"f32": 0, "f16x3": 1, "bf16": 2, "f16r": 3}[cfg.encoder],
                aux_dead_cap=cfg.aux_dead_cap, shard_world=cfg.shard_world,
                bound_mode={"guaranteed": 0, "predicted": 1}[cfg.bounds], max_backward_rows=cfg.max_backward_rows,
            )
            if cfg.dw_route not in ("slices", "rows", "slices_a", "slices_s") or cfg.fwd_route not in ("default", "rows", "sum_pass"):
                raise ValueError(f"EngineConfig.dw_route must be 'slices', 'slices_a', 'slices_s' or 'rows' and fwd_route 'default', 'rows' or 'sum_pass', got {cfg.dw_route!r} / {cfg.fwd_route!r}")
            dbg = _lib.SaevDebugCfg(
                struct_size=C.sizeof(_lib.SaevDebugCfg), dw_route={"slices": 0, "rows": 1, "slices_a": 2, "slices_s": 4}[cfg.dw_route], enc_mfma=cfg.enc_mfma,
                fused_chain=int(cfg.fused_chain), ngroups=cfg.ngroups, enc_wgs=cfg.enc_wgs, refresh_first=cfg.refresh_first,
                refresh_every=cfg.refresh_every, aux_small_max=cfg.aux_small_max, fwd_route={"default": 0, "rows": 1, "sum_pass": 2}[cfg.fwd_route],
                dead_lag=cfg.dead_lag, csc_route=cfg.csc_route, fin_route=cfg.fin_route, prep_route=cfg.prep_route, aux_dense_route=cfg.aux_dense_route, aux_small_route=cfg.aux_small_route,
                own_check=cfg.own_check, enc_rot=cfg.enc_rot, group_route=cfg.group_route, aux_split_route=cfg.aux_split_route, aux_wide_route=cfg.aux_wide_route)
            ctx = C.c_void_p()
            rc = self.lib.saev_create_ex(C.byref(ccfg), C.byref(dbg), self.device.index, C.byref(ctx))
            if rc != 0:
                raise _lib.SaevError(f"saev_create failed with status {rc} for {cfg}")
            self.ctx = ctx
            self._chk(self.lib.saev_bind(ctx, _ptr(self.params), _ptr(self.grads), _ptr(self.adam_m), _ptr(self.adam_v)), "saev_bind")
            self._params_version = self._pversion()
            self._chk(self.lib.saev_bind_tracker(ctx, _ptr(self.toks_since_active), _ptr(self.fired)), "saev_bind_tracker")
            # the tail's sum of squares lives in a torch tensor from the start, so that a collective can reach it
            self.sumsq = torch.zeros(1, device=self.device, dtype=torch.float64)
            self._chk(self.lib.saev_bind_sumsq(ctx, _ptr(self.sumsq)), "saev_bind_sumsq")
        self.adam_steps = 0
        self._x_keepalive = None
        self._w_enc_t = None

    # ---- plumbing -------------------------------------------------------------------------
    def _chk(self, rc, what):
        _lib.check(self.lib, self.ctx, rc, what)

    def close(self):
        if getattr(self, "ctx", None):
            if getattr(self, "_leader", None) is not None and getattr(self._leader, "ctx", None):
                self.lib.saev_share_x(self.ctx, None)
            self.lib.saev_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def view(self, name: str, flat: torch.Tensor | None = None) -> torch.Tensor:
        flat = self.params if flat is None else flat
        off, shape = self.offsets[name], self.shapes[name]
        return flat[off : off + math.prod(shape)].view(shape)

    def param_views(self) -> dict[str, torch.Tensor]:
        return {k: self.view(k) for k in self.offsets}

    def grad_views(self) -> dict[str, torch.Tensor]:
        return {k: self.view(k, self.grads) for k in self.offsets}

    def set_tracker(self, toks: torch.Tensor | None) -> None:
        """Overwrite the dead-latent tracker (``None`` zeroes it) and tell the context it changed."""
        if toks is None:
            self.toks_since_active.zero_()
        else:
            self.toks_since_active.copy_(toks.to(self.device, torch.int64))
        self._chk(self.lib.saev_tracker_touched(self.ctx), "saev_tracker_touched")

    def set_prefixes(self, prefixes) -> None:
        """Matryoshka cut points for the following steps (ascending, last == d_sae); None / one entry = plain."""
        if prefixes is None:
            self._n_prefixes = 1
            self._chk(self.lib.saev_set_prefixes(self.ctx, None, 0), "saev_set_prefixes")
            return
        pre = [int(p) for p in prefixes]
        self._n_prefixes = max(1, len(pre))
        arr = (C.c_int64 * len(pre))(*pre)
        self._chk(self.lib.saev_set_prefixes(self.ctx, arr, len(pre)), "saev_set_prefixes")

    def share_x(self, leader: "SaeEngine | None") -> None:
        """Borrow what a step derives from x alone (statistics, centring, operand images) from ``leader`` whenever it has
        just run its forward on the same batch tensor: several SAEs on the same batches (train()'s parallel groups)."""
        self._leader = leader  # keeps it alive for as long as the link exists
        self._chk(self.lib.saev_share_x(self.ctx, leader.ctx if leader is not None else None), "saev_share_x")

    def load_params(self, params: dict[str, torch.Tensor]) -> None:
        for k in self.offsets:
            self.view(k).copy_(params[k].to(self.device, torch.float32))
        self.params_touched()

    def params_touched(self) -> None:
        """Tell the context that the parameter buffer was written from outside the library (include/saev_amd.h: PARAMETER
        OWNERSHIP): it drops what it keeps of W_enc / W_dec between calls.  In-place torch operations on ``params`` or on views of
        it are noticed by themselves (torch's version counter, checked before every forward); writes that bypass it -- ``.data``,
        raw pointers, another library -- need this call."""
        self._chk(self.lib.saev_params_touched(self.ctx), "saev_params_touched")
        self._params_version = self._pversion()

    def _pversion(self):
        try:
            return self.params._version
        except RuntimeError:  # a buffer created under torch.inference_mode() has no version counter: nothing to go by
            return None

    def watch(self, tensors) -> None:
        """Tensors that alias the parameter buffer but carry version counters of their own -- the four Parameters of a
        ``SparseAutoencoder`` bound to this engine (``p.data = view`` keeps the Parameter's counter).  Every entry point that
        reads the parameters compares their counters too, so an in-place write through the module (``sae.W_enc.mul_()``,
        ``load_state_dict``, an initialiser) is noticed even when the caller drives the engine directly, as ``train()`` does."""
        import weakref

        self._watched = [weakref.ref(t) for t in tensors]
        self._watched_versions = self._wversions()

    def _wversions(self):
        out = []
        for r in getattr(self, "_watched", ()):
            t = r()
            try:
                out.append(None if t is None else t._version)
            except RuntimeError:  # inference tensors carry no version counter
                out.append(-1)
        return out

    def _note_param_writes(self) -> None:
        v = self._pversion()
        w = self._wversions()
        if v is None or v != self._params_version or w != getattr(self, "_watched_versions", []) or -1 in w:
            self.params_touched()
            self._watched_versions = w

    def _check_x(self, x: torch.Tensor) -> torch.Tensor:
        if x.device != self.device or x.dtype != torch.float32:
            raise _lib.SaevError(f"activations must be float32 on {self.device}, got {x.dtype} on {x.device}")
        if x.ndim != 2 or x.shape[1] != self.cfg.d_model:
            raise _lib.SaevError(f"activations must be (n, {self.cfg.d_model}), got {tuple(x.shape)}")
        return x.contiguous()

    # ---- single ops -----------------------------------------------------------------------
    def normalize_w_dec(self):
        self._chk(self.lib.saev_normalize_w_dec(self.ctx, _stream()), "saev_normalize_w_dec")

    def encode_dense(self, x: torch.Tensor) -> torch.Tensor:
        x = self._check_x(x)
        h = torch.empty(x.shape[0], self.cfg.d_sae, device=self.device, dtype=torch.float32)
        self._chk(self.lib.saev_encode_dense(self.ctx, _ptr(x), x.shape[0], _ptr(h), _stream()), "saev_encode_dense")
        return h

    def topk_dense(self, h: torch.Tensor, k: int, mask: torch.Tensor | None = None):
        h = h.contiguous()
        n = h.shape[0]
        idx = torch.empty(n, k, device=self.device, dtype=torch.int32)
        val = torch.empty(n, k, device=self.device, dtype=torch.float32)
        if mask is not None:
            mask = mask.to(self.device, torch.int32).contiguous()
        self._chk(self.lib.saev_topk_dense(self.ctx, _ptr(h), n, k, _ptr(mask), _ptr(idx), _ptr(val), _stream()), "saev_topk_dense")
        return idx, val

    def encode_topk(self, x: torch.Tensor):
        x = self._check_x(x)
        self._note_param_writes()
        n, k = x.shape[0], min(self.cfg.top_k, self.cfg.d_sae)
        idx = torch.empty(n, k, device=self.device, dtype=torch.int32)
        val = torch.empty(n, k, device=self.device, dtype=torch.float32)
        self._chk(self.lib.saev_encode_topk(self.ctx, _ptr(x), n, _ptr(idx), _ptr(val), _stream()), "saev_encode_topk")
        return idx, val

    def scatter_dense(self, idx: torch.Tensor, val: torch.Tensor) -> torch.Tensor:
        n, k = idx.shape
        f = torch.zeros(n, self.cfg.d_sae, device=self.device, dtype=torch.float32)
        self._chk(self.lib.saev_scatter_dense(self.ctx, _ptr(idx.contiguous()), _ptr(val.contiguous()), n, k, _ptr(f), _stream()), "saev_scatter_dense")
        return f

    def decode_sparse(self, idx: torch.Tensor, val: torch.Tensor, prefixes=None) -> torch.Tensor:
        n, k = idx.shape
        if prefixes is None:
            pre = [self.cfg.d_sae]
        else:
            pre = [int(p) for p in prefixes]
        arr = (C.c_int64 * len(pre))(*pre)
        out = torch.empty(n, len(pre), self.cfg.d_model, device=self.device, dtype=torch.float32)
        self._chk(self.lib.saev_decode_sparse(self.ctx, _ptr(idx.contiguous()), _ptr(val.contiguous()), n, k, arr, len(pre), _ptr(out), _stream()), "saev_decode_sparse")
        return out

    def remove_parallel_grads(self):
        self._chk(self.lib.saev_remove_parallel_grads(self.ctx, _stream()), "saev_remove_parallel_grads")

    def gather_rows(self, pool: torch.Tensor, rows: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        n = rows.shape[0]
        if out is None:
            out = torch.empty(n, self.cfg.d_model, device=self.device, dtype=torch.float32)
        self._chk(self.lib.saev_gather_rows(self.ctx, _ptr(pool), _ptr(rows), n, _ptr(out), _stream()), "saev_gather_rows")
        return out

    # ---- the step -------------------------------------------------------------------------
    def step_forward(self, x: torch.Tensor, *, training: bool = True, n_rows_global: int | None = None):
        x = self._check_x(x)
        self._x_keepalive = x
        self._note_param_writes()
        n = x.shape[0]
        self._chk(self.lib.saev_step_forward(self.ctx, _ptr(x), n, n_rows_global or n, int(training), _stream()), "saev_step_forward")

    def step_dead(self, n_rows_global: int):
        self._chk(self.lib.saev_step_dead(self.ctx, n_rows_global, _stream()), "saev_step_dead")

    def step_backward(self):
        self._chk(self.lib.saev_step_backward(self.ctx, _stream()), "saev_step_backward")

    # backward in pieces (data-parallel overlap, see framework/ddp.py)
    def grad_w_enc_t(self) -> torch.Tensor:
        """(d_sae, d_model) transposed W_enc gradient the ranged backward writes; allocated on first use and handed to
        the context so that collectives can run on it."""
        if self._w_enc_t is None:
            self._w_enc_t = torch.zeros(self.cfg.d_sae, self.cfg.d_model, device=self.device, dtype=torch.float32)
            self._chk(self.lib.saev_bind_w_enc_t(self.ctx, _ptr(self._w_enc_t)), "saev_bind_w_enc_t")
        return self._w_enc_t

    def backward_begin(self):
        self._chk(self.lib.saev_backward_begin(self.ctx, _stream()), "saev_backward_begin")

    def backward_rows(self, lo: int, hi: int, part: int = 0):
        """Gradient rows of the latents [lo, hi).  part 0: both matrices in one pass; 1: the decoder's only (after it the
        decoder half of the gradient -- W_dec and b_dec -- is final); 2: the encoder's (needs part 1 first)."""
        self._chk(self.lib.saev_backward_rows_part(self.ctx, lo, hi, part, _stream()), "saev_backward_rows_part")

    def backward_end(self):
        self._chk(self.lib.saev_backward_end(self.ctx, _stream()), "saev_backward_end")

    def step_tail(self, lr: float, max_norm: float = 1.0, grad_scale: float = 1.0, *, trusted: bool = False):
        """``trusted``: nothing wrote the gradient buffer since ``backward_end`` -- the tail may use the row statistics the
        backward left behind (projection inside Adam, no rpg pass), as ``train_step`` does."""
        self.adam_steps += 1
        if trusted:
            self._chk(self.lib.saev_trust_gradients(self.ctx, 1), "saev_trust_gradients")
        try:
            self._chk(self.lib.saev_step_tail(self.ctx, lr, max_norm, grad_scale, self.adam_steps, _stream()), "saev_step_tail")
        finally:
            if trusted:
                self.lib.saev_trust_gradients(self.ctx, 0)

    # ---- gathered backward (data-parallel runs that exchange the sparse step state instead of the gradient) -------------
    def gather_buffers(self, world: int, n_local: int):
        """(x_all, g_all, idx_all, val_all) for ``world`` ranks of ``n_local`` rows each, allocated once per shape."""
        P = getattr(self, "_n_prefixes", 1)  # Matryoshka: dL/dx_hat is P suffix-summed gradients per row
        key = (world, n_local, P)
        if getattr(self, "_gather_key", None) != key:
            n, D, K = world * n_local, self.cfg.d_model, min(self.cfg.top_k, self.cfg.d_sae)
            cap = max(self.cfg.max_batch, self.cfg.max_backward_rows)
            if n > cap:
                raise _lib.SaevError(f"gathered backward over {n} rows needs an engine with max_backward_rows >= {n} (the GLOBAL batch), "
                                     f"got {cap}")
            self._gather_bufs = (torch.empty(n, D, device=self.device), torch.empty(n, P * D, device=self.device),
                                 torch.empty(n, K, device=self.device, dtype=torch.int32), torch.empty(n, K, device=self.device))
            self._gather_key = key
        return self._gather_bufs

    def copy_step_state(self, n_rows: int, g_out: torch.Tensor, idx_out: torch.Tensor, val_out: torch.Tensor):
        """This rank's rows of dL/dx_hat and of the codes of the training forward in flight, into caller tensors."""
        self._chk(self.lib.saev_copy_step_state(self.ctx, n_rows, _ptr(g_out), _ptr(idx_out), _ptr(val_out), _stream()), "saev_copy_step_state")

    def backward_begin_gathered(self, x_all: torch.Tensor, g_all: torch.Tensor, idx_all: torch.Tensor, val_all: torch.Tensor):
        """``backward_begin`` over the rows of ALL ranks (rank-major); the following ``backward_rows`` cover them too."""
        assert x_all.is_contiguous() and g_all.is_contiguous() and idx_all.is_contiguous() and val_all.is_contiguous()
        self._gather_keepalive = (x_all, g_all, idx_all, val_all)
        self._chk(self.lib.saev_backward_override(self.ctx, _ptr(x_all), _ptr(g_all), _ptr(idx_all), _ptr(val_all), x_all.shape[0]),
                  "saev_backward_override")
        self.backward_begin()

    def aux_compact_export(self) -> torch.Tensor | None:
        """The auxiliary term's local gradient -- the dead latents' rows of dW_dec and dW_enc^T, their db_enc, its share of
        db_dec -- packed into one tensor (None when the step has no auxiliary work); sum it over ranks, then import."""
        rows = int(self.lib.saev_aux_compact_rows(self.ctx))
        if rows == 0:
            return None
        buf = torch.empty(rows * (2 * self.cfg.d_model + 1) + self.cfg.d_model, device=self.device)
        self._chk(self.lib.saev_aux_compact_export(self.ctx, _ptr(buf), _stream()), "saev_aux_compact_export")
        return buf

    def aux_compact_import(self, buf: torch.Tensor):
        self._chk(self.lib.saev_aux_compact_import(self.ctx, _ptr(buf), _stream()), "saev_aux_compact_import")

    # tail in two parts over this rank's chunks (data-parallel runs with a sharded tail, framework/ddp.py)
    def halves(self, flat: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
        """The [W_dec | b_dec | pad] and [W_enc | b_enc | pad] halves of a flat buffer (shard_world equal chunks each)."""
        a = self.chunk_a * self.cfg.shard_world
        return flat[:a], flat[a:]


    def tail_prepare(self, shard_rank: int = -1):
        self._chk(self.lib.saev_tail_prepare(self.ctx, shard_rank, _stream()), "saev_tail_prepare")

    def tail_apply(self, lr: float, max_norm: float = 1.0, grad_scale: float = 1.0, shard_rank: int = -1):
        self.adam_steps += 1
        self._chk(self.lib.saev_tail_apply(self.ctx, lr, max_norm, grad_scale, self.adam_steps, shard_rank, _stream()), "saev_tail_apply")

    def wdec_ready_after(self, event: "torch.cuda.Event | None"):
        """The next step_forward waits for ``event`` before it first touches W_dec (and renormalises W_dec there)."""
        self._wdec_event = event  # keep the handle alive until it has been consumed
        self._chk(self.lib.saev_wdec_ready_event(self.ctx, C.c_void_p(event.cuda_event) if event is not None else None),
                  "saev_wdec_ready_event")

    def wenc_ready_after(self, event: "torch.cuda.Event | None"):
        """The next forward prepares x first and waits for ``event`` only before it reads W_enc / b_enc."""
        self._wenc_event = event
        self._chk(self.lib.saev_wenc_ready_event(self.ctx, C.c_void_p(event.cuda_event) if event is not None else None),
                  "saev_wenc_ready_event")

    def train_step(self, x: torch.Tensor, lr: float, max_norm: float = 1.0):
        """Phases 1-4 on one GPU (reference train.py:332-460 loop body for one SAE).

        ``grad_views()`` is NOT a valid gradient afterwards: the W_enc gradient stays in the transposed scratch and the
        dW_dec rows are stored un-projected (the fused Adam projects them as it reads).  To look at gradients run the phases
        (``step_forward`` / ``step_dead`` / ``step_backward`` / ``step_tail``), as the log steps of ``train()`` do."""
        x = self._check_x(x)
        self._x_keepalive = x
        self._note_param_writes()
        self._chk(self.lib.saev_train_step(self.ctx, _ptr(x), x.shape[0], lr, max_norm, self.adam_steps + 1, _stream()), "saev_train_step")
        self.adam_steps += 1  # (counted once the step is enqueued: a refused call -- SAEV_STALE_PARAMS -- is not an optimizer step)

    # data parallel behind the C ABI (include/saev_amd.h: DATA PARALLEL): RCCL inside the library, two collectives per step
    def comm_unique_id(self) -> bytes:
        """128 bytes from ncclGetUniqueId: made on ONE rank, handed to all ranks (e.g. ``torch.distributed.broadcast_object_list``)."""
        buf = C.create_string_buffer(128)
        self._chk(self.lib.saev_comm_unique_id(buf), "saev_comm_unique_id")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        if len(unique_id) != 128:
            raise _lib.SaevError(f"a communicator id is 128 bytes, got {len(unique_id)}")
        self._chk(self.lib.saev_comm_init(self.ctx, C.create_string_buffer(unique_id, 128), rank, world), "saev_comm_init")

    def comm_world(self) -> int:
        return int(self.lib.saev_comm_world(self.ctx))

    def train_step_dp(self, x_local: torch.Tensor, lr: float, max_norm: float = 1.0):
        """One optimizer step on the global batch of which ``x_local`` is this rank's share (equal shares on all ranks): forward,
        all-reduce of the fired flags, AuxK + backward, all-reduce of the flat gradient, tail with the gradient averaged --
        all enqueued by ONE call into the library (saev_train_step_dp), RCCL on torch's current stream."""
        x = self._check_x(x_local)
        self._x_keepalive = x
        self._note_param_writes()
        self._chk(self.lib.saev_train_step_dp(self.ctx, _ptr(x), x.shape[0], lr, max_norm, self.adam_steps + 1, _stream()), "saev_train_step_dp")
        self.adam_steps += 1  # (counted once the step is enqueued: a refused call -- SAEV_STALE_PARAMS -- is not an optimizer step)

    def train_step_gather(self, pool: torch.Tensor, rows: torch.Tensor, lr: float, max_norm: float = 1.0, out: torch.Tensor | None = None) -> torch.Tensor:
        """``train_step`` on the batch ``pool[rows]``, drawn inside the step (saev_train_step_gather): the step's first kernel reads
        the pool rows and leaves the batch as a contiguous matrix -- returned -- on its way."""
        n = rows.shape[0]
        if pool.device != self.device or pool.dtype != torch.float32 or pool.ndim != 2 or pool.shape[1] != self.cfg.d_model or not pool.is_contiguous():
            raise _lib.SaevError(f"the pool must be a contiguous float32 (rows, {self.cfg.d_model}) matrix on {self.device}")
        if rows.device != self.device or rows.dtype != torch.int64 or not rows.is_contiguous():
            raise _lib.SaevError(f"rows must be contiguous int64 on {self.device}")
        if out is None:
            out = torch.empty(n, self.cfg.d_model, device=self.device, dtype=torch.float32)
        self._x_keepalive = (pool, rows, out)
        self._note_param_writes()
        self._chk(self.lib.saev_train_step_gather(self.ctx, _ptr(pool), _ptr(rows), _ptr(out), n, lr, max_norm, self.adam_steps + 1, _stream()),
                  "saev_train_step_gather")
        self.adam_steps += 1
        return out

    def read_stats(self) -> StepStats:
        st = _lib.SaevStepStats()
        self._chk(self.lib.saev_read_stats(self.ctx, C.byref(st), _stream()), "saev_read_stats")
        return StepStats(**{f: getattr(st, f) for f, _ in _lib.SaevStepStats._fields_})

    def last_codes(self, n_rows: int):
        k = min(self.cfg.top_k, self.cfg.d_sae)
        idx = torch.empty(n_rows, k, device=self.device, dtype=torch.int32)
        val = torch.empty(n_rows, k, device=self.device, dtype=torch.float32)
        x_hat = torch.empty(n_rows, self.cfg.d_model, device=self.device, dtype=torch.float32)
        self._chk(self.lib.saev_copy_last(self.ctx, n_rows, _ptr(idx), _ptr(val), _ptr(x_hat), _stream()), "saev_copy_last")
        return idx, val, x_hat

    def aux_route(self) -> int:
        """What the last step_dead did for the auxiliary loss: 0 nothing, 1 few-dead-latents kernels without reading
        n_dead back, 2 the same after a read-back, 3 dense algebra after a read-back."""
        return int(self.lib.saev_last_aux_route(self.ctx))

    def scratch_bytes(self, which: int = 0) -> int:
        """Device memory the context owns besides the four flat buffers: 0 all of it, 1 AuxK dead-set buffers, 2 Matryoshka blocks."""
        return int(self.lib.saev_scratch_bytes(self.ctx, which))

    def dead_readbacks(self) -> int:
        """Blocking reads of n_dead so far."""
        return int(self.lib.saev_dead_readbacks(self.ctx))

    def bound_state(self) -> dict:
        """z of the predicted bounds, launches that used them, how many had to be repeated, mean list length of the last."""
        z, mc = C.c_float(), C.c_float()
        n, r = C.c_int64(), C.c_int64()
        self._chk(self.lib.saev_bound_state(self.ctx, C.byref(z), C.byref(n), C.byref(r), C.byref(mc), _stream()), "saev_bound_state")
        return {"z": z.value, "launches": n.value, "repeats": r.value, "mean_candidates": mc.value}

    def enable_kernel_timing(self, on: bool = True):
        self._chk(self.lib.saev_enable_kernel_timing(self.ctx, int(on)), "saev_enable_kernel_timing")

    def encoder_ms(self) -> float:
        return float(self.lib.saev_last_encoder_ms(self.ctx))
