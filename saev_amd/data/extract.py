"""Activation extraction handed straight to training (BASELINE.json configs[4]): a vision transformer's forward hooks
fill an HBM reservoir, the SAE train step draws its batches from it -- no shard files in between.

Reference behaviour this replaces, end to end: ``RecordedTransformer`` (data/shards.py:205-300: forward hooks on the
chosen residual blocks copy ``output[:, tokens, :]`` into a ``(batch, n_layers, tokens_per_example, d_model)`` buffer),
``worker_fn`` writing those blocks to disk shards (shards.py:697-850), and ``ShuffledDataLoader`` reading them back
through the shared-memory reservoir (data/shuffled.py, data/buffers.py).  Kept: the hook placement and token selection
(CLS first, then the content tokens), ``example_idx`` / ``token_idx`` of every row, every (example, token) delivered
exactly once per epoch in random order from a reservoir of ``buffer_size * batch_size`` rows, the batch dict.

MI355X design: the transformer runs on the same device as the SAE, so a recorded block never leaves HBM.  The hooks write
into one preallocated device tensor; the feed scatters the selected layer's ``(examples * tokens, d_model)`` rows into free
reservoir slots with one ``index_copy_`` and serves a batch as one row gather (the train step's own ``saev_gather_rows``).
Producer and consumer alternate on the caller's stream: forward passes are run until the reservoir holds ``min_fill`` of
its capacity (or the images run out), then batches are drawn until it falls below that again.  The transformer itself is
whatever ``torch.nn.Module`` the caller brings (stock PyTorch-ROCm; ``vit.VisionTransformer`` is a plain one for tests and
offline runs); only the hand-off is this package's.
"""

from __future__ import annotations

import dataclasses
import math
import typing as tp

import numpy as np
import torch

from . import shards as shards_lib


class ActivationRecorder(torch.nn.Module):
    """Forward hooks on ``blocks[i]`` for i in ``layers`` (reference RecordedTransformer, shards.py:205-300).

    ``blocks``: the residual blocks of ``model`` in order; each must output ``(batch, tokens, d_model)`` with the CLS token
    (if the model has one) first.  ``forward(images)`` returns ``(model output, cache)`` where cache is a
    ``(batch, len(layers), tokens_per_example, d_model)`` DEVICE tensor, valid until the next forward."""

    def __init__(self, model: torch.nn.Module, blocks: tp.Sequence[torch.nn.Module], layers: tp.Sequence[int],
                 content_tokens_per_example: int, cls_token: bool, model_has_cls: bool = True):
        super().__init__()
        self.model = model
        self.layers = tuple(int(i) for i in layers)
        self.content_tokens_per_example = content_tokens_per_example
        self.cls_token = cls_token
        self.model_has_cls = model_has_cls
        self._storage: torch.Tensor | None = None
        self._i = 0
        self._handles = [blocks[i].register_forward_hook(self._hook) for i in self.layers]

    @property
    def tokens_per_example(self) -> int:
        return self.content_tokens_per_example + int(self.cls_token)

    def _hook(self, module, args, output) -> None:
        out = output[0] if isinstance(output, tuple) else output
        first = 1 if (self.model_has_cls and not self.cls_token) else 0
        sel = out[:, first : first + self.tokens_per_example, :]
        assert sel.shape[1] == self.tokens_per_example, (
            f"Shape mismatch: got {sel.shape[1]} tokens, expected {self.tokens_per_example} "
            f"(content_tokens_per_example={self.content_tokens_per_example}, cls_token={self.cls_token})")
        b, _, d = sel.shape
        if self._storage is None or self._storage.shape[0] != b or self._storage.shape[3] != d or self._storage.device != sel.device:
            self._storage = torch.empty(b, len(self.layers), self.tokens_per_example, d, device=sel.device, dtype=torch.float32)
        self._storage[:, self._i].copy_(sel.detach())
        self._i += 1

    def forward(self, batch: torch.Tensor, **kwargs):
        self._i = 0
        out = self.model(batch, **kwargs)
        assert self._i == len(self.layers), f"{self._i} of {len(self.layers)} hooked blocks ran"
        return out, self._storage

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


class DeviceReservoir:
    """Fixed-capacity row store in HBM with host-side slot bookkeeping; ``put`` places a device block into free slots,
    ``get`` draws rows uniformly without replacement from the filled ones and frees them.  Everything runs on the
    caller's stream, so a slot freed by a ``get`` is only ever overwritten by a later ``put`` behind the gather."""

    def __init__(self, capacity: int, d_model: int, device: torch.device, seed: int, gather=None):
        self.capacity, self.D, self.device = capacity, d_model, torch.device(device)
        self.rows = torch.empty(capacity, d_model, dtype=torch.float32, device=self.device)
        self.meta = torch.empty(capacity, 2, dtype=torch.int32, device=self.device)
        self.rng = np.random.default_rng(seed)
        self.gather = gather
        self.reset()

    def reset(self):
        self._free = np.arange(self.capacity, dtype=np.int64)[::-1].copy()
        self._n_free = self.capacity
        self._filled = np.empty(self.capacity, dtype=np.int64)
        self._n_filled = 0

    def fill(self) -> float:
        return self._n_filled / self.capacity

    def room(self) -> int:
        return self._n_free

    def put(self, act: torch.Tensor, example_idx: torch.Tensor, token_idx: torch.Tensor) -> None:
        n = act.shape[0]
        if n > self.capacity:
            raise ValueError(f"a block of {n} rows cannot enter a reservoir of {self.capacity}: raise buffer_size")
        if n > self._n_free:
            raise ValueError(f"reservoir overflow: {n} rows offered, room for {self._n_free} (check room() before put())")
        slots = self._free[self._n_free - n : self._n_free].copy()
        self._n_free -= n
        self._filled[self._n_filled : self._n_filled + n] = slots
        self._n_filled += n
        st = torch.from_numpy(slots).to(self.device)
        self.rows.index_copy_(0, st, act.to(torch.float32))
        self.meta.index_copy_(0, st, torch.stack([example_idx.to(torch.int32), token_idx.to(torch.int32)], dim=1).to(self.device))

    def get(self, b: int):
        n = self._n_filled
        b = min(b, n)
        if b == 0:
            return None
        pick = self.rng.permutation(n)[:b] if n < 4 * b else self.rng.choice(n, size=b, replace=False)
        slots = self._filled[pick].copy()
        keep = np.ones(n, dtype=bool)
        keep[pick] = False
        self._filled[: n - b] = self._filled[:n][keep]
        self._n_filled = n - b
        self._free[self._n_free : self._n_free + b] = slots
        self._n_free += b
        st = torch.from_numpy(slots).to(self.device)
        act = self.gather(self.rows, st) if self.gather is not None else self.rows[st]
        meta = self.meta[st]
        return act, meta[:, 0].contiguous(), meta[:, 1].contiguous()


@dataclasses.dataclass(frozen=True)
class ExtractConfig:
    """What of the recorded activations goes to the SAE, with ShuffledConfig's names where they mean the same
    (reference data/shuffled.py:31-70)."""

    layer: int = -1                     # a member of the recorder's ``layers``
    tokens: tp.Literal["special", "content", "all"] = "content"
    batch_size: int = 1024 * 16         # SAE batch (rows); the global batch under data parallelism
    drop_last: bool = False
    buffer_size: int = 64               # reservoir capacity in batches
    min_buffer_fill: float = 0.5        # forward passes run until the reservoir is this full before a batch is drawn
    seed: int = 17


class ExtractionFeed:
    """Loader-shaped object (``batch_size``, ``drop_last``, ``n_samples``, ``metadata``, ``reservoir``, iteration yields
    ``{"act", "example_idx", "token_idx"}``) whose rows come out of ``recorder`` on the fly.

    ``images``: a callable returning an iterator of ``(image_batch (b, ...), example_idx (b,) int)`` pairs -- one pass over
    the dataset; called once per epoch.  Under data parallelism every rank passes its own share of the examples and
    ``n_examples`` is the global count (rank r of w must be given examples so that all ranks see equally many)."""

    def __init__(self, cfg: ExtractConfig, recorder: ActivationRecorder, images: tp.Callable[[], tp.Iterator], *, n_examples: int,
                 d_model: int, device: torch.device | str = "cuda", rank: int = 0, world_size: int = 1, engine=None,
                 family: str = "vit", ckpt: str = "in-process"):
        self.cfg, self.recorder, self.images = cfg, recorder, images
        self.device = torch.device(device)
        self.rank, self.world = rank, world_size
        assert cfg.batch_size % world_size == 0, "global batch must divide evenly over ranks"
        self.batch_size, self.drop_last = cfg.batch_size, cfg.drop_last
        self.local_batch = cfg.batch_size // world_size
        self.engine = engine
        self.manager_pid = -1
        if cfg.layer not in recorder.layers:
            raise ValueError(f"layer {cfg.layer} not in recorded layers {recorder.layers}")
        self._li = recorder.layers.index(cfg.layer)
        first = 1 if recorder.cls_token else 0
        T = recorder.tokens_per_example
        if cfg.tokens == "content":
            self._tok = list(range(first, T))
        elif cfg.tokens == "special":
            if not recorder.cls_token:
                raise ValueError("tokens='special' but the recorder keeps no CLS token")
            self._tok = [0]
        else:
            self._tok = list(range(T))
        self._tok_out = torch.tensor([t - first * (cfg.tokens == "content") for t in self._tok], dtype=torch.int32)
        self.metadata = shards_lib.Metadata(
            family=family, ckpt=ckpt, layers=recorder.layers, content_tokens_per_example=recorder.content_tokens_per_example,
            cls_token=recorder.cls_token, d_model=d_model, n_examples=n_examples,
            max_tokens_per_shard=max(n_examples * T * len(recorder.layers), 1))
        self.n_samples = n_examples * len(self._tok)
        self.n_epoch = self.n_samples // world_size  # rows this rank delivers per epoch (equal shares assumed)
        cap = max(cfg.buffer_size * self.local_batch, 2 * self.local_batch)
        self.reservoir = DeviceReservoir(cap, d_model, self.device, cfg.seed + 7919 * rank, gather=self._gather)
        self._block_rows = 0  # rows one forward pass produces (known after the first)

    def _gather(self, pool, rows):
        return self.engine.gather_rows(pool, rows) if self.engine is not None and pool.is_cuda else pool[rows]

    def __len__(self) -> int:
        n = self.n_epoch
        return n // self.local_batch if self.drop_last else math.ceil(n / self.local_batch)

    @torch.no_grad()
    def _produce(self, it) -> bool:
        """One forward pass -> reservoir; False when the images are exhausted."""
        try:
            imgs, ex = next(it)
        except StopIteration:
            return False
        _, cache = self.recorder(imgs.to(self.device, non_blocking=True))
        b = cache.shape[0]
        tok = torch.as_tensor(self._tok, device=cache.device)
        act = cache[:, self._li].index_select(1, tok).reshape(b * len(self._tok), -1)
        ex = torch.as_tensor(ex, dtype=torch.int32)
        self._block_rows = act.shape[0]
        self.reservoir.put(act, ex.repeat_interleave(len(self._tok)), self._tok_out.repeat(b))
        return True

    def __iter__(self):
        res = self.reservoir
        res.reset()
        it = iter(self.images())
        more = True
        left = self.n_epoch
        want = max(self.local_batch, int(self.cfg.min_buffer_fill * res.capacity))
        while left > 0:
            while more and res._n_filled < want and res.room() >= max(self._block_rows, 1):
                more = self._produce(it)
                if self._block_rows > res.capacity:
                    raise ValueError("one forward pass produces more rows than the reservoir holds: raise buffer_size")
            if res._n_filled == 0:
                if not more:
                    # the images ran out before this rank delivered its share: under data parallelism the other ranks would
                    # wait in the step's collectives forever -- fail loudly instead of ending the epoch early
                    if self.world > 1:
                        raise RuntimeError(f"rank {self.rank}: images exhausted with {left} of {self.n_epoch} rows of the epoch "
                                           "still to deliver; every rank must be given examples for n_examples / world_size")
                    return
                if res.room() < self._block_rows:
                    raise RuntimeError("reservoir cannot take another block and holds nothing to draw")
                continue
            asked = min(self.local_batch, left)
            got = res.get(asked)
            act, ex, tk = got
            if act.shape[0] < asked and not more and self.world > 1:
                # fewer rows than asked for: the images ran out mid-epoch on this rank only -- the same hang as above
                raise RuntimeError(f"rank {self.rank}: images exhausted with {left} of {self.n_epoch} rows of the epoch still to deliver "
                                   f"(a draw of {asked} rows returned {act.shape[0]}); every rank must be given examples for n_examples / world_size")
            if act.shape[0] < self.local_batch and self.drop_last:
                return  # (the epoch's last, partial batch: n_epoch is the same on every rank, so all ranks stop here together)
            left -= act.shape[0]
            yield {"act": act, "example_idx": ex, "token_idx": tk}
