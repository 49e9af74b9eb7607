"""In-order activation feed for inference (replaces reference src/saev/data/ordered.py on the MI355X path).

The reference walks the cache one activation at a time in a manager process (one ``np.memmap`` + ``.copy()`` per
row, ordered.py:131-186) and ships batches through a queue.  Here a batch is a contiguous range of the global index
``g = example * content_tokens_per_example + token`` (shards.py:1042-1067), so it is cut out of at most a few
memory-mapped shards as whole strided blocks, staged through one pinned host buffer and copied to the device
asynchronously while the previous batch is being encoded.

Batches keep the reference's keys and dtypes (ordered.py:170-182): ``act (B, D) float32`` (on the device),
``example_idx (B,) int64``, ``token_idx (B,) int64`` and, when the cache has a ``labels.bin``
(``(n_examples, content_tokens_per_example) uint8``), ``token_labels (B,) int64`` (on the host).
"""

from __future__ import annotations

import dataclasses
import math
import os
import pathlib
import typing as tp

import numpy as np
import torch

from . import shards as shards_lib


@dataclasses.dataclass(frozen=True)
class Config:
    """Field names and defaults of ordered.py:46-69; queue tuning fields are accepted and ignored."""

    shards: pathlib.Path = pathlib.Path("$SAEV_SCRATCH/saev/shards/abcdefg")
    tokens: tp.Literal["content"] = "content"
    layer: int | tp.Literal["all"] = -2
    batch_size: int = 1024 * 16
    batch_timeout_s: float = 30.0
    drop_last: bool = False
    buffer_size: int = 64
    debug: bool = False
    log_every_s: float = 30.0


class DataLoader:
    def __init__(self, cfg: Config, *, device: torch.device | str = "cuda"):
        self.cfg = cfg
        root = pathlib.Path(os.path.expandvars(str(cfg.shards)))
        if not root.is_dir():
            raise RuntimeError(f"Activations are not saved at '{cfg.shards}'.")
        if cfg.tokens != "content" or not isinstance(cfg.layer, int):
            raise NotImplementedError("The ordered feed supports `content` tokens of one fixed `layer` (as the reference does).")
        self.root = root
        self.md = shards_lib.Metadata.load(root)
        assert cfg.layer in self.md.layers, f"Layer {cfg.layer} not in {self.md.layers}"
        self.layer_i = self.md.layers.index(cfg.layer)
        self.info = shards_lib.ShardInfo.load(root)
        self.info.validate(root, self.md)  # (reference ordered.py:226: every shard file checked before the first batch)
        for _, n_ex in self.info.shards[:-1]:
            assert n_ex == self.md.examples_per_shard, "all shards but the last hold examples_per_shard examples"
        self.device = torch.device(device)
        self.T = self.md.content_tokens_per_example
        self.first = 1 if self.md.cls_token else 0
        self._mm: dict[int, np.memmap] = {}
        lp = root / "labels.bin"
        self.labels = np.memmap(lp, mode="r", dtype=np.uint8, shape=(self.md.n_examples, self.T)) if lp.exists() else None

    # ---- reference properties (ordered.py:225-243) ----------------------------------------------------------------
    @property
    def n_samples(self) -> int:
        return self.md.n_examples * self.T

    @property
    def batch_size(self) -> int:
        return self.cfg.batch_size

    @property
    def drop_last(self) -> bool:
        return self.cfg.drop_last

    @property
    def n_batches(self) -> int:
        return len(self)

    def __len__(self) -> int:
        n, b = self.n_samples, self.cfg.batch_size
        return n // b if self.cfg.drop_last else math.ceil(n / b)

    def shutdown(self):
        self._mm.clear()

    # ---- reading ---------------------------------------------------------------------------------------------------
    def _shard(self, si: int) -> np.memmap:
        if si not in self._mm:
            name, n_ex = self.info.shards[si]
            self._mm[si] = shards_lib.open_shard(self.root, self.md, name, n_ex)
        return self._mm[si]

    def _read_range(self, g0: int, g1: int, out: np.ndarray) -> None:
        """Rows [g0, g1) of the global index into ``out`` ((g1-g0), D)."""
        T, eps = self.T, self.md.examples_per_shard
        pos = 0
        e0, e1 = g0 // T, (g1 - 1) // T  # first / last example touched
        for si in range(e0 // eps, e1 // eps + 1):
            mm = self._shard(si)
            lo = max(e0, si * eps)
            hi = min(e1, si * eps + mm.shape[0] - 1)
            block = mm[lo - si * eps : hi - si * eps + 1, self.layer_i, self.first :, :]  # (n_ex, T, D) strided view
            flat = np.ascontiguousarray(block).reshape(-1, self.md.d_model)
            a = max(g0, lo * T) - lo * T
            b = min(g1, (hi + 1) * T) - lo * T
            out[pos : pos + (b - a)] = flat[a:b]
            pos += b - a
        assert pos == g1 - g0

    def __iter__(self):
        n, B, T = self.n_samples, self.cfg.batch_size, self.T
        on_gpu = self.device.type == "cuda"
        # two pinned staging buffers: batch i+1 is read from disk while batch i's copy / compute is in flight
        stage = [torch.empty(B, self.md.d_model, dtype=torch.float32, pin_memory=on_gpu) for _ in range(2)]
        done: list[torch.cuda.Event | None] = [None, None]
        for bi, g0 in enumerate(range(0, n, B)):
            g1 = min(g0 + B, n)
            if g1 - g0 < B and self.cfg.drop_last:
                return
            buf = stage[bi & 1]
            if done[bi & 1] is not None:
                done[bi & 1].synchronize()  # the copy that last used this buffer has finished
            self._read_range(g0, g1, buf.numpy()[: g1 - g0])
            act = buf[: g1 - g0].to(self.device, non_blocking=True) if on_gpu else buf[: g1 - g0].clone()
            if on_gpu:
                done[bi & 1] = torch.cuda.Event()
                done[bi & 1].record()
            g = torch.arange(g0, g1, dtype=torch.int64)
            batch = {"act": act, "example_idx": g // T, "token_idx": g % T}
            if self.labels is not None:
                lab = self.labels.reshape(-1)[g0:g1]
                batch["token_labels"] = torch.from_numpy(np.asarray(lab).astype(np.int64))
            yield batch
