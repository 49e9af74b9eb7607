"""Device-resident shuffled activation feed (replaces reference src/saev/data/shuffled.py +
buffers.py on the hot path).

The reference streams shards through a manager process and I/O threads into a shared-memory
reservoir and pops random rows one by one under a lock (shuffled.py:131-376, buffers.py:179-216).
On an MI355X the whole working set fits in HBM (288 GB), so the pool lives on the device: shards are
memory-mapped, the selected layer / token slice is copied once into a (n_rows, d_model) device
tensor (this rank's share in data-parallel runs), and every epoch draws a seeded permutation and
serves batches with one HIP row-gather.  Semantics kept from the reference: every row is delivered
exactly once per epoch; batches are dicts ``{"act" (B,D) f32, "example_idx" (B,) i32, "token_idx"
(B,) i32}`` (shuffled.py:385-391); ``drop_last``; ``n_samples``; ``metadata``.

``Config`` keeps the reference's field names and defaults (shuffled.py:31-70); fields that only tune
the CPU reservoir (n_threads, buffer_size, ...) are accepted and ignored.
"""

from __future__ import annotations

import dataclasses
import math
import pathlib
import typing as tp

import numpy as np
import torch

from . import shards as shards_lib


@dataclasses.dataclass(frozen=True)
class Config:
    shards: pathlib.Path = pathlib.Path("$SAEV_SCRATCH/saev/shards/abcdefg")
    tokens: tp.Literal["special", "content", "all"] = "content"
    layer: int | tp.Literal["all"] = -1
    batch_size: int = 1024 * 16
    drop_last: bool = False
    scale_norm: bool = False
    ignore_labels: list[int] = dataclasses.field(default_factory=list)
    n_threads: int = 4
    buffer_size: int = 64
    min_buffer_fill: float = 0.0
    batch_timeout_s: float = 30.0
    seed: int = 17
    debug: bool = False
    log_every_s: float = 30.0
    use_tmpdir: bool = False


class DataLoader:
    """Iterable over shuffled batches of one epoch; re-iterable (a new permutation each epoch)."""

    def __init__(self, cfg: Config, *, device: torch.device | str = "cuda", rank: int = 0, world_size: int = 1,
                 pool: torch.Tensor | None = None, engine=None):
        self.cfg = cfg
        self.device = torch.device(device)
        self.rank, self.world = rank, world_size
        assert cfg.batch_size % world_size == 0, "global batch must divide evenly over ranks"
        self.local_batch = cfg.batch_size // world_size
        self.batch_size = cfg.batch_size
        self.drop_last = cfg.drop_last
        self.manager_pid = -1
        self.engine = engine
        self._epoch = 0
        if pool is not None:  # in-memory pool (tests, synthetic benchmarks)
            assert pool.ndim == 2
            self.metadata = shards_lib.Metadata(
                family="fake-clip", ckpt="in-memory", layers=(0,), content_tokens_per_example=1, cls_token=False,
                d_model=pool.shape[1], n_examples=pool.shape[0], max_tokens_per_shard=max(pool.shape[0], 1))
            self.n_samples = pool.shape[0]
            rows = torch.arange(rank, pool.shape[0], world_size)
            self.pool = pool[rows].to(self.device, torch.float32).contiguous()
            self.example_idx = rows.to(torch.int32).to(self.device)
            self.token_idx = torch.zeros_like(self.example_idx)
        else:
            self._load_shards()

    # -------------------------------------------------------------------------------------
    def _load_shards(self):
        cfg = self.cfg
        if cfg.ignore_labels:
            raise NotImplementedError("ignore_labels (patch filtering) is not supported by the device-resident feed yet")
        if cfg.scale_norm:
            raise NotImplementedError("scale_norm is not supported by the device-resident feed yet")
        d = pathlib.Path(cfg.shards)
        if not (d / "metadata.json").exists():
            raise FileNotFoundError(f"no metadata.json under {d}")
        md = shards_lib.Metadata.load(d)
        info = shards_lib.ShardInfo.load(d)
        self.metadata = md
        if cfg.layer == "all":
            layer_ids = list(range(len(md.layers)))
        else:
            if cfg.layer not in md.layers:
                raise ValueError(f"layer {cfg.layer} not in recorded layers {md.layers}")
            layer_ids = [md.layers.index(cfg.layer)]
        first = 1 if md.cls_token else 0
        if cfg.tokens == "content":
            tok = list(range(first, md.tokens_per_example))
        elif cfg.tokens == "special":
            if not md.cls_token:
                raise ValueError("tokens='special' but the cache has no CLS token")
            tok = [0]
        else:
            tok = list(range(md.tokens_per_example))
        self.n_samples = md.n_examples * len(tok) * len(layer_ids)
        # shard order: seeded permutation, as the reference's manager does (shuffled.py:327-328);
        # ranks take shards round-robin
        order = np.random.default_rng(cfg.seed).permutation(len(info))
        acts, exs, tks = [], [], []
        ex_base = np.cumsum([0] + [n for _, n in info.shards])
        for pos, si in enumerate(order):
            if pos % self.world != self.rank:
                continue
            name, n_ex = info.shards[si]
            mm = shards_lib.open_shard(d, md, name, n_ex)
            for li in layer_ids:
                block = np.ascontiguousarray(mm[:, li][:, tok])  # (n_ex, n_tok, D)
                acts.append(torch.from_numpy(block.reshape(-1, md.d_model)).to(self.device))
                ex = np.repeat((np.arange(n_ex) + ex_base[si]).astype(np.int32), len(tok))
                tk = np.tile((np.asarray(tok) - first * (cfg.tokens == "content")).astype(np.int32), n_ex)
                exs.append(torch.from_numpy(ex))
                tks.append(torch.from_numpy(tk))
        if not acts:
            raise ValueError(f"rank {self.rank} of {self.world} received no shards ({len(info)} shards in cache)")
        self.pool = torch.cat(acts).contiguous()
        self.example_idx = torch.cat(exs).to(self.device)
        self.token_idx = torch.cat(tks).to(self.device)

    # -------------------------------------------------------------------------------------
    @property
    def n_local(self) -> int:
        return self.pool.shape[0]

    def __len__(self) -> int:
        n = self.n_local
        return n // self.local_batch if self.drop_last else math.ceil(n / self.local_batch)

    def __iter__(self):
        # host-side permutation: the row order is then independent of the device type (tests replay it on CPU)
        g = torch.Generator().manual_seed(self.cfg.seed + 1000 * self._epoch + self.rank)
        self._epoch += 1
        perm = torch.randperm(self.n_local, generator=g).to(self.device)
        B = self.local_batch
        for lo in range(0, self.n_local, B):
            rows = perm[lo : lo + B]
            if rows.shape[0] < B and self.drop_last:
                return
            if self.engine is not None:
                act = self.engine.gather_rows(self.pool, rows)
            else:
                act = self.pool[rows]
            yield {"act": act, "example_idx": self.example_idx[rows], "token_idx": self.token_idx[rows]}
