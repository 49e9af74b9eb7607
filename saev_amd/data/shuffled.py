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

Caches larger than the device budget (``SAEV_AMD_RESIDENT_GB``, default half of the device memory) are
streamed instead: an HBM reservoir of ``buffer_size * batch_size`` rows (the reference's capacity,
shuffled.py:45-63) is refilled from the shards by a background reader and batches are uniform draws from
it -- see reservoir.py.  ``min_buffer_fill`` and ``batch_timeout_s`` apply to that mode.

``Config`` keeps the reference's field names and defaults (shuffled.py:31-70); ``n_threads`` is the number
of reader threads of the streaming mode.
"""

from __future__ import annotations

import dataclasses
import math
import os
import pathlib
import threading
import typing as tp
import warnings

import numpy as np
import torch

from . import shards as shards_lib
from .reservoir import StreamingReservoir


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


class _DeferredBatch(dict):
    """A batch whose rows the consumer draws itself (``defer_gather``): ``{"act": None, "rows", "pool"}`` plus the cache indices
    ``example_idx`` / ``token_idx`` of the rows, which are gathered only when somebody asks for them -- the train loop does on
    its log steps (two indexing kernels and their dispatch per step otherwise, for values nobody reads)."""

    _LAZY = ("example_idx", "token_idx")

    def __init__(self, loader, rows):
        super().__init__(act=None, rows=rows, pool=loader.pool)
        self._loader = loader

    def __missing__(self, key):
        if key in self._LAZY:
            value = getattr(self._loader, key)[self["rows"]]
            self[key] = value
            return value
        raise KeyError(key)

    def __contains__(self, key):
        return key in self._LAZY or super().__contains__(key)

    def get(self, key, default=None):
        return self[key] if key in self else default


class DataLoader:
    """Iterable over shuffled batches of one epoch; re-iterable (a new permutation each epoch)."""

    def __init__(self, cfg: Config, *, device: torch.device | str = "cuda", rank: int = 0, world_size: int = 1,
                 pool: torch.Tensor | None = None, engine=None, resident: bool | None = None):
        self.cfg = cfg
        self.reservoir: StreamingReservoir | None = None
        self._resident_arg = resident
        self.device = torch.device(device)
        self.rank, self.world = rank, world_size
        assert cfg.batch_size % world_size == 0, "global batch must divide evenly over ranks"
        self.local_batch = cfg.batch_size // world_size
        self.batch_size = cfg.batch_size
        self.drop_last = cfg.drop_last
        self.manager_pid = -1
        self.engine = engine
        self.defer_gather = False  # resident mode: yield (pool, rows) and leave the row gather to the consumer's train step
        self._epoch = 0
        self._fds: dict[str, int] = {}
        self._fd_lock = threading.Lock()
        if pool is not None:  # in-memory pool (tests, synthetic benchmarks)
            assert pool.ndim == 2
            self.metadata = shards_lib.Metadata(
                family="fake-clip", ckpt="in-memory", layers=(0,), content_tokens_per_example=1, cls_token=False,
                d_model=pool.shape[1], n_examples=pool.shape[0], max_tokens_per_shard=max(pool.shape[0], 1))
            self.n_samples = pool.shape[0]
            rows = torch.arange(rank, pool.shape[0], world_size)
            self._n_epoch = pool.shape[0] // world_size  # the same on every rank (see n_epoch)
            self.pool = pool[rows].to(self.device, torch.float32).contiguous()
            self.example_idx = rows.to(torch.int32).to(self.device)
            self.token_idx = torch.zeros_like(self.example_idx)
        else:
            self._load_shards()

    # -------------------------------------------------------------------------------------
    def _load_shards(self):
        cfg = self.cfg
        if cfg.scale_norm:
            raise NotImplementedError("scale_norm not implemented.")  # nor in the reference (shuffled.py:414-415)
        d = pathlib.Path(os.path.expandvars(str(cfg.shards)))
        if not (d / "metadata.json").exists():
            raise FileNotFoundError(f"no metadata.json under {d}")
        md = shards_lib.Metadata.load(d)
        info = shards_lib.ShardInfo.load(d)
        # every shard file checked up front, all problems in ONE error (reference shuffled.py:421 -> shards.py:638-694): a bad
        # shard must not surface mid-epoch as "reservoir reader failed"
        info.validate(d, md)
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
        # patch-label filtering (shuffled.py:207-216, 636-693): content tokens of one layer whose label is not ignored
        labels = None
        if cfg.ignore_labels:
            if cfg.tokens != "content" or not isinstance(cfg.layer, int):
                raise NotImplementedError("Patch label filtering only supports 'content' patches with fixed layer")
            if not (d / "labels.bin").exists():
                raise FileNotFoundError(f"ignore_labels filtering requested but labels.bin not found at {d / 'labels.bin'}")
            labels = np.memmap(d / "labels.bin", mode="r", dtype=np.uint8, shape=(md.n_examples, md.content_tokens_per_example))
            keep_all = ~np.isin(np.asarray(labels), cfg.ignore_labels)
            self.n_samples = int(keep_all.sum())
        # shard order: seeded permutation, as the reference's manager does (shuffled.py:327-328);
        # ranks take shards round-robin
        order = np.random.default_rng(cfg.seed).permutation(len(info))
        if len(info) < self.world:
            raise ValueError(f"rank {self.rank} of {self.world} received no shards ({len(info)} shards in cache)")
        ex_base = np.cumsum([0] + [n for _, n in info.shards])
        rows_per_example = len(tok) * len(layer_ids)

        def rows_of_rank(r: int) -> int:
            sis = [int(si) for pos, si in enumerate(order) if pos % self.world == r]
            if labels is None:
                return sum(info.shards[si][1] for si in sis) * rows_per_example
            return int(sum(keep_all[ex_base[si] : ex_base[si] + info.shards[si][1]].sum() for si in sis))

        mine = [int(si) for pos, si in enumerate(order) if pos % self.world == self.rank]
        self._n_local = rows_of_rank(self.rank)
        # every rank can work out every rank's share: an epoch is cut at the smallest one so that all ranks take the
        # same number of steps with the same batch sizes (the surplus rows of the larger shares are the ones the
        # epoch's permutation puts last, so they differ from epoch to epoch)
        self._n_epoch = min(rows_of_rank(r) for r in range(self.world))
        tok_arr = np.asarray(tok)
        tok_contiguous = tok == list(range(tok[0], tok[0] + len(tok)))
        tok_out = (tok_arr - first * (cfg.tokens == "content")).astype(np.int32)

        def blocks(epoch: int, max_rows: int | None):
            """(act (n, D), example_idx (n,), token_idx (n,)) host blocks of this rank's shards, in the epoch's order."""
            seq = mine if epoch == 0 else [mine[i] for i in np.random.default_rng(cfg.seed + epoch).permutation(len(mine))]
            for si in seq:
                name, n_ex = info.shards[si]
                mm = shards_lib.open_shard(d, md, name, n_ex)
                step = n_ex if max_rows is None else max(1, max_rows // len(tok))
                for li in layer_ids:
                    for lo in range(0, n_ex, step):
                        hi = min(n_ex, lo + step)
                        def read(out=None, mm=mm, si=si, li=li, lo=lo, hi=hi, name=name):
                            """Rows of examples [lo, hi) of one shard/layer into out[:n]; returns (n, example_idx, token_idx)
                            (or (rows, example_idx, token_idx) when no output buffer is given)."""
                            ex = np.repeat((np.arange(lo, hi) + ex_base[si]).astype(np.int32), len(tok))
                            tk = np.tile(tok_out, hi - lo)
                            def src():  # (n_ex, n_tok, D) of the mapped file: a strided view when the tokens are a range
                                if tok_contiguous:
                                    return mm[lo:hi, li, tok[0] : tok[0] + len(tok)]
                                return mm[lo:hi, li][:, tok]
                            keep = None
                            if labels is not None:
                                keep = keep_all[ex_base[si] + lo : ex_base[si] + hi].reshape(-1)
                                ex, tk = ex[keep], tk[keep]
                            n = ex.shape[0]
                            if out is None:
                                rows = np.ascontiguousarray(src()).reshape(-1, md.d_model)
                                return (rows if keep is None else rows[keep]), ex, tk
                            if keep is None and tok_contiguous:
                                # one positional read per example straight into the (pinned) staging rows: no page
                                # faults on a shared mapping, the GIL is released for the whole copy, and any number
                                # of reader threads can share the descriptor
                                fd = self._shard_fd(d / name)
                                row_bytes = md.d_model * 4
                                ex_bytes = len(tok) * row_bytes
                                flat = memoryview(out[:n].reshape(-1)).cast("B")
                                for e in range(lo, hi):
                                    off = ((e * len(md.layers) + li) * md.tokens_per_example + tok[0]) * row_bytes
                                    pos, end = (e - lo) * ex_bytes, (e - lo + 1) * ex_bytes
                                    while pos < end:
                                        got = os.preadv(fd, [flat[pos:end]], off)
                                        if got <= 0:
                                            raise IOError(f"short read in shard {name} at byte {off}")
                                        pos += got
                                        off += got
                            elif keep is None:
                                np.copyto(out[:n].reshape(hi - lo, len(tok), md.d_model), src())
                            elif n:
                                out[:n] = np.ascontiguousarray(src()).reshape(-1, md.d_model)[keep]
                            return n, ex, tk

                        yield read

        resident = self._resident_arg
        if resident is None:
            budget = os.environ.get("SAEV_AMD_RESIDENT_GB")
            if budget is not None:
                limit = float(budget) * 1e9
            elif self.device.type == "cuda":
                limit = 0.5 * torch.cuda.get_device_properties(self.device).total_memory
            else:
                limit = float("inf")
            resident = self._n_local * md.d_model * 4 <= limit
        if resident:
            acts, exs, tks = [], [], []
            for read in blocks(0, None):
                a, ex, tk = read()
                if self.device.type == "cuda":  # read once by the H2D copy; a read-only mapped view is fine as a source
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore", UserWarning)
                        acts.append(torch.from_numpy(a).to(self.device))
                else:
                    acts.append(torch.from_numpy(np.array(a, dtype=np.float32)))  # never alias the mapped file
                exs.append(torch.from_numpy(ex))
                tks.append(torch.from_numpy(tk))
            self.pool = torch.cat(acts).contiguous()
            self.example_idx = torch.cat(exs).to(self.device)
            self.token_idx = torch.cat(tks).to(self.device)
            return
        self.pool = None
        chunk = min(4 * self.local_batch, 65536)
        capacity = max(cfg.buffer_size * self.local_batch, (cfg.n_threads + 1) * chunk + self.local_batch)
        self.reservoir = StreamingReservoir(
            lambda: blocks(self._epoch - 1, chunk), d_model=md.d_model, capacity=capacity, chunk_rows=chunk,
            device=self.device, seed=cfg.seed + 7919 * self.rank, min_fill=cfg.min_buffer_fill,
            gather=lambda pool, rows: self.engine.gather_rows(pool, rows) if self.engine is not None else pool[rows],
            timeout_s=cfg.batch_timeout_s, n_threads=cfg.n_threads)

    # -------------------------------------------------------------------------------------
    @property
    def n_local(self) -> int:
        return self.pool.shape[0] if self.pool is not None else self._n_local

    @property
    def n_epoch(self) -> int:
        """Rows this rank delivers per epoch: its share, cut to the smallest share over all ranks (equal step counts
        and batch sizes on every rank are what keeps the per-step collectives of a data-parallel run matched)."""
        return self._n_epoch

    def __len__(self) -> int:
        n = self.n_epoch
        return n // self.local_batch if self.drop_last else math.ceil(n / self.local_batch)

    def _shard_fd(self, path) -> int:
        key = str(path)
        fd = self._fds.get(key)
        if fd is None:
            with self._fd_lock:
                fd = self._fds.get(key)
                if fd is None:
                    fd = self._fds[key] = os.open(key, os.O_RDONLY)
        return fd

    def shutdown(self):
        """Stop the reader threads of the streaming mode (no-op in resident mode); reference shuffled.py shutdown()."""
        if self.reservoir is not None:
            self.reservoir.stop()
        with self._fd_lock:
            for fd in self._fds.values():
                os.close(fd)
            self._fds.clear()

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    def _iter_streaming(self):
        res = self.reservoir
        self._epoch += 1
        res.start_epoch()
        left = self.n_epoch
        try:
            while left > 0:
                got = res.get(min(self.local_batch, left))
                if got is None:
                    return
                act, ex, tk = got
                if act.shape[0] < self.local_batch and self.drop_last:
                    return
                left -= act.shape[0]
                yield {"act": act, "example_idx": ex, "token_idx": tk}
        finally:
            res.stop()

    def _take_perm(self, epoch: int) -> torch.Tensor:
        """This epoch's row order on the device.  Drawn on the host (the order is then independent of the device type: tests replay
        it on CPU) into one of two page-locked staging buffers and uploaded WITHOUT blocking: a pageable ``.to(device)`` is a
        synchronous copy behind everything the stream still holds -- the host then sits out the whole queue at every epoch
        boundary and the device runs dry while it catches up (profiles/r06_train_host_profile.txt: ~10 ms per epoch, the gap
        between train() and the bare engine loop on a slow host)."""
        g = torch.Generator().manual_seed(self.cfg.seed + 1000 * epoch + self.rank)
        if self.device.type != "cuda":
            return torch.randperm(self.n_local, generator=g)[: self.n_epoch].to(self.device)
        stage = self.__dict__.get("_perm_stage")
        if stage is None or stage[0][0].numel() != self.n_local:
            stage = self._perm_stage = [[torch.empty(self.n_local, dtype=torch.int64).pin_memory(), None] for _ in range(2)]
        buf, done = stage[epoch % 2]
        if done is not None:
            done.synchronize()  # (the upload of two epochs ago has long run; an epoch of one or two steps is the exception)
        torch.randperm(self.n_local, generator=g, out=buf)
        perm = buf[: self.n_epoch].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        stage[epoch % 2][1] = ev
        return perm

    def __iter__(self):
        if self.reservoir is not None:
            yield from self._iter_streaming()
            return
        # (host-side permutation, uploaded without blocking: _take_perm)
        epoch = self._epoch
        self._epoch += 1
        perm = self._take_perm(epoch)
        B = self.local_batch
        for lo in range(0, self.n_epoch, B):
            rows = perm[lo : lo + B]
            if rows.shape[0] < B and self.drop_last:
                return
            if self.defer_gather and self.engine is not None:
                # the consumer draws the rows itself, inside its train step (SaeEngine.train_step_gather): "act" is what it gets back
                yield _DeferredBatch(self, rows)
                continue
            if self.engine is not None:
                act = self.engine.gather_rows(self.pool, rows)
            else:
                act = self.pool[rows]
            yield {"act": act, "example_idx": self.example_idx[rows], "token_idx": self.token_idx[rows]}
