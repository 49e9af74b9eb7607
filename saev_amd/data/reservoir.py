"""HBM-resident reservoir fed from the sharded cache by a background reader (the streaming half of the shuffled
feed; replaces reference src/saev/data/buffers.py ReservoirBuffer + shuffled.py _io_worker/_manager_main for caches
that do not fit in device memory).

Reference behaviour kept (buffers.py:90-231, shuffled.py:131-376): shards are visited in a seeded permutation, rows
enter a fixed-capacity reservoir (``buffer_size * batch_size`` rows) as they are read, a batch is B rows drawn
uniformly without replacement from whatever the reservoir holds, every row is delivered exactly once per epoch, and
``min_buffer_fill`` delays the first draw until the reservoir is that full.

MI355X design: the reservoir is one (capacity, d_model) fp32 tensor in HBM plus an int32 (capacity, 2) tensor of
(example_idx, token_idx).  A reader thread cuts whole (examples x tokens) blocks out of the memory-mapped shard,
stages them in pinned host memory and scatters them into free slots with an asynchronous copy on its own HIP stream;
slot bookkeeping (which slots hold unread rows) lives on the host, so a draw costs one index shuffle on the host and
one row-gather kernel on the device -- not B Python-level pops under a lock.  Freed slots return to the reader once
the gather that read them has finished (HIP event).
"""

from __future__ import annotations

import threading
import time

import numpy as np
import torch


class StreamingReservoir:
    """``blocks`` is a callable returning an iterator over one epoch whose items are either
    ``(act (n, D) float32 ndarray, example_idx (n,) int32, token_idx (n,) int32)`` host arrays (n <= chunk_rows) or
    callables ``read(out) -> (n, example_idx, token_idx)`` that write their rows into ``out[:n]`` (a pinned
    ``(chunk_rows, D)`` staging buffer), so the memory-mapped read is the only host copy and happens in whichever of
    the ``n_threads`` reader threads picked the item up."""

    def __init__(self, blocks, *, d_model: int, capacity: int, chunk_rows: int, device: torch.device, seed: int,
                 min_fill: float = 0.0, gather=None, timeout_s: float = 30.0, n_threads: int = 1):
        assert chunk_rows > 0 and capacity >= chunk_rows * max(1, n_threads), "every reader must be able to place a block"
        self.blocks = blocks
        self.D, self.capacity, self.chunk_rows = d_model, capacity, chunk_rows
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.min_fill = min_fill
        self.gather = gather  # HIP row gather (engine.gather_rows); torch indexing when None (CPU tests)
        self.timeout_s = timeout_s
        self.rng = np.random.default_rng(seed)
        self.rows = torch.empty(capacity, d_model, dtype=torch.float32, device=self.device)
        self.meta = torch.empty(capacity, 2, dtype=torch.int32, device=self.device)
        self._cv = threading.Condition()
        self._filled = np.empty(capacity, dtype=np.int64)
        self._n_filled = 0
        self._free = np.arange(capacity, dtype=np.int64)[::-1].copy()
        self._n_free = capacity
        self._pending: list[tuple[object, np.ndarray]] = []  # (event, slots) read by a gather still in flight
        self._done = True
        self._err: BaseException | None = None
        self.n_threads = max(1, n_threads)
        self._threads: list[threading.Thread] = []
        self._n_running = 0
        self._src = None
        self._src_lock = threading.Lock()
        self._stop = False
        self._staging: dict[int, tuple[list[torch.Tensor], list[torch.Tensor]]] = {}
        self._slot_ring: list[list] = []
        self._slot_i = 0
        # seconds spent per phase, summed over reader threads / consumer calls (tools/bench_feed.py prints them)
        self.phase_s = {"alloc": 0.0, "read": 0.0, "wait_free": 0.0, "wait_copy": 0.0, "enqueue": 0.0, "get_wait": 0.0,
                        "get_draw": 0.0, "get_gather": 0.0}

    # ---- reader side -----------------------------------------------------------------------------------------------
    def start_epoch(self):
        self.stop()
        with self._cv:
            self._n_filled = 0
            self._free = np.arange(self.capacity, dtype=np.int64)[::-1].copy()
            self._n_free = self.capacity
            self._pending.clear()
            self._done, self._err, self._stop = False, None, False
            self._n_running = self.n_threads
        self._src = iter(self.blocks())
        self._threads = [threading.Thread(target=self._reader, args=(i,), name=f"saev-reservoir-reader-{i}", daemon=True)
                         for i in range(self.n_threads)]
        for t in self._threads:
            t.start()

    def stop(self):
        if self._threads:
            with self._cv:
                self._stop = True
                self._cv.notify_all()
            for t in self._threads:
                t.join()
            self._threads = []

    def _next_item(self):
        with self._src_lock:  # the generator itself only does index arithmetic; reads happen outside the lock
            return next(self._src, None)

    def _reclaim_locked(self):
        keep = []
        for ev, slots in self._pending:
            if ev is None or ev.query():
                n = slots.shape[0]
                self._free[self._n_free : self._n_free + n] = slots
                self._n_free += n
            else:
                keep.append((ev, slots))
        self._pending = keep

    def _take_free(self, n: int) -> np.ndarray | None:
        with self._cv:
            while True:
                self._reclaim_locked()
                if self._stop:
                    return None
                if self._n_free >= n:
                    self._n_free -= n
                    return self._free[self._n_free : self._n_free + n].copy()
                self._cv.wait(timeout=0.002 if self._pending else 0.05)

    def _reader(self, tid: int = 0):
        try:
            ph = self.phase_s
            t0 = time.perf_counter()
            copy_stream = torch.cuda.Stream(self.device) if self.on_gpu else None
            # pinned staging is expensive to allocate (hundreds of ms per reader): kept across epochs
            if tid not in self._staging:
                self._staging[tid] = (
                    [torch.empty(self.chunk_rows, self.D, dtype=torch.float32, pin_memory=self.on_gpu) for _ in range(2)],
                    [torch.empty(self.chunk_rows, 2, dtype=torch.int32, pin_memory=self.on_gpu) for _ in range(2)])
            stage, stage_meta = self._staging[tid]
            inflight: list[tuple[object, np.ndarray] | None] = [None, None]
            ph["alloc"] += time.perf_counter() - t0

            def publish(j):
                if inflight[j] is None:
                    return
                ev, slots = inflight[j]
                if ev is not None:
                    t1 = time.perf_counter()
                    ev.synchronize()  # rows are published only once they are in HBM
                    ph["wait_copy"] += time.perf_counter() - t1
                inflight[j] = None
                with self._cv:
                    n = slots.shape[0]
                    self._filled[self._n_filled : self._n_filled + n] = slots
                    self._n_filled += n
                    self._cv.notify_all()

            i = 0
            while True:
                item = self._next_item()
                if item is None:
                    break
                j = i & 1
                i += 1
                publish(j)  # the copy that last used this staging buffer has landed
                # host read (page cache / disk) straight into pinned memory; overlaps the other buffer's copy
                t1 = time.perf_counter()
                if callable(item):
                    n, ex, tk = item(stage[j].numpy())
                else:
                    act, ex, tk = item
                    n = act.shape[0]
                    assert n <= self.chunk_rows
                    stage[j].numpy()[:n] = act
                ph["read"] += time.perf_counter() - t1
                if n == 0:
                    continue
                sm = stage_meta[j].numpy()
                sm[:n, 0], sm[:n, 1] = ex, tk
                publish(1 - j)  # never wait for free slots while holding unpublished rows
                t1 = time.perf_counter()
                slots = self._take_free(n)
                ph["wait_free"] += time.perf_counter() - t1
                if slots is None:
                    return
                t1 = time.perf_counter()
                slots_t = torch.from_numpy(slots)
                ev = None
                if self.on_gpu:
                    with torch.cuda.stream(copy_stream):
                        sd = slots_t.to(self.device, non_blocking=True)
                        self.rows.index_copy_(0, sd, stage[j][:n].to(self.device, non_blocking=True))
                        self.meta.index_copy_(0, sd, stage_meta[j][:n].to(self.device, non_blocking=True))
                        ev = torch.cuda.Event()
                        ev.record()
                else:
                    self.rows.index_copy_(0, slots_t, stage[j][:n])
                    self.meta.index_copy_(0, slots_t, stage_meta[j][:n])
                inflight[j] = (ev, slots)
                ph["enqueue"] += time.perf_counter() - t1
            publish(0)
            publish(1)
        except BaseException as e:  # surfaced to the consumer
            self._err = e
        finally:
            with self._cv:
                self._n_running -= 1
                if self._n_running == 0 or self._err is not None:
                    self._done = True
                self._cv.notify_all()

    # ---- consumer side ---------------------------------------------------------------------------------------------
    def fill(self) -> float:
        return self._n_filled / self.capacity

    def _draw(self, n: int, b: int) -> np.ndarray:
        """b distinct positions of range(n), uniform over all b-subsets, in O(b log b) (not O(n)) when b << n."""
        if b >= n:
            return np.arange(n)
        if n < 4 * b:
            return self.rng.permutation(n)[:b]
        got = np.unique(self.rng.integers(0, n, size=b + b // 8 + 16))
        while got.shape[0] < b:
            got = np.unique(np.concatenate([got, self.rng.integers(0, n, size=b)]))
        if got.shape[0] > b:  # drop a uniformly chosen surplus
            got = np.delete(got, self.rng.choice(got.shape[0], got.shape[0] - b, replace=False))
        return got

    def get(self, batch_size: int):
        """Up to ``batch_size`` rows drawn uniformly without replacement from the reservoir; fewer only when the epoch
        is running out; ``None`` when it is exhausted."""
        deadline = time.monotonic() + self.timeout_s
        ph = self.phase_s
        t0 = time.perf_counter()
        with self._cv:
            while True:
                if self._err is not None:
                    raise RuntimeError("reservoir reader failed") from self._err
                # the fill threshold leaves room for every reader's next block, or nobody could make progress
                room = self.capacity - self.chunk_rows * self.n_threads
                assert room >= batch_size, "capacity must cover one batch plus one block per reader thread"
                want = min(max(batch_size, int(self.min_fill * self.capacity)), max(room, 1)) if not self._done else 1
                if self._n_filled >= want or (self._done and self._n_filled > 0):
                    break
                if self._done:
                    return None
                if not self._cv.wait(timeout=0.5) and time.monotonic() > deadline:
                    raise TimeoutError(f"no batch within {self.timeout_s}s (reservoir fill {self.fill():.3f})")
            t1 = time.perf_counter()
            ph["get_wait"] += t1 - t0
            n = self._n_filled
            b = min(batch_size, n)
            pick = self._draw(n, b)
            slots = self._filled[pick].copy()
            # close the holes with the tail entries that were not drawn themselves
            in_tail = pick >= n - b
            tail_kept = np.ones(b, dtype=bool)
            tail_kept[pick[in_tail] - (n - b)] = False
            self._filled[pick[~in_tail]] = self._filled[n - b : n][tail_kept]
            self._n_filled = n - b
        t2 = time.perf_counter()
        ph["get_draw"] += t2 - t1
        if self.on_gpu:
            # the slot list goes up through a small ring of pinned buffers with an asynchronous copy: a pageable source
            # would make the copy synchronous, i.e. stall the host until the stream has drained, once per batch
            ring = self._slot_ring
            if not ring:
                ring.extend([torch.empty(batch_size, dtype=torch.int64, pin_memory=True), None] for _ in range(8))
            ent = ring[self._slot_i % len(ring)]
            self._slot_i += 1
            if ent[0].shape[0] < b:
                ent[0], ent[1] = torch.empty(b, dtype=torch.int64, pin_memory=True), None
            if ent[1] is not None:
                ent[1].synchronize()
            ent[0][:b].copy_(torch.from_numpy(slots))
            slots_t = ent[0][:b].to(self.device, non_blocking=True)
            ent[1] = torch.cuda.Event()
            ent[1].record()
        else:
            slots_t = torch.from_numpy(slots)
        act = self.gather(self.rows, slots_t) if self.gather is not None else self.rows[slots_t]
        meta = self.meta[slots_t]
        ev = None
        if self.on_gpu:
            ev = torch.cuda.Event()
            ev.record()
        with self._cv:
            self._pending.append((ev, slots))
            self._cv.notify_all()
        ph["get_gather"] += time.perf_counter() - t2
        return act, meta[:, 0].contiguous(), meta[:, 1].contiguous()
