"""Host-side schedules of the train loop; must match the reference step for step
(reference: src/saev/utils/scheduling.py).

* ``WarmupCosine`` (scheduling.py:43-71): call n (1-based) returns the linear ramp ``peak*n/n_warmup``
  while ``n < n_warmup``, the half-cosine from ``peak`` to ``final`` while ``n < n_steps``, and
  ``final`` afterwards.  The train loop applies the value returned at the end of step g to step g+1,
  and step 0 runs with lr = 0 (train.py:118,444-451).
* ``BatchLimiter`` (scheduling.py:83-122): re-iterates a loader until ``n_samples`` rows were seen.
  Quirk kept on purpose: when the loader has ``drop_last == False`` the seen-counter is decremented by
  one nominal batch after every exhausted epoch, so a small dataset yields *more* steps than
  ``len()`` reports (the extra steps run with lr = final).
"""

from __future__ import annotations

import collections.abc
import math
from typing import Any, Iterator, Protocol, runtime_checkable


class Scheduler:
    def step(self) -> float:
        raise NotImplementedError(f"{type(self).__name__} must implement step().")


class Warmup(Scheduler):
    """Linear ramp from ``init`` to ``final`` over ``n_steps`` calls (scheduling.py:19-39)."""

    def __init__(self, init: float, final: float, n_steps: int):
        self.init, self.final, self.n_steps = init, final, n_steps
        self._step = 0

    def step(self) -> float:
        self._step += 1
        if self._step < self.n_steps:
            return self.init + (self.final - self.init) * (self._step / self.n_steps)
        return self.final

    def __repr__(self) -> str:
        return f"Warmup(init={self.init}, final={self.final}, n_steps={self.n_steps})"


class WarmupCosine(Scheduler):
    def __init__(self, init: float, n_warmup: int, peak: float, n_steps: int, final: float):
        self.init, self.n_warmup, self.peak, self.n_steps, self.final = init, n_warmup, peak, n_steps, final
        self._step = 0

    def step(self) -> float:
        self._step += 1
        n = self._step
        if n < self.n_warmup:
            return self.init + (self.peak - self.init) * (n / self.n_warmup)
        if n < self.n_steps:
            t = (n - self.n_warmup) / (self.n_steps - self.n_warmup)
            return self.final + (self.peak - self.final) * ((1 + math.cos(math.pi * t)) / 2)
        return self.final

    def __repr__(self) -> str:
        return (f"WarmupCosine(init={self.init}, peak={self.peak}, final={self.final}, "
                f"n_warmup={self.n_warmup}, n_steps={self.n_steps})")


@runtime_checkable
class DataLoaderLike(Protocol):
    drop_last: bool
    batch_size: int

    def __iter__(self) -> Iterator[Any]: ...


def _rows_in(batch: Any, fallback: int) -> int:
    """Rows of a batch without assuming its schema: first value of a mapping, else len()."""
    try:
        if isinstance(batch, collections.abc.Mapping):
            if not batch:
                return fallback
            n = len(next(iter(batch.values())))
        else:
            n = len(batch)
        return n if isinstance(n, int) and n > 0 else fallback
    except Exception:
        return fallback


class BatchLimiter:
    def __init__(self, dataloader: DataLoaderLike, n_samples: int):
        self.dataloader = dataloader
        self.n_samples = n_samples
        self.batch_size = dataloader.batch_size
        self.drop_last = dataloader.drop_last

    def __len__(self) -> int:
        return math.ceil(self.n_samples / self.batch_size)

    def __getattr__(self, name: str) -> Any:
        # only reached when normal lookup fails: delegate to the wrapped loader
        try:
            return getattr(self.__dict__["dataloader"], name)
        except (KeyError, AttributeError):
            raise AttributeError(f"'{type(self).__name__}' object and its wrapped dataloader have no attribute '{name}'")

    def __iter__(self):
        self.n_seen = 0
        while True:
            for batch in self.dataloader:
                yield batch
                self.n_seen += _rows_in(batch, self.batch_size)
                if self.n_seen >= self.n_samples:
                    return
            if not self.dataloader.drop_last:
                self.n_seen -= self.batch_size
