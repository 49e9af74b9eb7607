"""Host-side schedules of the train loop.  Two behaviours of the reference are pinned here because they change the
parameters a run ends with (fixture G11 holds the reference's own learning-rate lists and step counts):

* ``WarmupCosine`` -- reference utils/scheduling.py:43-71.  The n-th call (n counted from 1) yields
  ``peak * n / n_warmup`` for n < n_warmup, the half cosine from ``peak`` down to ``final`` for n < n_steps, and
  ``final`` from then on.  The train loop uses the value produced at the end of step g for step g + 1 and runs step 0
  with lr = 0 (train.py:118, 444-451).
* ``BatchLimiter`` -- reference utils/scheduling.py:83-122.  Cycles over a loader until ``n_samples`` rows went by.
  When the loader keeps its ragged last batch (``drop_last`` false) the reference takes one nominal batch off the
  running count after every exhausted epoch; a small dataset therefore produces MORE steps than ``len()`` says, and
  those extra steps run at the schedule's final value.  Kept, because it decides how many optimizer steps happen.

Not in the reference: ``rows_scale``.  Under data parallelism a loader hands each rank ``batch_size / world`` rows of
every global batch; the limiter must count the global rows (``rows_scale = world``) so that every rank performs
``len()``-many steps and the cosine, which is sized with ``len()``, ends where the run ends.
"""

from __future__ import annotations

import math
from collections.abc import Mapping, Sized


class Scheduler:
    """Base of the schedules (reference utils/scheduling.py:9-17): ``step()`` yields the next value."""

    def step(self) -> float:
        raise NotImplementedError(f"{self.__class__.__name__} must implement step().")

    def __repr__(self) -> str:
        raise NotImplementedError(f"{self.__class__.__name__} must implement __repr__().")


class Warmup(Scheduler):
    """Linear ramp from ``init`` to ``final`` over ``n_steps`` calls, ``final`` from then on (reference
    utils/scheduling.py:20-39)."""

    def __init__(self, init: float, final: float, n_steps: int):
        self.init, self.final, self.n_steps = init, final, n_steps
        self.n_calls = 0

    def step(self) -> float:
        self.n_calls += 1
        if self.n_calls < self.n_steps:
            return self.init + (self.final - self.init) * (self.n_calls / self.n_steps)
        return self.final

    def __repr__(self) -> str:
        return f"Warmup(init={self.init}, final={self.final}, n_steps={self.n_steps})"


class WarmupCosine(Scheduler):
    def __init__(self, init: float, n_warmup: int, peak: float, n_steps: int, final: float):
        self.init, self.n_warmup, self.peak, self.n_steps, self.final = init, n_warmup, peak, n_steps, final
        self.n_calls = 0

    def step(self) -> float:
        self.n_calls += 1
        n = self.n_calls
        if n < self.n_warmup:
            return self.init + (self.peak - self.init) * (n / self.n_warmup)
        if n >= self.n_steps:
            return self.final
        t = (n - self.n_warmup) / (self.n_steps - self.n_warmup)
        return self.final + (self.peak - self.final) * ((1 + math.cos(math.pi * t)) / 2)

    def __repr__(self) -> str:
        return f"WarmupCosine({self.init} -> {self.peak} over {self.n_warmup}, -> {self.final} at {self.n_steps})"


def rows_of(batch, nominal: int) -> int:
    """Row count of one batch: the length of a mapping's first value (the loaders yield dicts of equally long
    tensors), else ``len(batch)``; ``nominal`` when neither gives a positive integer."""
    probe = batch
    if isinstance(batch, Mapping):
        probe = next((v for v in batch.values() if v is not None), None)  # (a deferred draw carries "act": None, then its row indices)
    if isinstance(probe, Sized):
        n = len(probe)
        if n > 0:
            return n
    return nominal


class BatchLimiter:
    """Iterates ``loader`` over and over until ``n_samples`` rows have been yielded (see the module docstring for the
    end-of-epoch correction).  Unknown attributes are looked up on the loader (``n_samples``, ``metadata`` ...)."""

    def __init__(self, loader, n_samples: int, *, rows_scale: int = 1):
        self.dataloader = loader
        self.n_samples = n_samples
        self.batch_size = loader.batch_size
        self.drop_last = loader.drop_last
        self.rows_scale = rows_scale
        self.n_seen = 0

    def __len__(self) -> int:
        return -(-self.n_samples // self.batch_size)

    def __getattr__(self, name: str):
        loader = self.__dict__.get("dataloader")
        if loader is None or not hasattr(loader, name):
            raise AttributeError(f"neither BatchLimiter nor the loader it wraps has '{name}'")
        return getattr(loader, name)

    def __iter__(self):
        self.n_seen = 0
        while True:
            for batch in self.dataloader:
                yield batch
                self.n_seen += self.rows_scale * rows_of(batch, self.batch_size // self.rows_scale)
                if self.n_seen >= self.n_samples:
                    return
            if not self.drop_last:
                self.n_seen -= self.batch_size
