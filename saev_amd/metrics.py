"""Reconstruction metrics of one evaluation corpus (reference src/saev/metrics.py:14-160).

Primary totals: ``sse_recon`` (sum of squared reconstruction errors) and ``sse_baseline`` (sum of squared errors of
the mean predictor); the rest is derived: normalized_mse = sse_recon / sse_baseline, per-dimension and per-token
means over ``n_elements = n_tokens * d_model``."""

from __future__ import annotations

import dataclasses
import math
from collections import abc


def close(a: float, b: float) -> bool:
    return math.isclose(a, b, rel_tol=1e-9, abs_tol=1e-12)


@dataclasses.dataclass(frozen=True)
class Metrics:
    mse_per_dim: float
    mse_per_token: float
    normalized_mse: float
    baseline_mse_per_dim: float
    baseline_mse_per_token: float
    sse_recon: float
    sse_baseline: float
    n_tokens: int
    d_model: int
    n_elements: int

    def __post_init__(self):
        for name in ("n_tokens", "d_model", "n_elements"):
            v = getattr(self, name)
            assert type(v) is int, f"{name} must be an int, got {type(v)}."
        assert self.n_tokens > 0, f"n_tokens must be positive, got {self.n_tokens}."
        assert self.d_model > 0, f"d_model must be positive, got {self.d_model}."
        assert self.n_elements == self.n_tokens * self.d_model, (
            f"n_elements={self.n_elements} != n_tokens*d_model={self.n_tokens * self.d_model}.")
        assert self.sse_recon >= 0.0, f"sse_recon must be >= 0, got {self.sse_recon}."
        assert self.sse_baseline > 0.0, f"sse_baseline must be > 0, got {self.sse_baseline}."
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            assert math.isfinite(v), f"{f.name} must be finite, got {v}."
        derived = {
            "mse_per_dim": self.sse_recon / self.n_elements, "mse_per_token": self.sse_recon / self.n_tokens,
            "baseline_mse_per_dim": self.sse_baseline / self.n_elements,
            "baseline_mse_per_token": self.sse_baseline / self.n_tokens,
            "normalized_mse": self.sse_recon / self.sse_baseline,
        }
        for name, want in derived.items():
            assert close(getattr(self, name), want), f"{name}={getattr(self, name)} is inconsistent with {want}."

    @classmethod
    def from_accumulators(cls, *, sse_recon: float, sse_baseline: float, n_tokens: int, d_model: int) -> "Metrics":
        assert n_tokens > 0, f"n_tokens must be positive, got {n_tokens}."
        assert d_model > 0, f"d_model must be positive, got {d_model}."
        assert sse_recon >= 0.0, f"sse_recon must be >= 0, got {sse_recon}."
        assert sse_baseline > 0.0, f"sse_baseline must be > 0, got {sse_baseline}."
        n = n_tokens * d_model
        return cls(mse_per_dim=sse_recon / n, mse_per_token=sse_recon / n_tokens, normalized_mse=sse_recon / sse_baseline,
                   baseline_mse_per_dim=sse_baseline / n, baseline_mse_per_token=sse_baseline / n_tokens,
                   sse_recon=sse_recon, sse_baseline=sse_baseline, n_tokens=n_tokens, d_model=d_model, n_elements=n)

    @classmethod
    def from_dict(cls, dct: abc.Mapping[str, object]) -> "Metrics":
        vals = {}
        for f in dataclasses.fields(cls):
            assert f.name in dct, f"missing key {f.name!r}"
            v = dct[f.name]
            if f.type in (int, "int"):
                assert type(v) is int, f"{f.name} must be an int, got {type(v)}."
                vals[f.name] = v
            else:
                assert isinstance(v, (int, float)) and not isinstance(v, bool), f"{f.name} must be a number, got {type(v)}."
                vals[f.name] = float(v)
        return cls(**vals)

    def to_dict(self) -> dict[str, float | int]:
        return dataclasses.asdict(self)
