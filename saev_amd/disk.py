"""Run-directory layout helpers (reference src/saev/disk.py:12-160; docs/src/developers/disk-layout.md).

    <runs_root = .../saev/runs>/<run_id>/{checkpoint/{sae.pt,config.json}, links/{train,val}-shards, inference/}
    <shards_root = .../saev/shards>/<metadata hash>/{metadata.json, shards.json, acts*.bin, labels.bin?}
"""

from __future__ import annotations

import json
import pathlib


def _ends_with(path: pathlib.Path, *tail: str) -> bool:
    return path.parts[-len(tail):] == tail


def is_runs_root(path: pathlib.Path) -> bool:
    """An existing directory whose last two components are ``saev/runs`` (disk.py:12-24)."""
    return path.is_dir() and _ends_with(path, "saev", "runs")


def is_shards_root(path: pathlib.Path) -> bool:
    """An existing directory whose last two components are ``saev/shards`` (disk.py:28-40)."""
    return path.is_dir() and _ends_with(path, "saev", "shards")


def is_shards_dir(path: pathlib.Path) -> bool:
    """An existing ``.../saev/shards/<hash>`` directory (disk.py:44-63; the file check is not enforced there either)."""
    return path.is_dir() and len(path.parts) >= 3 and path.parts[-3:-1] == ("saev", "shards")


class Run:
    """One training run on disk (disk.py:67-160).  ``Run(dir)`` validates an existing run; ``Run.new`` creates one."""

    SUBDIRS = ("checkpoint", "links", "inference")

    def __init__(self, run_dir: pathlib.Path):
        self.run_dir = pathlib.Path(run_dir)
        if len(self.run_dir.parts) < 3 or self.run_dir.parts[-3:-1] != ("saev", "runs"):
            raise ValueError(f"'{self.run_dir}' is not of the form <...>/saev/runs/<run id>")
        missing = [p for p in (self.run_dir, *(self.run_dir / sub for sub in self.SUBDIRS)) if not p.exists()]
        if missing:
            raise FileNotFoundError(f"not a complete run on disk, missing: {', '.join(map(str, missing))} (Run.new(...) lays one out)")

    @classmethod
    def new(cls, run_id: str, *, train_shards_dir: pathlib.Path, val_shards_dir: pathlib.Path,
            runs_root: pathlib.Path) -> "Run":
        run_dir = pathlib.Path(runs_root) / run_id
        run_dir.mkdir(parents=True)
        for sub in cls.SUBDIRS:
            (run_dir / sub).mkdir()
        (run_dir / "links" / "train-shards").symlink_to(train_shards_dir)
        (run_dir / "links" / "val-shards").symlink_to(val_shards_dir)
        return cls(run_dir)

    @property
    def run_id(self) -> str:
        return self.run_dir.name

    @property
    def config(self) -> dict[str, object]:
        with open(self.run_dir / "checkpoint" / "config.json") as fd:
            return json.load(fd)

    @property
    def ckpt(self) -> pathlib.Path:
        return self.run_dir / "checkpoint" / "sae.pt"

    @property
    def train_shards(self) -> pathlib.Path:
        return (self.run_dir / "links" / "train-shards").resolve()

    @property
    def val_shards(self) -> pathlib.Path:
        return (self.run_dir / "links" / "val-shards").resolve()

    @property
    def inference(self) -> pathlib.Path:
        return self.run_dir / "inference"
