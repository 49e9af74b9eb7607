"""Reader (and a minimal writer) for saev's sharded activation cache, protocol 2.1
(reference: docs/src/developers/protocol.md; src/saev/data/shards.py:42-185).

On disk:  ``<root>/saev/shards/<hash>/{metadata.json, shards.json, acts000000.bin, ...}`` where each
``acts*.bin`` is a raw C-contiguous float32 array ``(examples_in_shard, n_layers, tokens_per_example,
d_model)``, ``tokens_per_example = content_tokens_per_example + (1 if cls_token else 0)`` with the CLS
token first, and ``examples_per_shard = max_tokens_per_shard // (tokens_per_example * n_layers)``.
``<hash>`` is the first 8 hex digits of sha256 over the canonical metadata JSON.

Only what the train step's feed needs is here: parse the two JSON files, memory-map shards, and write
small caches for tests/synthetic runs.  The ViT extraction side of the reference (forward hooks,
image datasets) is out of scope.
"""

from __future__ import annotations

import dataclasses
import hashlib
import json
import pathlib

import numpy as np


@dataclasses.dataclass(frozen=True)
class Metadata:
    family: str
    ckpt: str
    layers: tuple[int, ...]
    content_tokens_per_example: int
    cls_token: bool
    d_model: int
    n_examples: int
    max_tokens_per_shard: int
    data: str = ""
    dataset: str = ""
    pixel_agg: str = "majority"
    dtype: str = "float32"
    protocol: str = "2.1"

    @property
    def tokens_per_example(self) -> int:
        return self.content_tokens_per_example + (1 if self.cls_token else 0)

    @property
    def examples_per_shard(self) -> int:
        return self.max_tokens_per_shard // (self.tokens_per_example * len(self.layers))

    @property
    def shard_shape(self) -> tuple[int, int, int, int]:
        return (self.examples_per_shard, len(self.layers), self.tokens_per_example, self.d_model)

    @property
    def n_shards(self) -> int:
        return -(-self.n_examples // self.examples_per_shard)

    def __post_init__(self):
        assert self.examples_per_shard >= 1, "At least one example per shard must fit; increase max_tokens_per_shard."
        assert self.dtype == "float32"

    def to_json(self) -> dict:
        d = dataclasses.asdict(self)
        d["layers"] = list(self.layers)
        d["dataset"] = str(pathlib.Path(self.dataset))  # the reference holds a pathlib.Path here ("" reads back as ".")
        return d

    @property
    def hash(self) -> str:
        # protocol.md: sha256(json.dumps(metadata, sort_keys=True, separators=(',', ':'))); the reference produces the
        # same bytes with orjson (compact, sorted keys, raw UTF-8), shards.py:126-135
        blob = json.dumps(self.to_json(), sort_keys=True, separators=(",", ":"), ensure_ascii=False).encode("utf-8")
        return hashlib.sha256(blob).hexdigest()[:8]

    @classmethod
    def load(cls, shards_dir: pathlib.Path) -> "Metadata":
        with open(pathlib.Path(shards_dir) / "metadata.json") as fd:
            d = json.load(fd)
        d["layers"] = tuple(d.pop("layers"))
        known = {f.name for f in dataclasses.fields(cls)}
        return cls(**{k: v for k, v in d.items() if k in known})


@dataclasses.dataclass(frozen=True)
class ShardInfo:
    """shards.json: a list of {name, n_examples} (protocol.md section 2.2)."""

    shards: tuple[tuple[str, int], ...]

    @classmethod
    def load(cls, shards_dir: pathlib.Path) -> "ShardInfo":
        with open(pathlib.Path(shards_dir) / "shards.json") as fd:
            arr = json.load(fd)
        return cls(tuple((s["name"], int(s["n_examples"])) for s in arr))

    def __iter__(self):
        return iter(self.shards)

    def __len__(self):
        return len(self.shards)

    def validate(self, shards_dir: pathlib.Path | str, md: "Metadata | None" = None) -> None:
        """Check every shard file of shards.json BEFORE a loader starts reading (reference data/shards.py:638-694, called from
        its loaders' constructors): files that are missing, empty, unreadable or not regular files are collected and reported
        together in one FileNotFoundError, instead of surfacing one at a time from a reader thread in the middle of an
        epoch.  With ``md`` a file shorter than its (n_examples, layers, tokens, d_model) float32 block counts as truncated."""
        import stat

        root = pathlib.Path(shards_dir)
        problems: dict[str, list[str]] = {"Missing files": [], "Empty files": [], "Unreadable files": [], "Not regular files": [],
                                          "Truncated files": []}
        for name, n_examples in self.shards:
            fpath = root / name
            shown = str(fpath.resolve())
            try:
                st = fpath.stat()
            except FileNotFoundError:
                problems["Missing files"].append(shown)
                continue
            except OSError:  # (PermissionError included)
                problems["Unreadable files"].append(shown)
                continue
            if not stat.S_ISREG(st.st_mode):
                problems["Not regular files"].append(shown)
            elif st.st_size == 0:
                problems["Empty files"].append(shown)
            elif md is not None and st.st_size < 4 * n_examples * len(md.layers) * md.tokens_per_example * md.d_model:
                problems["Truncated files"].append(f"{shown} ({st.st_size} bytes, shards.json promises {n_examples} examples)")
        if not any(problems.values()):
            return
        lines = [f"Shard validation failed in '{root.resolve()}':"]
        for title, items in problems.items():
            if items:
                lines += ["", f"{title} ({len(items)}):", *(f"  - {it}" for it in items)]
        raise FileNotFoundError("\n".join(lines))


def locate_content_token(md: Metadata, g: int, layer: int) -> tuple[int, int, int, int, int, int]:
    """Global content-token index -> (example, content token, shard, example in shard, layer slot, token slot in the
    shard's token axis), the arithmetic of the reference's IndexMap.from_global for ("content", fixed layer)
    (shards.py:1042-1067): tokens of an example are consecutive, examples fill shards in order, and the token axis of
    a shard starts with the CLS token when there is one."""
    T, eps = md.content_tokens_per_example, md.examples_per_shard
    if not 0 <= g < md.n_examples * T:
        raise IndexError(f"Index {g} out of range for dataset of length {md.n_examples * T}")
    example, tok = divmod(g, T)
    return example, tok, example // eps, example % eps, md.layers.index(layer), tok + (1 if md.cls_token else 0)


def open_shard(shards_dir: pathlib.Path, md: Metadata, name: str, n_examples: int) -> np.memmap:
    shape = (n_examples, len(md.layers), md.tokens_per_example, md.d_model)
    return np.memmap(pathlib.Path(shards_dir) / name, mode="r", dtype=np.float32, shape=shape)


def write_shards(root: pathlib.Path, acts: np.ndarray, *, layers: tuple[int, ...] = (0,), cls_token: bool = False,
                 max_tokens_per_shard: int | None = None, family: str = "fake-clip", ckpt: str = "synthetic",
                 labels: np.ndarray | None = None) -> pathlib.Path:
    """Write ``acts`` (n_examples, n_layers, tokens_per_example, d_model) float32 as a protocol-2.1
    cache under ``root/saev/shards/<hash>`` and return that directory (tests, synthetic runs).
    ``labels`` (n_examples, content_tokens_per_example) uint8 becomes ``labels.bin`` (protocol.md section 2.3)."""
    acts = np.ascontiguousarray(acts, dtype=np.float32)
    n_ex, n_layers, tokens, d = acts.shape
    assert n_layers == len(layers)
    md = Metadata(
        family=family, ckpt=ckpt, layers=tuple(layers), content_tokens_per_example=tokens - (1 if cls_token else 0),
        cls_token=cls_token, d_model=d, n_examples=n_ex,
        max_tokens_per_shard=max_tokens_per_shard or n_ex * tokens * n_layers,
    )
    out = pathlib.Path(root) / "saev" / "shards" / md.hash
    out.mkdir(parents=True, exist_ok=True)
    infos = []
    eps = md.examples_per_shard
    for i, lo in enumerate(range(0, n_ex, eps)):
        name = f"acts{i:06d}.bin"
        chunk = acts[lo : lo + eps]
        with open(out / name, "wb") as fd:
            chunk.tofile(fd)
            # every acts file has the full shard_shape on disk (the reference creates them as zero-filled
            # np.memmap(mode="w+", shape=md.shard_shape), shards.py:522-524, and its readers map that shape);
            # shards.json records how many examples are real
            pad = (eps - chunk.shape[0]) * n_layers * tokens * d * 4
            if pad:
                fd.write(b"\0" * pad)
        infos.append({"name": name, "n_examples": int(chunk.shape[0])})
    with open(out / "metadata.json", "w") as fd:
        json.dump(md.to_json(), fd, indent=2)
    with open(out / "shards.json", "w") as fd:
        json.dump(infos, fd, indent=2)
    if labels is not None:
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        assert labels.shape == (n_ex, md.content_tokens_per_example)
        labels.tofile(out / "labels.bin")
    return out
