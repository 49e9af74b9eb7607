"""Host-side pieces of the inference row against fixtures produced by running the
reference's own inference pass (oracle/gen_golden.py G14): cache hash, index arithmetic, ordered feed, run-dir and
metrics helpers.  No GPU needed."""

import json
import pathlib

import numpy as np
import pytest
import torch

from conftest import load_golden
from saev_amd import disk
from saev_amd.data import Metadata, OrderedConfig, OrderedDataLoader, write_shards
from saev_amd.data import shards as shards_lib
from saev_amd.metrics import Metrics


def write_cache(tmp_path, g):
    labels = g["labels"].numpy() if g["labels"].numel() else None
    return write_shards(tmp_path, g["acts"].numpy(), layers=tuple(g["layers"].tolist()), cls_token=True,
                        max_tokens_per_shard=int(g["max_tokens_per_shard"]), labels=labels)


@pytest.mark.parametrize("tag", ["plain", "labels"])
def test_cache_hash_and_index_math_match_the_reference(tmp_path, tag):
    g = load_golden(f"g14_inference_{tag}")
    d = write_cache(tmp_path, g)
    md = Metadata.load(d)
    assert d.name == md.hash == bytes(g["ref_hash"].numpy()).decode()
    assert md.n_shards == 4 and md.examples_per_shard == 4
    # every acts file is a full shard on disk, the last one zero-padded (the reference maps md.shard_shape)
    sizes = {f.stat().st_size for f in d.glob("acts*.bin")}
    assert sizes == {int(np.prod(md.shard_shape)) * 4}
    for row in g["index_probes"].tolist():
        assert list(shards_lib.locate_content_token(md, row[0], 11)) == row[1:]
    with pytest.raises(IndexError):
        shards_lib.locate_content_token(md, md.n_examples * md.content_tokens_per_example, 11)


@pytest.mark.parametrize("tag", ["plain", "labels"])
def test_ordered_feed_walks_the_global_index(tmp_path, tag):
    g = load_golden(f"g14_inference_{tag}")
    d = write_cache(tmp_path, g)
    acts = g["acts"]
    T = acts.shape[2] - 1
    dl = OrderedDataLoader(OrderedConfig(shards=d, layer=11, batch_size=int(g["batch_size"]) // T * T), device="cpu")
    assert dl.n_samples == acts.shape[0] * T and len(dl) == -(-dl.n_samples // dl.batch_size)
    got = list(dl)
    x = torch.cat([b["act"] for b in got])
    torch.testing.assert_close(x, acts[:, 1, 1:, :].reshape(-1, acts.shape[-1]), rtol=0, atol=0)
    gi = torch.cat([b["example_idx"] * T + b["token_idx"] for b in got])
    assert torch.equal(gi, torch.arange(dl.n_samples)) and got[0]["example_idx"].dtype == torch.int64
    if tag == "labels":
        assert torch.equal(torch.cat([b["token_labels"] for b in got]), g["labels"].reshape(-1).long())
    else:
        assert "token_labels" not in got[0]
    # drop_last drops the ragged tail
    dl2 = OrderedDataLoader(OrderedConfig(shards=d, layer=5, batch_size=20, drop_last=True), device="cpu")
    assert [b["act"].shape[0] for b in dl2] == [20] * (dl.n_samples // 20)
    with pytest.raises(NotImplementedError):
        OrderedDataLoader(OrderedConfig(shards=d, layer="all"), device="cpu")
    with pytest.raises(AssertionError):
        OrderedDataLoader(OrderedConfig(shards=d, layer=3), device="cpu")


def test_run_directory_layout(tmp_path):
    runs_root = tmp_path / "saev" / "runs"
    shards = tmp_path / "saev" / "shards" / "abcd1234"
    shards.mkdir(parents=True)
    runs_root.mkdir(parents=True)
    assert disk.is_runs_root(runs_root) and not disk.is_runs_root(tmp_path)
    assert disk.is_shards_root(shards.parent) and disk.is_shards_dir(shards) and not disk.is_shards_dir(shards.parent)
    run = disk.Run.new("r1", train_shards_dir=shards, val_shards_dir=shards, runs_root=runs_root)
    assert run.run_id == "r1" and run.ckpt == runs_root / "r1" / "checkpoint" / "sae.pt"
    assert run.train_shards == shards.resolve() and run.val_shards == shards.resolve()
    assert run.inference == runs_root / "r1" / "inference"
    (run.run_dir / "checkpoint" / "config.json").write_text(json.dumps({"lr": 0.1}))
    assert disk.Run(run.run_dir).config == {"lr": 0.1}
    with pytest.raises(FileExistsError):
        disk.Run.new("r1", train_shards_dir=shards, val_shards_dir=shards, runs_root=runs_root)
    with pytest.raises(ValueError):
        disk.Run(tmp_path)
    with pytest.raises(FileNotFoundError):
        disk.Run(runs_root / "nope")
    (runs_root / "half").mkdir()
    with pytest.raises(FileNotFoundError):
        disk.Run(runs_root / "half")


def test_metrics_derivations_and_validation():
    g = load_golden("g14_inference_labels")
    want = dict(zip(g["metrics_keys"].tolist(), g["metrics_vals"].tolist()))
    m = Metrics.from_accumulators(sse_recon=want["sse_recon"], sse_baseline=want["sse_baseline"],
                                  n_tokens=int(want["n_tokens"]), d_model=int(want["d_model"]))
    for k, v in want.items():
        assert getattr(m, k) == pytest.approx(v, rel=1e-12), k
    assert Metrics.from_dict(m.to_dict()) == m
    assert list(m.to_dict()) == list(want)  # same key order as the reference's metrics.json
    with pytest.raises(AssertionError):
        Metrics.from_accumulators(sse_recon=1.0, sse_baseline=0.0, n_tokens=3, d_model=2)
    with pytest.raises(AssertionError):
        Metrics.from_accumulators(sse_recon=-1.0, sse_baseline=1.0, n_tokens=3, d_model=2)
    with pytest.raises(AssertionError):
        Metrics.from_dict({**m.to_dict(), "n_tokens": 3.0})
    with pytest.raises(AssertionError):
        Metrics.from_dict({**m.to_dict(), "normalized_mse": 0.5})
    bad = m.to_dict()
    bad.pop("mse_per_dim")
    with pytest.raises(AssertionError):
        Metrics.from_dict(bad)
