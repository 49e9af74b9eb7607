"""One ordered pass over a cache that dumps a trained SAE's inference artifacts
(reference src/saev/framework/inference.py:1-285).

Writes under ``<run>/inference/<metadata hash>/``:

    config.json        the inference config
    metrics.json       saev_amd.metrics.Metrics (fp64 accumulators: SSE of the SAE, SSE of the mean predictor)
    token_acts.npz     scipy CSR (n_tokens, d_sae) of the sparse codes            (save=True only)
    mean_values.pt     (d_sae,) sum of activations / number of tokens with f > 0   (save=True only)
    sparsity.pt        (d_sae,) fraction of tokens with f > 0                      (save=True only)
    distributions.pt   (n_tokens, n_dists), rows written at ``example_idx``        (save=True only)

The reference materialises the dense (B, d_sae) ``f_x`` per batch, copies it to the host and lets scipy compress it
(inference.py:189-246).  Here the codes never leave their sparse form: the HIP encoder returns (idx, val) with
ascending latent indices per row, which *is* a CSR block; per-latent sums are index-adds over the B*k codes.
"""

from __future__ import annotations

import collections.abc
import dataclasses
import json
import logging
import os
import pathlib

import numpy as np
import scipy.sparse
import torch

from .. import disk, nn
from ..data import Metadata, OrderedConfig, OrderedDataLoader
from ..metrics import Metrics

logger = logging.getLogger("inference.py")


@dataclasses.dataclass(frozen=True)
class Config:
    """Field names and defaults of inference.py:42-75."""

    run: pathlib.Path = pathlib.Path("./runs/abcdefg")
    data: OrderedConfig = OrderedConfig()
    n_dists: int = 25
    ignore_labels: list[int] = dataclasses.field(default_factory=list)
    force_recompute: bool = False
    save: bool = True
    device: str = "cuda"
    slurm_acct: str = ""
    slurm_partition: str = ""
    n_hours: float = 4.0
    mem_gb: int = 80
    log_to: str = os.path.join(".", "logs")


@dataclasses.dataclass(frozen=True)
class Filepaths:
    mean_values: pathlib.Path
    sparsity: pathlib.Path
    distributions: pathlib.Path
    token_acts: pathlib.Path
    metrics: pathlib.Path

    @classmethod
    def from_run(cls, run: disk.Run, md: Metadata) -> "Filepaths":
        root = run.inference / md.hash
        root.mkdir(exist_ok=True, parents=True)
        return cls(mean_values=root / "mean_values.pt", sparsity=root / "sparsity.pt",
                   distributions=root / "distributions.pt", token_acts=root / "token_acts.npz",
                   metrics=root / "metrics.json")

    def __iter__(self) -> collections.abc.Iterator[pathlib.Path]:
        yield from (self.mean_values, self.sparsity, self.distributions, self.token_acts, self.metrics)


def _shards_dir(cfg: Config) -> pathlib.Path:
    return pathlib.Path(os.path.expandvars(str(cfg.data.shards)))


def need_compute(cfg: Config) -> tuple[bool, str, Filepaths]:
    """(inference.py:108-134) recompute when forced or when a required output is missing."""
    run = disk.Run(cfg.run)
    fpaths = Filepaths.from_run(run, Metadata.load(_shards_dir(cfg)))
    required, mode = (list(fpaths), "full artifacts") if cfg.save else ([fpaths.metrics], "metrics only")
    missing = [f for f in required if not f.exists()]
    if cfg.force_recompute:
        return True, f"Force recompute flag set; computing {mode}.", fpaths
    if not missing:
        return False, f"Found all required files ({mode}).", fpaths
    return True, f"Missing files {', '.join(str(f) for f in missing)}; computing {mode}.", fpaths


def _jsonable(o):
    if dataclasses.is_dataclass(o) and not isinstance(o, type):
        return {f.name: _jsonable(getattr(o, f.name)) for f in dataclasses.fields(o)}
    if isinstance(o, pathlib.Path):
        return str(o)
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, dict):
        return {k: _jsonable(v) for k, v in o.items()}
    return o


@torch.inference_mode()
def worker_fn(cfg: Config):
    run = disk.Run(cfg.run)
    md = Metadata.load(_shards_dir(cfg))
    root = run.inference / md.hash
    do, reason, fpaths = need_compute(cfg)
    logger.info(reason)
    if not do:
        return
    with open(root / "config.json", "w") as fd:
        json.dump(_jsonable(cfg), fd)
    assert cfg.data.tokens == "content"
    device = torch.device(cfg.device)
    if device.type != "cuda":
        raise RuntimeError("saev_amd inference runs on a HIP device only (there is no CPU path)")
    sae = nn.load(run.ckpt, device=device)
    S, D = sae.cfg.d_sae, sae.cfg.d_model
    T = md.content_tokens_per_example
    batch_size = cfg.data.batch_size // T * T  # whole examples per batch (inference.py:158-165)
    loader = OrderedDataLoader(dataclasses.replace(cfg.data, batch_size=batch_size), device=device)
    eng = sae._eng(batch_size)

    if cfg.save:
        value_sum = torch.zeros(S, device=device)
        n_pos = torch.zeros(S, device=device)
        distributions = np.zeros((loader.n_samples, cfg.n_dists), dtype=np.float32)
        csr_data: list[np.ndarray] = []
        csr_cols: list[np.ndarray] = []
        csr_counts: list[np.ndarray] = []
    ignore = torch.tensor(cfg.ignore_labels, dtype=torch.int64)
    sse = torch.zeros((), dtype=torch.float64, device=device)
    sum_sq = torch.zeros((), dtype=torch.float64, device=device)
    sum_vec = torch.zeros(D, dtype=torch.float64, device=device)
    n_tokens = 0
    prev_i = -1
    logger.info("Loaded SAE and data.")

    for batch in loader:
        x = batch["act"]
        b = x.shape[0]
        eng.step_forward(x, training=False)
        idx, val, x_hat = eng.last_codes(b)
        keep_host = torch.ones(b, dtype=torch.bool)
        if "token_labels" in batch:  # segmentation caches: drop tokens whose label is ignored
            keep_host = torch.isin(batch["token_labels"], ignore, invert=True)
        n_keep = int(keep_host.sum())
        n_tokens += n_keep
        keep = keep_host.to(device)
        if n_keep > 0:
            if n_keep == b:
                st = eng.read_stats()  # fp64 sums of this batch from the step's own reduction
                sse += st.sse
                sum_sq += st.sum_sq
                sum_vec += x.to(torch.float64).sum(dim=0)
            else:
                x64 = x[keep].to(torch.float64)
                diff = x64 - x_hat[keep].to(torch.float64)
                sse += (diff * diff).sum()
                sum_sq += (x64 * x64).sum()
                sum_vec += x64.sum(dim=0)
        if not cfg.save:
            continue

        g = batch["example_idx"] * T + batch["token_idx"]
        assert g[0].item() == prev_i + 1 and bool((g[1:] == g[:-1] + 1).all()), "batches must arrive in global order"
        prev_i = int(g[-1].item())

        live = (val != 0) & keep[:, None]  # what a dense -> CSR conversion of the masked f_x would keep
        cols, vals = idx[live].long(), val[live]
        value_sum.index_add_(0, cols, vals)
        n_pos.index_add_(0, cols, (vals > 0).to(torch.float32))
        csr_counts.append(live.sum(dim=1).cpu().numpy())
        csr_cols.append(cols.to(torch.int32).cpu().numpy())
        csr_data.append(vals.cpu().numpy())
        # first n_dists latents of every kept token, stored at row example_idx (last token of an example wins)
        head = torch.zeros(b, cfg.n_dists, device=device)
        small = live & (idx < cfg.n_dists)
        rows = torch.arange(b, device=device)[:, None].expand_as(idx)[small]
        head[rows, idx[small].long()] = val[small]
        distributions[batch["example_idx"][keep_host].numpy()] = head.cpu().numpy()[keep_host.numpy()]

    if cfg.save:
        counts = np.concatenate(csr_counts) if csr_counts else np.zeros(0, dtype=np.int64)
        indptr = np.zeros(counts.shape[0] + 1, dtype=np.int64)
        np.cumsum(counts, out=indptr[1:])
        nnz = int(indptr[-1])
        itype = np.int32 if max(nnz, S) < 2**31 else np.int64
        token_acts = scipy.sparse.csr_array(
            (np.concatenate(csr_data) if csr_data else np.zeros(0, np.float32),
             (np.concatenate(csr_cols) if csr_cols else np.zeros(0, np.int32)).astype(itype), indptr.astype(itype)),
            shape=(counts.shape[0], S))
        scipy.sparse.save_npz(fpaths.token_acts, token_acts)
        torch.save((value_sum / n_pos).cpu(), fpaths.mean_values)
        torch.save((n_pos / loader.n_samples).cpu(), fpaths.sparsity)
        torch.save(torch.from_numpy(distributions), fpaths.distributions)

    assert n_tokens > 0, "Inference dataloader yielded zero valid tokens; cannot compute metrics."
    sse_baseline = sum_sq.item() - torch.dot(sum_vec, sum_vec).item() / n_tokens
    if sse_baseline <= 0.0:
        raise RuntimeError(
            f"Baseline variance is non-positive (sse_baseline={sse_baseline:.6e}); cannot compute normalized MSE.")
    metrics = Metrics.from_accumulators(sse_recon=sse.item(), sse_baseline=sse_baseline, n_tokens=n_tokens, d_model=D)
    with open(fpaths.metrics, "w") as fd:
        json.dump(metrics.to_dict(), fd, indent=2)
    return metrics


def main(cfgs: Config | list[Config]) -> int:
    """Run the configs one after another in this process (the reference can also submit them to Slurm,
    inference.py:288-364; cluster submission is outside this package)."""
    cfgs = [cfgs] if isinstance(cfgs, Config) else list(cfgs)
    for i, c in enumerate(cfgs, start=1):
        if c.slurm_acct:
            raise NotImplementedError("Slurm submission is not part of saev_amd; run worker_fn on the node directly")
        logger.info("Running config %d/%d locally.", i, len(cfgs))
        worker_fn(c)
    logger.info("Jobs done.")
    return 0
