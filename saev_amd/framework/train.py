"""saev.framework.train's training surface (reference: src/saev/framework/train.py) on the HIP engine.

Kept from the reference: ``Config`` (same fields and defaults, train.py:50-105), ``make_saes``
(train.py:108-189), ``train`` (train.py:238-462), ``evaluate`` + ``EvalMetrics`` (train.py:465-618),
``split_cfgs`` (train.py:626-695), ``worker_fn`` (train.py:192-235) and the metric key names of the
log block (train.py:419-432).  Step order is the reference's: renormalise decoder rows -> objective ->
backward -> remove parallel grads -> clip -> (log) -> Adam with the lr set at the end of the previous
step (first step lr = 0) -> scheduler step.

Different by design: each SAE's step is ONE call into libsaev_amd.so (``saev_train_step``) instead of
autograd over dense GEMMs; activations come from a device-resident pool (saev_amd.data); with
``torch.distributed`` initialised (one process per GPU, RCCL) the batch is sharded over ranks and the
gradient buffer / fired flags are all-reduced (saev_amd.framework.ddp).  Slurm/submitit launching
(train.py:705-797) is out of scope; ``main`` runs groups in-process.
"""

from __future__ import annotations

import collections
import dataclasses
import json
import logging
import math
import os
import pathlib
import time
import typing as tp
import uuid

import torch
from torch import Tensor

from .. import data as saev_data
from .. import nn
from ..nn import modeling, objectives
from ..utils import scheduling
from ..utils.statistics import batch_entropy
from .ddp import DataParallelStepper

logger = logging.getLogger("train")


@dataclasses.dataclass(frozen=True, slots=True)
class Config:
    """Configuration for training a sparse autoencoder on transformer activations (train.py:50-105)."""

    train_data: saev_data.ShuffledConfig = saev_data.ShuffledConfig()
    val_data: saev_data.ShuffledConfig = saev_data.ShuffledConfig()
    n_train: int = 100_000_000
    n_val: int = 10_000_000
    sae: nn.SparseAutoencoderConfig = nn.SparseAutoencoderConfig()
    objective: nn.ObjectiveConfig = objectives.Matryoshka()
    n_sparsity_warmup: int = 0
    optim: tp.Literal["adam", "muon"] = "adam"
    lr: float = 0.0004
    n_lr_warmup: int = 500
    grad_clip: float = 1.0
    track: bool = True
    wandb_project: str = "saev"
    tags: tuple[str, ...] = ()
    log_every: int = 25
    runs_root: pathlib.Path = pathlib.Path("$SAEV_NFS/saev/runs")
    device: tp.Literal["cuda", "cpu"] = "cuda"
    seed: int = 42
    slurm_acct: str = ""
    slurm_partition: str = ""
    n_hours: float = 24.0
    mem_gb: int = 128
    log_to: str = os.path.join(".", "logs")
    # additive (not in the reference): the S x S dictionary-coherence metric costs 2*S^2*D flops per log step
    log_coherence: bool = True


# ------------------------------------------------------------------------------------------------
# run logging (the reference multiplexes W&B runs, utils/wandb.py; here: W&B if importable and
# track=True, else a JSONL file per SAE under runs_root)
# ------------------------------------------------------------------------------------------------


class RunLog:
    def __init__(self, cfgs: list[Config], n: int):
        self.ids = [uuid.uuid4().hex[:8] for _ in range(n)]
        self.records: list[list[tuple[int, dict]]] = [[] for _ in range(n)]
        self.summary: dict[str, object] = {}

    def log(self, metrics: list[dict[str, object]], *, step: int):
        for rec, m in zip(self.records, metrics):
            rec.append((step, m))

    def set_summary(self, key: str, value: object):
        self.summary[key] = value

    def finish(self) -> list[str]:
        return self.ids


# ------------------------------------------------------------------------------------------------
# distributed context
# ------------------------------------------------------------------------------------------------


def _dist():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


# ------------------------------------------------------------------------------------------------
# init
# ------------------------------------------------------------------------------------------------


def _datapoint_rows(dl, d_sae: int) -> tuple[Tensor, int]:
    """The activation rows a datapoint initialisation is built from: whole batches off the loader until max(d_sae, 65 536)
    rows -- or all the loader has, if that is fewer -- are in hand (train.py:141-157)."""
    have = getattr(dl, "n_samples", None)
    if have is not None and have < d_sae:
        raise ValueError(f"datapoint initialisation of {d_sae} latents needs at least as many activation rows; the loader holds {have}")
    want = max(d_sae, 65_536) if have is None else min(max(d_sae, 65_536), have)
    chunks, n = [], 0
    for batch in dl:
        chunks.append(batch["act"])
        n += len(batch["act"])
        if n >= want:
            break
    if n < want:
        raise RuntimeError(f"the loader ran dry after {n} of the {want} rows the datapoint initialisation asked for")
    return torch.cat(chunks, dim=0), want


def make_saes(
    cfgs: list[tuple[nn.SparseAutoencoderConfig, nn.ObjectiveConfig]], dl, device: torch.device | str = "cuda"
) -> tuple[torch.nn.ModuleList, torch.nn.ModuleList, list[dict[str, object]]]:
    """SAEs, their objectives and Adam's parameter groups (lr 0.0: the first step only warms the moments up), with the
    reference's optional datapoint initialisation (train.py:108-189): encoder columns are a blend
    ``reinit_blend * (shuffled, mean-centred activation rows) + (1 - reinit_blend) * Kaiming``, the decoder is their
    transpose with unit rows, and the encoder is set to the transpose of that normalised decoder.

    The random draws come from torch's global CPU generator in the reference's order -- one permutation of the sampled rows,
    one Kaiming matrix shared by all SAEs, one row permutation per SAE -- so seed + batches determine the initial weights
    exactly as they do there (fixture G10); the arithmetic runs on whatever device the activations live on."""
    saes = [nn.SparseAutoencoder(sae_cfg) for sae_cfg, _ in cfgs]
    objs = [nn.get_objective(obj_cfg) for _, obj_cfg in cfgs]
    param_groups = [{"params": sae.parameters(), "lr": 0.0} for sae in saes]
    modules = torch.nn.ModuleList(saes), torch.nn.ModuleList(objs), param_groups
    blends = [sae.cfg.reinit_blend for sae in saes]
    for p in blends:
        if not 0.0 <= p <= 1.0:
            raise ValueError(f"reinit_blend = {p} is outside [0, 1]")
    if not any(blends):
        logger.info("every reinit_blend is 0: parameters keep their Kaiming / zero initialisation")
        return modules
    d_sae = saes[0].cfg.d_sae
    if any(sae.cfg.d_sae != d_sae for sae in saes):
        raise ValueError("datapoint initialisation shares one sample of rows: every SAE of the group needs the same d_sae")
    with torch.no_grad():
        acts, n_rows = _datapoint_rows(dl, d_sae)
        acts = acts[torch.randperm(n_rows).to(acts.device)]
        centred = acts[:d_sae] - acts.mean(dim=0, keepdim=True)
        kaiming = torch.nn.init.kaiming_uniform_(torch.empty(centred.shape, dtype=centred.dtype)).to(acts.device)
        for sae, p in zip(saes, blends):
            order = torch.randperm(d_sae).to(acts.device)
            enc_cols = (p * centred[order] + (1 - p) * kaiming[order]).to("cpu")  # (d_sae, d_model): column s of W_enc
            sae.W_enc.data.copy_(enc_cols.T)
            if sae.cfg.reinit_enc_dec_tranpose:
                sae.W_dec.data.copy_(sae.W_enc.data.T)
            if sae.cfg.normalize_w_dec:
                sae.W_dec.data /= torch.norm(sae.W_dec.data, dim=1, keepdim=True)
            sae.W_enc.data.copy_(sae.W_dec.data.T)
    logger.info("datapoint initialisation done for %d SAE(s); mean blend %.2f", len(saes), sum(blends) / len(saes))
    return modules


# ------------------------------------------------------------------------------------------------
# train
# ------------------------------------------------------------------------------------------------


def _make_loader(cfg_data, device, rank, world, pool=None):
    return saev_data.ShuffledDataLoader(cfg_data, device=device, rank=rank, world_size=world, pool=pool)


def train(cfgs: list[Config], *, train_pool: Tensor | None = None, train_feed=None) -> tuple[torch.nn.ModuleList, torch.nn.ModuleList, RunLog, int]:
    """Train all SAEs of one parallel group on the same batches (train.py:238-462).

    ``train_pool`` optionally supplies an in-memory (n, d_model) activation pool instead of a shard dir; ``train_feed`` a
    ready loader-shaped object (e.g. data.ExtractionFeed: activations straight out of a transformer's forward hooks)."""
    if len(split_cfgs(cfgs)) != 1:
        raise ValueError(f"Configs are not parallelizeable: {cfgs}.")
    cfg = cfgs[0]
    if cfg.device != "cuda" or not torch.cuda.is_available():
        raise RuntimeError("saev_amd trains on a HIP device only (Config.device must be 'cuda'); there is no CPU path")
    for c in cfgs:
        if c.optim != "adam":
            raise NotImplementedError("optim='muon' is outside the MI355X hot path (Adam only)")
    dist, rank, world = _dist()
    device = torch.device("cuda", torch.cuda.current_device())

    dataloader = train_feed if train_feed is not None else _make_loader(cfg.train_data, device, rank, world, train_pool)
    limiter = scheduling.BatchLimiter(dataloader, cfg.n_train, rows_scale=world)
    torch.manual_seed(cfg.seed)
    # (datapoint init reads local rows: under data parallelism it gets a limiter that counts them as such -- the step
    # limiter ends a pass after n_train GLOBAL rows, i.e. n_train / world local ones, which can be fewer than the init needs)
    init_dl = limiter if world == 1 else scheduling.BatchLimiter(dataloader, cfg.n_train)
    saes, objs, _ = make_saes([(c.sae, c.objective) for c in cfgs], init_dl, device)
    run = RunLog(cfgs, len(cfgs))

    saes.train()
    saes = saes.to(device)
    objs.train()
    steppers, scheds, lrs = [], [], []
    # How the ranks exchange a step (framework/ddp.py: choose_exchange).  Default "auto", the selection bench.py makes: the
    # sparse step state when a rank holds at most 4 096 rows, else the gradient with the sharded tail (reduce-scatter, 1/world
    # of the tail per rank, all-gather) -- each taken only if a start-up self-check on a small SAE reproduces the plain
    # all-reduce path on every rank; the flat all-reduce with a replicated tail otherwise.  SAEV_AMD_DDP_TAIL
    # (replicated | sharded) and SAEV_AMD_DDP_EXCHANGE (dense | sparse) pin a choice.
    tail_mode, exchange = "replicated", "dense"
    if world > 1:
        from .ddp import choose_exchange

        tail_mode, exchange, report = choose_exchange(dist, world, rank, device, dataloader.local_batch,
                                                      tail=os.environ.get("SAEV_AMD_DDP_TAIL", "auto"),
                                                      exchange=os.environ.get("SAEV_AMD_DDP_EXCHANGE", "auto"))
        logger.info("data-parallel exchange: tail=%s exchange=%s (%s)", tail_mode, exchange, report)
    for sae, obj, c in zip(saes, objs, cfgs):
        if tail_mode == "sharded":
            sae._shard_world = world  # the engine lays its flat buffers out in `world` equal chunks per half
        if exchange == "sparse":
            # the gathered backward covers every rank's rows: ITS scratch is sized for the global batch, the forward's stays local
            sae._max_backward_rows = dataloader.local_batch * world
        eng = obj._bind(sae, dataloader.local_batch)
        if world > 1:  # identical replicas: rank 0's initial parameters everywhere
            dist.broadcast(eng.params, src=0)
        if steppers:  # one batch feeds every SAE of the group (train.py:334-348): the first engine's x statistics,
            eng.share_x(steppers[0].engine)  # centring and operand images serve the others
        steppers.append(DataParallelStepper(eng, dist, world, tail=tail_mode, exchange=exchange))
        scheds.append(scheduling.WarmupCosine(0.0, c.n_lr_warmup, c.lr, len(limiter), 0.0))
        lrs.append(0.0)  # first optimizer step is pure warm-up (train.py:118)
    dataloader.engine = steppers[0].engine
    # One rank, a resident pool: the loader hands over (pool, row indices) and the first SAE's step draws the batch in its own
    # first kernel (SaeEngine.train_step_gather) -- no gather pass, and for a single SAE the streamed preparation of the step.
    if world == 1 and hasattr(dataloader, "defer_gather"):
        dataloader.defer_gather = True

    global_step, n_patches_seen = 0, 0
    drawn_buf = None
    t_start = time.time()
    for batch in limiter:
        x = batch["act"]
        log_now = (global_step + 1) % cfg.log_every == 0
        drawn_by_step = False
        if x is None:  # (deferred gather)
            if log_now:
                x = steppers[0].engine.gather_rows(batch["pool"], batch["rows"])
            else:
                drawn_by_step = True
        n_patches_seen += (len(batch["rows"]) if x is None else len(x)) * world
        metrics = []
        spread = _loader_spread(batch, dataloader, dist, world) if log_now else {}
        for i, (sae, st, c) in enumerate(zip(saes, steppers, cfgs)):
            # Matryoshka cut points: sampled per SAE per step from torch's global CPU RNG, like the reference
            # (objectives.py:125); every rank draws the same sequence (same seed, same call order)
            if c.objective.n_prefixes > 1:
                st.engine.set_prefixes(objectives.sample_prefixes(c.sae.d_sae, c.objective.n_prefixes))
            if drawn_by_step and i == 0:
                # (the drawn batch lands in one buffer that every step reuses: it is read by this step's group only)
                if drawn_buf is None or drawn_buf.shape[0] != len(batch["rows"]):
                    drawn_buf = torch.empty(len(batch["rows"]), batch["pool"].shape[1], device=batch["pool"].device, dtype=torch.float32)
                x = st.engine.train_step_gather(batch["pool"], batch["rows"], lrs[i], c.grad_clip, out=drawn_buf)
                lrs[i] = scheds[i].step()
                continue
            if log_now:
                pre = {}
                st.train_step(x, lrs[i], c.grad_clip, pre_tail=lambda sae=sae, pre=pre, c=c: pre.update(_decoder_metrics(sae, c)))
                m = _log_metrics(sae, st.engine, x, lrs[i], n_patches_seen, c, pre, dataloader, dist, world)
                m.update(spread)
                metrics.append(m)
            else:
                st.train_step(x, lrs[i], c.grad_clip)
            lrs[i] = scheds[i].step()
        if log_now and rank == 0:
            run.log(metrics, step=global_step)
            logger.info("step %d: %s", global_step, ", ".join(f"{k}: {v:.5f}" for k, v in metrics[0].items()
                                                                if k.startswith("loss/") and isinstance(v, float)))
        global_step += 1
    for st in steppers:
        st.sync_params()  # (sharded tail: the last step's parameter gathers run on a side stream)
    logger.info("trained %d steps in %.1fs", global_step, time.time() - t_start)
    return saes, objs, run, global_step


def _loader_spread(batch, dataloader, dist=None, world: int = 1) -> dict[str, float]:
    """``loader/{example,token}_{entropy,entropy_normalized,coverage}`` of the log block (train.py:369-377) for the GLOBAL
    batch: under data parallelism the ranks' index vectors are all-gathered first (2 x 4 B per row, log steps only).  Feeds
    that carry no cache indices (e.g. an extraction feed) log nothing here."""
    ex, tk, md = batch.get("example_idx"), batch.get("token_idx"), getattr(dataloader, "metadata", None)
    if ex is None or tk is None or md is None:
        return {}
    if dist is not None and world > 1:
        both = torch.stack([ex.to(torch.int32), tk.to(torch.int32)])
        parts = [torch.empty_like(both) for _ in range(world)]
        dist.all_gather(parts, both)
        ex, tk = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    return batch_entropy(ex, tk, md.n_examples, md.content_tokens_per_example)


@torch.no_grad()
def _decoder_metrics(sae, cfg: Config) -> dict[str, object]:
    """The two log-block metrics that look at W_dec.  The reference evaluates them after the backward and BEFORE the
    optimizer step (train.py:365-442 precedes opt.step() at :444), i.e. on the rows normalised at the top of the step."""
    out = {"metrics/avg_decoder_row_norm": sae.W_dec.norm(dim=1).mean().item()}
    if cfg.log_coherence:
        out["metrics/dictionary_coherence"] = _coherence(sae.W_dec)
    return out


@torch.no_grad()
def _log_metrics(sae, eng, x: Tensor, lr: float, n_patches_seen: int, cfg: Config, pre: dict[str, object],
                 dataloader=None, dist=None, world: int = 1) -> dict[str, object]:
    """The reference's log block (train.py:365-442) from the step's device-side statistics.  Under data parallelism
    the batch is the union of the ranks' shards: every quantity is formed from sums that are all-reduced first (one
    collective of D + 8 doubles and one of d_sae flags per log step), so all ranks log the global-batch values."""
    st = eng.read_stats()
    n = x.shape[0]
    idx, val, x_hat = eng.last_codes(n)
    x64 = x.to(torch.float64)
    residual = x - x_hat
    r64 = residual.to(torch.float64)
    # [sum_vec (D) | sum x | sum r | sum r^2 (centred later) | sse | sum_sq | mse | aux | l0 | l1 | n]
    sums = torch.cat([x64.sum(dim=0), torch.stack([x64.sum(), r64.sum(), (r64 * r64).sum()]),
                      torch.tensor([st.sse, st.sum_sq, st.mse * n, st.aux * n, st.l0 * n, st.l1 * n, float(n)],
                                   dtype=torch.float64, device=x.device)])
    live = torch.zeros(sae.cfg.d_sae, device=x.device, dtype=torch.int32)
    live[idx[val.abs() > 1e-12].long()] = 1
    if dist is not None:
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(live, op=dist.ReduceOp.MAX)
    D = x.shape[1]
    sum_vec, (sx, sr, srr, sse, sum_sq, mse_n, aux_n, l0_n, l1_n, n_all) = sums[:D], sums[D:].tolist()
    sse_baseline = sum_sq - torch.dot(sum_vec, sum_vec).item() / n_all
    assert sse_baseline > 0, f"Batch baseline variance non-positive: sse_baseline={sse_baseline:.6e}"
    if dist is None:  # the reference's own expression, in fp32 (train.py:402)
        explained = (1 - residual.var() / x.var()).item()
    else:  # same quantity from the reduced sums (unbiased variances over all n_all * D elements)
        m = n_all * D
        explained = 1 - ((srr - sr * sr / m) / (m - 1)) / ((sum_sq - sx * sx / m) / (m - 1))
    mse, aux = mse_n / n_all, aux_n / n_all
    fill = dataloader.reservoir.fill() if getattr(dataloader, "reservoir", None) is not None else 1.0
    return {
        "loss/loss": mse + aux, "loss/mse": mse, "loss/l0": l0_n / n_all, "loss/l1": l1_n / n_all, "loss/sparsity": 0.0,
        "loss/aux": aux, "loss/n_dead": st.n_dead,
        "progress/n_patches_seen": n_patches_seen, "progress/learning_rate": lr,
        "metrics/explained_variance": explained,
        "metrics/dead_unit_pct": (live == 0).float().mean().item(),
        "metrics/grad_norm": st.grad_norm,
        "metrics/sse_sae": sse, "metrics/sse_baseline": sse_baseline,
        "metrics/normalized_mse": sse / sse_baseline,
        "loader/buffer_fill": fill,
        **pre,
    }


def _coherence(W: Tensor, block: int = 4096) -> float:
    """max_{i<j} |<w_i, w_j>| over unit-normalised decoder rows (train.py:410-414), in row blocks."""
    Wn = W / W.norm(dim=1, keepdim=True)
    best = 0.0
    S = Wn.shape[0]
    for lo in range(0, S, block):
        g = (Wn[lo : lo + block] @ Wn[lo:].T).abs()
        g = torch.triu(g, diagonal=1)
        best = max(best, g.max().item())
    return best


# ------------------------------------------------------------------------------------------------
# evaluate
# ------------------------------------------------------------------------------------------------


@dataclasses.dataclass(frozen=True)
class EvalMetrics:
    """train.py:465-507."""

    l0: float
    l1: float
    mse: float
    normalized_mse: float
    sse_sae: float
    sse_baseline: float
    n_dead: int
    n_almost_dead: int
    n_dense: int
    freqs: Tensor
    mean_values: Tensor
    almost_dead_threshold: float
    dense_threshold: float

    def for_wandb(self) -> dict[str, object]:
        d = dataclasses.asdict(self)
        d["freqs"] = d["freqs"].tolist()
        d["mean_values"] = d["mean_values"].tolist()
        return {f"eval/{k}": v for k, v in d.items()}


@torch.no_grad()
def evaluate(cfgs: list[Config], saes: torch.nn.ModuleList, objs: torch.nn.ModuleList, *,
             val_pool: Tensor | None = None) -> list[EvalMetrics]:
    """Eval-mode pass over the validation feed (train.py:510-618): fp64 baseline sums, SAE SSE, per-latent
    firing counts (f > 0) and value sums, dead / almost-dead (<1e-7) / dense (>1e-2) counts."""
    if len(split_cfgs(cfgs)) != 1:
        raise ValueError(f"Configs are not parallelizeable: {cfgs}.")
    saes.eval()
    objs.eval()
    cfg = cfgs[0]
    dist, rank, world = _dist()
    device = torch.device("cuda", torch.cuda.current_device())
    dataloader = _make_loader(cfg.val_data, device, rank, world, val_pool)
    n_val = min(dataloader.n_samples, cfg.n_val)
    limiter = scheduling.BatchLimiter(dataloader, n_val, rows_scale=world)
    S, D = saes[0].cfg.d_sae, saes[0].cfg.d_model
    n_fired = torch.zeros(len(cfgs), S, device=device)
    values = torch.zeros(len(cfgs), S, device=device)
    acc = torch.zeros(len(cfgs), 4, dtype=torch.float64, device=device)  # l0*b, l1*b, mse*b, sse
    sum_sq = torch.zeros((), dtype=torch.float64, device=device)
    sum_vec = torch.zeros(D, dtype=torch.float64, device=device)
    n_tokens = 0
    for batch in limiter:
        x = batch["act"]
        b = x.shape[0]
        x64 = x.to(torch.float64)
        sum_vec += x64.sum(dim=0)
        n_tokens += b
        for i, (sae, obj) in enumerate(zip(saes, objs)):
            eng = obj._bind(sae, b)
            n_pre = cfgs[i].objective.n_prefixes
            eng.set_prefixes(objectives.sample_prefixes(sae.cfg.d_sae, n_pre) if n_pre > 1 else None)
            eng.step_forward(x, training=False)
            st = eng.read_stats()
            if i == 0:
                sum_sq += st.sum_sq
            idx, val, _ = eng.last_codes(b)
            pos = val > 0
            n_fired[i].index_add_(0, idx[pos].long(), torch.ones_like(val[pos]))
            values[i].index_add_(0, idx.reshape(-1).long().clamp_min(0), val.reshape(-1))
            acc[i] += torch.tensor([st.l0 * b, st.l1 * b, st.mse * b, st.sse], dtype=torch.float64, device=device)
    if dist is not None:
        t = torch.tensor([float(n_tokens)], dtype=torch.float64, device=device)
        for buf in (n_fired, values, acc, sum_sq, sum_vec, t):
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        n_tokens = int(t.item())
    assert n_tokens > 0, "Validation dataloader yielded zero tokens; cannot compute normalized MSE."
    sse_baseline = (sum_sq - torch.dot(sum_vec, sum_vec) / n_tokens).item()
    assert sse_baseline > 0, f"Validation baseline variance non-positive: sse_baseline={sse_baseline:.6e}"
    freqs = (n_fired / n_tokens).cpu()
    mean_values = (values / n_fired).cpu()
    acc = acc.cpu()
    out = []
    for i in range(len(cfgs)):
        out.append(EvalMetrics(
            l0=acc[i, 0].item() / n_tokens, l1=acc[i, 1].item() / n_tokens, mse=acc[i, 2].item() / n_tokens,
            normalized_mse=acc[i, 3].item() / sse_baseline, sse_sae=acc[i, 3].item(), sse_baseline=sse_baseline,
            n_dead=int((freqs[i] == 0).sum()), n_almost_dead=int((freqs[i] < 1e-7).sum()),
            n_dense=int((freqs[i] > 1e-2).sum()), freqs=freqs[i], mean_values=mean_values[i],
            almost_dead_threshold=1e-7, dense_threshold=1e-2,
        ))
    return out


# ------------------------------------------------------------------------------------------------
# grouping / entry points
# ------------------------------------------------------------------------------------------------

CANNOT_PARALLELIZE = {
    "train_data", "val_data", "n_train", "n_val", "track", "wandb_project", "tags", "log_every", "runs_root",
    "device", "slurm_acct", "slurm_partition", "n_hours", "mem_gb", "log_to", "sae.d_sae", "sae.d_model",
    "sae.reinit_blend", "sae.reinit_enc_dec_tranpose",
}


def _hashable(v):
    if isinstance(v, dict):
        return tuple(sorted((k, _hashable(x)) for k, x in v.items()))
    if isinstance(v, (list, tuple)):
        return tuple(_hashable(x) for x in v)
    return v


def _get(d: dict, dotted: str):
    for part in dotted.split("."):
        d = d[part]
    return d


def _parallel_key(cfg: Config):
    d = dataclasses.asdict(cfg)
    for split in ("train_data", "val_data"):
        d[split] = dict(d[split], seed="IGNORED_FOR_PARALLEL")
    return tuple((k, _hashable(_get(d, k))) for k in sorted(CANNOT_PARALLELIZE))


def split_cfgs(cfgs: list[Config]) -> list[list[Config]]:
    """Group configs that may share one data stream (train.py:669-695): equal on every key in
    CANNOT_PARALLELIZE, loader seeds ignored for grouping and then set from ``cfg.seed``."""
    groups = collections.defaultdict(list)
    for cfg in cfgs:
        groups[_parallel_key(cfg)].append(cfg)
    return [
        [dataclasses.replace(c, train_data=dataclasses.replace(c.train_data, seed=c.seed),
                             val_data=dataclasses.replace(c.val_data, seed=c.seed)) for c in group]
        for _, group in sorted(groups.items(), key=lambda kv: repr(kv[0]))
    ]


def _jsonable(o):
    if dataclasses.is_dataclass(o) and not isinstance(o, type):
        return {f.name: _jsonable(getattr(o, f.name)) for f in dataclasses.fields(o)}
    if isinstance(o, pathlib.PurePath):
        return str(o)
    if isinstance(o, (list, tuple)):
        return [_jsonable(x) for x in o]
    if isinstance(o, dict):
        return {k: _jsonable(v) for k, v in o.items()}
    return o


def worker_fn(cfgs: list[Config], *, train_pool: Tensor | None = None, val_pool: Tensor | None = None) -> list[str]:
    """Train, evaluate, and write ``<runs_root>/<id>/checkpoint/{sae.pt, config.json}`` per SAE
    (train.py:192-235; run-dir layout of disk.py:98-128)."""
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s] [%(levelname)s] [%(name)s] %(message)s")
    saes, objs, run, steps = train(cfgs, train_pool=train_pool)
    evals = evaluate(cfgs, saes, objs, val_pool=val_pool)
    run.log([m.for_wandb() for m in evals], step=steps)
    ids = run.finish()
    _, rank, _ = _dist()
    if rank != 0:
        return ids
    for cfg, rid, metric, sae, rec in zip(cfgs, ids, evals, saes, run.records):
        logger.info("Checkpoint %s has %d dense, %d dead, %d almost dead features", rid, metric.n_dense, metric.n_dead,
                    metric.n_almost_dead)
        run_dir = pathlib.Path(os.path.expandvars(str(cfg.runs_root))) / rid
        (run_dir / "checkpoint").mkdir(parents=True, exist_ok=True)
        (run_dir / "links").mkdir(exist_ok=True)
        (run_dir / "inference").mkdir(exist_ok=True)
        for name, target in (("train-shards", cfg.train_data.shards), ("val-shards", cfg.val_data.shards)):
            link = run_dir / "links" / name
            if not link.exists() and pathlib.Path(os.path.expandvars(str(target))).exists():
                link.symlink_to(pathlib.Path(os.path.expandvars(str(target))))
        nn.dump(run_dir / "checkpoint" / "sae.pt", sae)
        with open(run_dir / "checkpoint" / "config.json", "w") as fd:
            json.dump(_jsonable(cfg), fd, indent=2)
        with open(run_dir / "metrics.jsonl", "w") as fd:
            for step, m in rec:
                fd.write(json.dumps({"step": step, **{k: v for k, v in m.items() if not isinstance(v, list)}}) + "\n")
        logger.info("Dumped checkpoint to '%s'.", run_dir / "checkpoint" / "sae.pt")
    return ids


def main(cfgs: list[Config]) -> list[str]:
    """Run every parallel group in-process (the reference submits them to Slurm, train.py:705-797)."""
    ids: list[str] = []
    for group in split_cfgs(cfgs):
        ids.extend(worker_fn(group))
    return ids
