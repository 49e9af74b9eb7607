"""Generate golden vectors by RUNNING the upstream reference (build container only).

Test infrastructure.  Imports ``/root/reference/src/saev`` through ``oracle/_refshim.py``, drives the
reference's own ``SparseAutoencoder`` / ``MatryoshkaObjective`` / ``train()`` / ``evaluate()`` on
seeded synthetic inputs and stores inputs + outputs as small ``.npz`` fixtures under
``tests/golden/``.  The fixtures are data (tensors in, tensors out); the reference's code never
leaves this container.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Fixture index:
  G1  encode_topk      x, W_enc, b_enc -> h, f (dense), selected values
  G2  decode           f, W_dec, b_dec, prefixes -> x_hats            (P=1 and P=3)
  G3  mse              x_hat, x -> elementwise mse                    (normal + huge magnitude)
  G4  auxk             value + grads for n_dead <, =, > k_aux
  G5  objective        full fwd/bwd: losses + 4 param grads, with and without dead latents
  G6  rpg              remove_parallel_grads
  G7  clip             clip_grad_norm_ (coef < 1 and coef = 1)
  G8  adam             5 fused-Adam steps incl. the lr=0 first step
  G9  train_a/train_b  20ish-step train() trajectories + evaluate() metrics
  G10 make_saes        datapoint initialisation (reinit_blend 0.8 and 1.0, two SAEs sharing one stream) from fixed
                       batches under torch.manual_seed: W_enc / W_dec of the reference's make_saes
  G11 schedule         WarmupCosine lr lists and BatchLimiter step counts
  G12 checkpoint       header bytes/JSON written by the reference's nn.dump
  G13 matryoshka       objective fwd/bwd with 4 fixed prefixes (with and without dead latents)
  G9c train_c          the train_b run with grad_clip = 0.02: the clip coefficient is < 1 on every step
  G15 sample_prefixes  the reference's Matryoshka prefix draws under fixed seeds
  G16 legacy_ckpt      headers in every older checkpoint layout the reference's nn.load still reads (pre-schema, schema 1 in
                       both of its forms, schemas 2-4) and the config the reference's loader makes of each
  G17 batch_entropy    the reference's loader-coverage metrics of the log block (utils/statistics.py) on seeded index batches
  G14 inference        the reference's framework/inference.worker_fn over a small protocol-2.1 cache (with and
                       without labels.bin / ignore_labels): CSR token_acts, mean_values, sparsity, distributions,
                       metrics.json; plus Metadata.hash and IndexMap known answers for the same cache
"""

import dataclasses
import io
import json
import math
import os
import pathlib
import sys

import numpy as np
import torch

HERE = pathlib.Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import _refshim  # noqa: E402

OUT = pathlib.Path(os.environ["SAEV_GOLDEN_OUT"]) if os.environ.get("SAEV_GOLDEN_OUT") else HERE.parent / "tests" / "golden"


def npz(name, **arrays):
    OUT.mkdir(parents=True, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(OUT / f"{name}.npz", **conv)
    print(f"wrote {name}.npz  ({sum(a.nbytes for a in conv.values())/1e6:.2f} MB raw)")


def lowrank_data(n, d, seed, n_atoms_mult=4, sparsity=16, noise=0.1):
    """x = A s + noise*eps with unit-norm atoms and sparse non-negative codes."""
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(d, n_atoms_mult * d, generator=g)
    A = A / A.norm(dim=0, keepdim=True)
    s = torch.zeros(n, n_atoms_mult * d)
    for i in range(n):
        idx = torch.randperm(n_atoms_mult * d, generator=g)[:sparsity]
        s[i, idx] = torch.empty(sparsity).exponential_(1.0, generator=g)
    return s @ A.T + noise * torch.randn(n, d, generator=g)


def make_sae(ref, d, s, k, k_aux=512, alpha=1 / 32, seed=0, **kw):
    torch.manual_seed(seed)
    cfg = ref.modeling.SparseAutoencoderConfig(
        d_model=d, d_sae=s, reinit_blend=0.0,
        activation=ref.modeling.TopK(top_k=k, aux=ref.modeling.AuxK(k_aux=k_aux, alpha=alpha)), **kw,
    )
    sae = ref.modeling.SparseAutoencoder(cfg)
    with torch.no_grad():  # make the biases non-trivial and decouple W_enc from W_dec^T
        sae.b_enc.copy_(0.05 * torch.randn(s))
        sae.b_dec.copy_(0.1 * torch.randn(d))
        sae.W_enc.add_(0.02 * torch.randn(d, s))
    return sae


def params_of(sae, prefix="p_"):
    return {prefix + k: v.detach().clone() for k, v in sae.state_dict().items()}


def g1_g2_g3(ref):
    sae = make_sae(ref, 48, 320, 8, seed=1)
    x = lowrank_data(96, 48, seed=2)
    enc = sae.encode(x)
    sel = enc.f_x != 0
    npz("g1_encode_topk", x=x, **params_of(sae), h=enc.h_x, f=enc.f_x, n_sel=sel.sum(1))
    # k >= d_sae keeps everything, negatives included
    sae2 = make_sae(ref, 16, 24, 64, seed=3)
    x2 = torch.randn(5, 16, generator=torch.Generator().manual_seed(4))
    enc2 = sae2.encode(x2)
    npz("g1_encode_topk_kfull", x=x2, **params_of(sae2), h=enc2.h_x, f=enc2.f_x)

    f = enc.f_x
    xh1 = sae.decode(f)
    pref = torch.tensor([7, 100, 320], dtype=torch.int64)
    xh3 = sae.decode(f, prefixes=pref)
    npz("g2_decode", f=f, W_dec=sae.W_dec, b_dec=sae.b_dec, x_hats_p1=xh1, prefixes=pref, x_hats_p3=xh3)

    g = torch.Generator().manual_seed(5)
    xa, xb = torch.randn(6, 3, 10, generator=g), torch.randn(6, 3, 10, generator=g)
    big = 1e20 * torch.randn(4, 2, 8, generator=g)
    big_hat = big * (1 + 1e-3 * torch.randn(4, 2, 8, generator=g))
    npz(
        "g3_mse", x_hat=xa, x=xb, mse=ref.objectives.mean_squared_err(xa, xb),
        big_x_hat=big_hat, big_x=big, big_mse=ref.objectives.mean_squared_err(big_hat, big),
    )


def g4_auxk(ref):
    d, s = 32, 256
    for tag, n_dead, k_aux in (("lt", 5, 16), ("eq", 16, 16), ("gt", 60, 16)):
        sae = make_sae(ref, d, s, 8, k_aux=k_aux, alpha=1 / 32, seed=10 + n_dead)
        sae.train()
        x = lowrank_data(40, d, seed=20 + n_dead)
        g = torch.Generator().manual_seed(30 + n_dead)
        dead = torch.zeros(s, dtype=torch.bool)
        dead[torch.randperm(s, generator=g)[:n_dead]] = True
        h = (x @ sae.W_enc.detach() + sae.b_enc.detach()).requires_grad_(True)
        x_hat = torch.randn(40, 1, d, generator=g)
        out = ref.modeling.SparseAutoencoder.Output(h_x=h, f_x=h, x_hats=x_hat)
        loss = sae.cfg.activation.aux.loss(sae=sae, x=x, out=out, dead_mask=dead)
        loss.backward()
        npz(
            f"g4_auxk_{tag}", x=x, h=h, x_hat=x_hat[:, 0], dead=dead, k_aux=k_aux, alpha=1 / 32,
            W_dec=sae.W_dec, b_dec=sae.b_dec, loss=loss, g_h=h.grad, g_W_dec=sae.W_dec.grad,
            g_b_dec=sae.b_dec.grad,
            # (round 4: the encoder that produced h, so that the HIP step can re-derive h from x: tests/test_gpu_known_answers.py)
            W_enc=sae.W_enc, b_enc=sae.b_enc,
        )


def g5_objective(ref):
    d, s, k, b = 64, 512, 8, 128
    for tag, thr, k_aux in (("nodead", 10_000_000, 512), ("dead", 300, 24), ("dead_few", 300, 512)):
        sae = make_sae(ref, d, s, k, k_aux=k_aux, seed=40)
        sae.train()
        obj = ref.objectives.get_objective(ref.objectives.Matryoshka(n_prefixes=1, dead_threshold_tokens=thr))
        obj.train()
        toks0 = torch.zeros(s, dtype=torch.int64)
        if tag == "dead":
            toks0[torch.randperm(s, generator=torch.Generator().manual_seed(41))[:200]] = 250
        if tag == "dead_few":
            toks0[torch.randperm(s, generator=torch.Generator().manual_seed(42))[:30]] = 250
        obj.toks_since_active = toks0.clone()
        x = lowrank_data(b, d, seed=43)
        loss, out = obj(sae, x)
        loss.loss.backward()
        npz(
            f"g5_objective_{tag}", x=x, **params_of(sae), toks_before=toks0, toks_after=obj.toks_since_active,
            thr=thr, k=k, k_aux=k_aux, alpha=1 / 32, mse=loss.mse, aux=loss.aux, l0=loss.l0, l1=loss.l1,
            n_dead=int(loss.n_dead), h=out.h_x, f=out.f_x, x_hat=out.x_hats[:, -1],
            g_W_dec=sae.W_dec.grad, g_b_dec=sae.b_dec.grad, g_W_enc=sae.W_enc.grad, g_b_enc=sae.b_enc.grad,
        )
    # eval mode: no tracking, aux == 0
    sae = make_sae(ref, d, s, k, seed=44).eval()
    obj = ref.objectives.get_objective(ref.objectives.Matryoshka(n_prefixes=1)).eval()
    x = lowrank_data(b, d, seed=45)
    with torch.no_grad():
        loss, out = obj(sae, x)
    npz("g5_objective_eval", x=x, **params_of(sae), k=k, mse=loss.mse, aux=loss.aux, l0=loss.l0, l1=loss.l1,
        n_dead=int(loss.n_dead), f=out.f_x, x_hat=out.x_hats[:, -1])


def g13_matryoshka(ref):
    """Objective fwd/bwd with FIXED Matryoshka prefixes (the sampler is patched to return them)."""
    d, s, k, b = 64, 512, 8, 128
    fixed = torch.tensor([5, 60, 200, 512], dtype=torch.int64)
    orig = ref.objectives.sample_prefixes
    ref.objectives.sample_prefixes = lambda d_sae, n_prefixes, *a, **kw: fixed
    try:
        for tag, thr, k_aux in (("nodead", 10_000_000, 512), ("dead", 300, 24)):
            sae = make_sae(ref, d, s, k, k_aux=k_aux, seed=90)
            sae.train()
            obj = ref.objectives.get_objective(ref.objectives.Matryoshka(n_prefixes=4, dead_threshold_tokens=thr))
            obj.train()
            toks0 = torch.zeros(s, dtype=torch.int64)
            if tag == "dead":
                toks0[torch.randperm(s, generator=torch.Generator().manual_seed(91))[:150]] = 250
            obj.toks_since_active = toks0.clone()
            x = lowrank_data(b, d, seed=92)
            loss, out = obj(sae, x)
            loss.loss.backward()
            npz(
                f"g13_matryoshka_{tag}", x=x, **params_of(sae), prefixes=fixed, toks_before=toks0,
                toks_after=obj.toks_since_active, thr=thr, k=k, k_aux=k_aux, alpha=1 / 32, mse=loss.mse, aux=loss.aux,
                l0=loss.l0, l1=loss.l1, n_dead=int(loss.n_dead), x_hats=out.x_hats, f=out.f_x,
                g_W_dec=sae.W_dec.grad, g_b_dec=sae.b_dec.grad, g_W_enc=sae.W_enc.grad, g_b_enc=sae.b_enc.grad,
            )
    finally:
        ref.objectives.sample_prefixes = orig


def g6_g7_g8(ref):
    sae = make_sae(ref, 24, 96, 4, seed=50)
    g = torch.Generator().manual_seed(51)
    sae.W_dec.grad = torch.randn(96, 24, generator=g)
    with torch.no_grad():
        sae.W_dec[5].zero_()  # zero-norm row is skipped
    before = sae.W_dec.grad.clone()
    sae.remove_parallel_grads()
    npz("g6_rpg", W_dec=sae.W_dec, g_in=before, g_out=sae.W_dec.grad)

    for tag, scale in (("clipped", 3.0), ("unclipped", 1e-3)):
        ps = [torch.nn.Parameter(torch.zeros(*shape)) for shape in ((96, 24), (24,), (24, 96), (96,))]
        for p in ps:
            p.grad = scale * torch.randn(*p.shape, generator=g)
        gin = [p.grad.clone() for p in ps]
        total = torch.nn.utils.clip_grad_norm_(ps, max_norm=1.0)
        npz(f"g7_clip_{tag}", total=total, **{f"in{i}": t for i, t in enumerate(gin)},
            **{f"out{i}": p.grad for i, p in enumerate(ps)})

    p = torch.nn.Parameter(torch.randn(37, 11, generator=g))
    p0 = p.detach().clone()
    opt = torch.optim.Adam([{"params": [p], "lr": 0.0}], fused=True)
    lrs = [0.0, 1e-3, 2e-3, 4e-4, 4e-4]
    grads, ps, ms, vs = [], [], [], []
    for lr in lrs:
        opt.param_groups[0]["lr"] = lr
        p.grad = torch.randn(37, 11, generator=g) * 0.1
        grads.append(p.grad.clone())
        opt.step()
        st = opt.state[p]
        ps.append(p.detach().clone()); ms.append(st["exp_avg"].clone()); vs.append(st["exp_avg_sq"].clone())
    npz("g8_adam", p0=p0, lrs=np.array(lrs), grads=torch.stack(grads), p=torch.stack(ps), m=torch.stack(ms),
        v=torch.stack(vs))


class MemLoader:
    """In-memory stand-in for the reference's ShuffledDataLoader (batch dict of shuffled.py:385-391)."""

    @dataclasses.dataclass(frozen=True)
    class Meta:
        n_examples: int
        content_tokens_per_example: int

    def __init__(self, acts, batch_size):
        self.acts, self.batch_size, self.drop_last = acts, batch_size, False
        self.n_samples = len(acts)
        self.metadata = self.Meta(len(acts), 1)
        self.manager_pid = -1
        self.reservoir = None

    def __len__(self):
        return math.ceil(self.n_samples / self.batch_size)

    def __iter__(self):
        for i in range(0, self.n_samples, self.batch_size):
            a = self.acts[i : i + self.batch_size]
            yield {"act": a, "example_idx": torch.arange(i, i + len(a), dtype=torch.int32),
                   "token_idx": torch.zeros(len(a), dtype=torch.int32)}


def g9_train(ref, tag, d, s, k, bsz, n_rows, n_train, thr, k_aux, lr, n_warm, grad_clip=1.0):
    import wandb

    T = ref.train
    acts = lowrank_data(n_rows, d, seed=60 + d)
    val = lowrank_data(n_rows // 2, d, seed=61 + d)
    loaders = {"train": acts, "val": val}

    class Run:
        id = "gold0001"
        summary = {}
        logs = []

        def log(self, m, step=None):
            self.logs.append((step, m))

        def finish(self):
            pass

    run = Run()
    wandb.init = lambda **kw: run

    cfg_data_train = ref.data.ShuffledConfig(batch_size=bsz)
    state = {"which": "train"}

    def fake_loader(cfg):
        which = state["which"]
        return MemLoader(loaders[which], bsz)

    ref.data.ShuffledDataLoader = fake_loader
    import saev.data as sd

    sd.ShuffledDataLoader = fake_loader
    # the monitor/entropy helpers are loader-observability, not part of the path: stub to {}
    # (both are put back after the run)
    orig_monitor, orig_entropy = T.DataloaderMonitor, T.statistics.calc_batch_entropy
    T.DataloaderMonitor = lambda dl: type("M", (), {"compute": lambda self, now=None: {}})()
    T.statistics.calc_batch_entropy = lambda *a, **kw: {}

    cfg = T.Config(
        train_data=cfg_data_train, val_data=cfg_data_train, n_train=n_train, n_val=10**9,
        sae=ref.modeling.SparseAutoencoderConfig(
            d_model=d, d_sae=s, reinit_blend=0.0,
            activation=ref.modeling.TopK(top_k=k, aux=ref.modeling.AuxK(k_aux=k_aux, alpha=1 / 32)),
        ),
        objective=ref.objectives.Matryoshka(n_prefixes=1, dead_threshold_tokens=thr),
        lr=lr, n_lr_warmup=n_warm, grad_clip=grad_clip, track=False, log_every=1, device="cpu",
    )
    init = {}
    orig_make = T.make_saes

    def make_and_record(cfgs, dl):
        saes, objs, pgs = orig_make(cfgs, dl)
        init.update({k: v.detach().clone() for k, v in saes[0].state_dict().items()})
        return saes, objs, pgs

    T.make_saes = make_and_record
    torch.manual_seed(cfg.seed)
    try:
        saes, objs, _, steps = T.train([cfg])
        T.make_saes = orig_make
        final = {k: v.detach().clone() for k, v in saes[0].state_dict().items()}
        toks = objs[0].toks_since_active.clone()
        state["which"] = "val"
        ev = T.evaluate([cfg], saes, objs)[0]
    finally:
        T.make_saes = orig_make
        T.DataloaderMonitor, T.statistics.calc_batch_entropy = orig_monitor, orig_entropy

    keys = ["loss/mse", "loss/aux", "loss/l0", "loss/l1", "loss/n_dead", "loss/loss", "metrics/grad_norm",
            "progress/learning_rate", "metrics/normalized_mse", "metrics/sse_sae", "metrics/sse_baseline",
            "metrics/explained_variance", "metrics/dead_unit_pct", "metrics/avg_decoder_row_norm",
            "metrics/dictionary_coherence"]
    traj = {}
    for key in keys:
        vals = []
        for _, m in run.logs:
            v = m[key]
            vals.append(float(v.item() if hasattr(v, "item") else v))
        traj["log_" + key.replace("/", "_")] = np.array(vals, dtype=np.float64)
    npz(
        f"g9_train_{tag}", acts=acts, val=val, d=d, s=s, k=k, bsz=bsz, n_train=n_train, thr=thr, k_aux=k_aux,
        lr=lr, n_warm=n_warm, grad_clip=grad_clip, n_steps=steps, toks_final=toks,
        **{"init_" + k_: v for k_, v in init.items()}, **{"final_" + k_: v for k_, v in final.items()}, **traj,
        ev_l0=ev.l0, ev_l1=ev.l1, ev_mse=ev.mse, ev_normalized_mse=ev.normalized_mse, ev_sse_sae=ev.sse_sae,
        ev_sse_baseline=ev.sse_baseline, ev_n_dead=ev.n_dead, ev_n_almost_dead=ev.n_almost_dead,
        ev_n_dense=ev.n_dense, ev_freqs=ev.freqs, ev_mean_values=torch.nan_to_num(ev.mean_values, nan=-1.0),
    )
    print(f"  {tag}: {steps} steps, final mse {traj['log_loss_mse'][-1]:.6f}, n_dead {traj['log_loss_n_dead'][-1]}, "
          f"eval nmse {ev.normalized_mse:.6f}")


def g15_sample_prefixes(ref):
    """The reference's Matryoshka prefix draws (objectives.py:159-201) under torch.manual_seed: consecutive calls, as
    the train loop makes them (one per SAE per step)."""
    out = {}
    cases = [(512, 4), (1024, 10), (32768, 10), (6144, 2), (64, 64), (300, 1)]
    for seed in (0, 1, 42, 1234):
        torch.manual_seed(seed)
        for d_sae, n in cases:
            out[f"s{seed}_{d_sae}_{n}"] = torch.stack([ref.objectives.sample_prefixes(d_sae, n) for _ in range(3)])
    torch.manual_seed(7)
    out["s7_alt_1000_8"] = torch.stack([ref.objectives.sample_prefixes(1000, 8, pareto_power=1.5) for _ in range(3)])
    npz("g15_sample_prefixes", cases=np.array(cases), seeds=np.array([0, 1, 42, 1234]), **out)


def g17_batch_entropy(ref):
    """The reference's calc_batch_entropy on seeded index batches (uniform, skewed, single-token support)."""
    import importlib

    st = importlib.import_module("saev.utils.statistics")
    g = torch.Generator().manual_seed(170)
    out = {}
    for tag, (n_ex, n_tok, b) in {"a": (1000, 256, 4096), "b": (37, 16, 64), "c": (5, 1, 200), "d": (1, 3, 7)}.items():
        e = torch.randint(0, n_ex, (b,), generator=g, dtype=torch.int32)
        t = (torch.rand(b, generator=g) ** 2 * n_tok).to(torch.int32).clamp_(max=n_tok - 1)  # skewed towards the first positions
        m = st.calc_batch_entropy(e, t, n_ex, n_tok)
        out[f"{tag}_example_idx"], out[f"{tag}_token_idx"] = e, t
        out[f"{tag}_support"] = np.array([n_ex, n_tok])
        out[f"{tag}_keys"] = np.array(sorted(m))
        out[f"{tag}_vals"] = np.array([m[k] for k in sorted(m)], dtype=np.float64)
    npz("g17_batch_entropy", **out)


def g10_make_saes(ref):
    d, s, bsz = 24, 96, 64
    acts = lowrank_data(320, d, seed=100)
    cfgs = [(ref.modeling.SparseAutoencoderConfig(d_model=d, d_sae=s, reinit_blend=blend,
                                                  activation=ref.modeling.TopK(top_k=4)),
             ref.objectives.Matryoshka(n_prefixes=1)) for blend in (0.8, 1.0)]
    torch.manual_seed(1234)
    saes, _, groups = ref.train.make_saes(cfgs, MemLoader(acts, bsz))
    assert len(groups) == 2 and groups[0]["lr"] == 0.0
    npz("g10_make_saes", acts=acts, bsz=bsz, seed=1234, blends=np.array([0.8, 1.0]),
        **{f"W_enc_{i}": sae.W_enc.detach() for i, sae in enumerate(saes)},
        **{f"W_dec_{i}": sae.W_dec.detach() for i, sae in enumerate(saes)},
        **{f"b_enc_{i}": sae.b_enc.detach() for i, sae in enumerate(saes)})


def g11_schedule(ref):
    S = ref.scheduling
    out = {}
    for tag, args, n in (("a", (0.0, 5, 4e-4, 16, 0.0), 20), ("b", (0.0, 500, 4e-4, 100, 0.0), 110),
                         ("c", (0.1, 100, 0.9, 1000, 0.0), 1005)):
        sc = S.WarmupCosine(*args)
        out[f"lr_{tag}"] = np.array([sc.step() for _ in range(n)])
        out[f"args_{tag}"] = np.array(args)
    counts = []
    for n_rows, bsz, n_train in ((1024, 128, 2048), (1000, 128, 2048), (1024, 128, 1024), (300, 128, 1000),
                                 (4096, 128, 1000)):
        dl = MemLoader(torch.zeros(n_rows, 2), bsz)
        lim = S.BatchLimiter(dl, n_train)
        sizes = [len(b["act"]) for b in lim]
        counts.append([n_rows, bsz, n_train, len(lim), len(sizes), sum(sizes)])
    out["limiter"] = np.array(counts)
    npz("g11_schedule", **out)


def g12_checkpoint(ref):
    sae = make_sae(ref, 16, 48, 4, k_aux=7, alpha=0.125, seed=70)
    buf = pathlib.Path("/tmp/_gold_sae.pt")
    ref.modeling.dump(buf, sae)
    raw = buf.read_bytes()
    header, _, rest = raw.partition(b"\n")
    hdr = json.loads(header)
    hdr["commit"] = "unknown"  # container-specific
    sd = torch.load(io.BytesIO(rest), weights_only=True)
    npz("g12_checkpoint", header_json=np.frombuffer(json.dumps(hdr, sort_keys=True).encode(), dtype=np.uint8),
        keys=np.array(list(sd.keys())), **{"sd_" + k: v for k, v in sd.items()})
    buf.unlink()


def g16_legacy_checkpoints(ref):
    """Legacy header layouts (described in the reference loader's branches, modeling.py:586-645), each written in front of a
    state dict and read back by the REFERENCE's nn.load; the fixture holds the header and the loaded config as the
    reference's own dump would write it."""
    M = ref.modeling
    tree = lambda act: M._serialize_dataclass(act)
    old_tree = {"cls": "TopK", "params": {"kind": "top-k", "top_k": 6, "sparsity": {}, "aux": {"cls": "AuxK", "params": {"kind": "auxk", "k_aux": 9, "alpha": 0.25}}}}
    old_relu = {"cls": "Relu", "params": {"key": "relu", "sparsity": {"coeff": 0.002}, "aux": {"cls": "NoAux", "params": {"key": "no-aux"}}}}
    headers = {
        "pre_schema": {"d_vit": 16, "exp_factor": 3, "sparsity_coeff": 4e-4, "ghost_grads": False, "seed": 3, "n_reinit_samples": 1024,
                       "remove_parallel_grads": True, "normalize_w_dec": True},
        "s1a_topk": {"schema": 1, "cls": "TopK", "cfg": {"d_model": 16, "exp_factor": 3, "seed": 1}},  # (a "top_k" entry here makes the reference's loader raise)
        "s1a_relu": {"schema": 1, "cls": "Relu", "cfg": {"d_model": 16, "d_sae": 48, "n_reinit_samples": 8}},
        "s1b_tree": {"schema": 1, "cls": "SparseAutoencoderConfig", "cfg": {"d_model": 16, "d_sae": 48, "activation": old_tree}},
        "s1b_nocls": {"schema": 1, "cfg": {"d_model": 16, "exp_factor": 3, "activation": tree(M.TopK(top_k=7))}},
        "s2_kind": {"schema": 2, "cfg": {"d_model": 16, "d_sae": 48, "activation": old_tree, "reinit_blend": 0.5}},
        "s3_l1": {"schema": 3, "cfg": {"d_model": 16, "d_sae": 48, "activation": old_relu, "seed": 7}},
        "s4_current_tree": {"schema": 4, "cfg": {"d_model": 16, "d_sae": 48, "normalize_w_dec": False,
                                                  "activation": tree(M.TopK(top_k=4, aux=M.AuxK(k_aux=11, alpha=0.5)))}},
    }
    sae = make_sae(ref, 16, 48, 4, seed=71)
    blob = io.BytesIO()
    torch.save(sae.state_dict(), blob)
    out = {"names": np.array(list(headers)), **{"sd_" + k: v for k, v in sae.state_dict().items()}}
    tmp = pathlib.Path("/tmp/_gold_legacy.pt")
    for name, hdr in headers.items():
        tmp.write_bytes(json.dumps(hdr).encode() + b"\n" + blob.getvalue())
        got = M.load(tmp)
        cfg = dataclasses.asdict(got.cfg)
        cfg["activation"] = M._serialize_dataclass(got.cfg.activation)
        assert all(torch.equal(v, sae.state_dict()[k]) for k, v in got.state_dict().items())
        out["hdr_" + name] = np.frombuffer(json.dumps(hdr).encode(), dtype=np.uint8)
        out["cfg_" + name] = np.frombuffer(json.dumps(cfg, sort_keys=True).encode(), dtype=np.uint8)
    tmp.unlink()
    npz("g16_legacy_checkpoints", **out)


def g14_inference(ref, tag, with_labels):
    """Runs the reference's inference pass (its own OrderedDataLoader, manager process included) on a cache written
    by this repo's protocol-2.1 writer and a checkpoint written by the reference's nn.dump."""
    import importlib
    import shutil
    import tempfile

    import scipy.sparse

    sys.path.insert(0, str(HERE.parent))
    from saev_amd.data import shards as my_shards

    ordered = importlib.import_module("saev.data.ordered")
    ref.data.OrderedConfig, ref.data.OrderedDataLoader = ordered.Config, ordered.DataLoader
    inf = importlib.import_module("saev.framework.inference")
    rshards = importlib.import_module("saev.data.shards")

    d, s, k, n_ex, n_tok, layers = 32, 256, 8, 13, 6, (5, 11)
    rows = lowrank_data(n_ex * len(layers) * (n_tok + 1), d, seed=140 + with_labels)
    acts = rows.reshape(n_ex, len(layers), n_tok + 1, d).numpy()
    labels = None
    if with_labels:
        labels = np.random.default_rng(14).integers(0, 4, (n_ex, n_tok)).astype(np.uint8)
    tmp = pathlib.Path(tempfile.mkdtemp(prefix="g14_"))
    try:
        shards_dir = my_shards.write_shards(tmp, acts, layers=layers, cls_token=True,
                                            max_tokens_per_shard=4 * (n_tok + 1) * len(layers), labels=labels)
        # the reference parses what this repo wrote, and agrees on the content hash
        md = rshards.Metadata.load(shards_dir)
        my_md = my_shards.Metadata.load(shards_dir)
        imap = rshards.IndexMap(md, "content", 11)
        probes = [0, 1, n_tok - 1, n_tok, 4 * n_tok - 1, 4 * n_tok, len(imap) - 1]
        index_rows = []
        for g in probes:
            ix = imap.from_global(g)
            index_rows.append([g, ix.example_idx, ix.content_token_idx, ix.shard_idx, ix.example_idx_in_shard,
                               ix.layer_idx_in_shard, ix.token_idx_in_shard])
        sae = make_sae(ref, d, s, k, k_aux=16, seed=141)
        with torch.no_grad():
            sae.b_enc.copy_(0.05 * torch.randn(s, generator=torch.Generator().manual_seed(142)))
            sae.b_dec.copy_(rows.mean(dim=0))
        run = ref_disk_new(tmp, shards_dir)
        ref.modeling.dump(run / "checkpoint" / "sae.pt", sae)
        cfg = inf.Config(run=run, data=ordered.Config(shards=shards_dir, layer=11, batch_size=4 * n_tok + 1),
                         n_dists=5, ignore_labels=[2] if with_labels else [], device="cpu")
        inf.worker_fn(cfg)
        out = run / "inference" / md.hash
        csr = scipy.sparse.load_npz(out / "token_acts.npz")
        metrics = json.loads((out / "metrics.json").read_text())
        npz(f"g14_inference_{tag}", acts=acts, labels=labels if labels is not None else np.zeros((0, 0), np.uint8),
            layers=np.array(layers), k=k, k_aux=16, n_dists=5, batch_size=4 * n_tok + 1,
            max_tokens_per_shard=4 * (n_tok + 1) * len(layers),
            ignore_labels=np.array([2] if with_labels else [], dtype=np.int64),
            ref_hash=np.frombuffer(md.hash.encode(), dtype=np.uint8), my_hash=np.frombuffer(my_md.hash.encode(), dtype=np.uint8),
            index_probes=np.array(index_rows, dtype=np.int64),
            csr_data=csr.data, csr_indices=csr.indices, csr_indptr=csr.indptr, csr_shape=np.array(csr.shape),
            mean_values=torch.load(out / "mean_values.pt"), sparsity=torch.load(out / "sparsity.pt"),
            distributions=torch.load(out / "distributions.pt"),
            metrics_keys=np.array(list(metrics.keys())), metrics_vals=np.array([float(v) for v in metrics.values()]),
            **params_of(sae))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def ref_disk_new(tmp, shards_dir):
    import saev.disk as rdisk

    runs_root = tmp / "saev" / "runs"
    runs_root.mkdir(parents=True)
    return rdisk.Run.new("gold0014", train_shards_dir=shards_dir, val_shards_dir=shards_dir, runs_root=runs_root).run_dir


def main():
    ref = _refshim.install()
    if "--only-g10" in sys.argv:
        g10_make_saes(ref)
        return
    if "--only-g4" in sys.argv:
        g4_auxk(ref)
        return
    if "--only-g16" in sys.argv:
        g16_legacy_checkpoints(ref)
        g17_batch_entropy(ref)
        return
    if "--only-g14" in sys.argv:
        g14_inference(ref, "plain", False)
        g14_inference(ref, "labels", True)
        return
    torch.set_num_threads(8)
    if "--only-r2" in sys.argv:  # fixtures added in round 2 (the others regenerate bit-identically; skip them)
        g9_train(ref, "c", d=128, s=1024, k=16, bsz=256, n_rows=2048, n_train=6144, thr=512, k_aux=32, lr=2e-3, n_warm=4,
                 grad_clip=0.02)
        g15_sample_prefixes(ref)
        return
    g1_g2_g3(ref)
    g4_auxk(ref)
    g5_objective(ref)
    g6_g7_g8(ref)
    g9_train(ref, "a", d=64, s=512, k=8, bsz=128, n_rows=1024, n_train=2048, thr=10_000_000, k_aux=512, lr=4e-4, n_warm=5)
    g9_train(ref, "b", d=128, s=1024, k=16, bsz=256, n_rows=2048, n_train=6144, thr=512, k_aux=32, lr=2e-3, n_warm=4)
    g9_train(ref, "c", d=128, s=1024, k=16, bsz=256, n_rows=2048, n_train=6144, thr=512, k_aux=32, lr=2e-3, n_warm=4,
             grad_clip=0.02)
    g10_make_saes(ref)
    g15_sample_prefixes(ref)
    g11_schedule(ref)
    g12_checkpoint(ref)
    g16_legacy_checkpoints(ref)
    g17_batch_entropy(ref)
    g13_matryoshka(ref)
    g14_inference(ref, "plain", False)
    g14_inference(ref, "labels", True)


if __name__ == "__main__":
    main()
