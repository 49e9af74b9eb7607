"""The reference's module-level API and known-answer tests, replayed through the HIP path
(needs an MI355X: -m gpu).  Reference tests restated: tests/test_nn_activations.py:29-94,
tests/test_auxk.py (scenarios reachable through the objective), tests/test_nn_modeling.py:60-67,
:99-175, :323-338, tests/test_nn_objectives.py:102-146."""

import dataclasses
import json
import math

import numpy as np
import pytest
import torch

import sae_ref as R
from conftest import load_golden

pytestmark = pytest.mark.gpu


def M():
    from saev_amd.nn import modeling

    return modeling


def O():
    from saev_amd.nn import objectives

    return objectives


def identity_sae(d, top_k, k_aux=2, alpha=1.0):
    m = M()
    cfg = m.SparseAutoencoderConfig(d_model=d, d_sae=d, normalize_w_dec=False, remove_parallel_grads=False,
                                    activation=m.TopK(top_k=top_k, aux=m.AuxK(k_aux=k_aux, alpha=alpha)))
    sae = m.SparseAutoencoder(cfg)
    with torch.no_grad():
        sae.W_dec.copy_(torch.eye(d)); sae.W_enc.copy_(torch.eye(d)); sae.b_dec.zero_(); sae.b_enc.zero_()
    return sae.cuda()


def topk_act(k, width=4):
    sae = identity_sae(width, k)
    return sae.activation


# ---- tests/test_nn_activations.py -------------------------------------------------------------


def test_topk_basic_forward():
    y = topk_act(2)(torch.tensor([[5.0, 1.0, 3.0, 2.0], [2.0, 4.0, 1.0, 3.0]]).cuda()).cpu()
    torch.testing.assert_close(y, torch.tensor([[5.0, 0.0, 3.0, 0.0], [0.0, 4.0, 0.0, 3.0]]))


def test_topk_ties():
    y = topk_act(2)(torch.full((1, 4), 2.0).cuda()).cpu()
    assert (y != 0).sum() == 2 and y[y != 0].unique().item() == 2.0


def test_topk_k_equals_size():
    x = torch.tensor([[5.0, 1.0, 3.0, 2.0]])
    torch.testing.assert_close(topk_act(4)(x.cuda()).cpu(), x)


def test_topk_negative_values():
    y = topk_act(2)(torch.tensor([[-5.0, -1.0, -3.0, -2.0]]).cuda()).cpu()
    torch.testing.assert_close(y, torch.tensor([[0.0, -1.0, 0.0, -2.0]]))


# ---- module surface -----------------------------------------------------------------------------


def test_encode_decode_forward_shapes_and_values():
    sae = identity_sae(4, 2)
    x = torch.tensor([[1.0, 2.0, 3.0, 4.0]]).cuda()
    enc = sae.encode(x)
    assert enc.h_x.cpu().tolist() == [[1.0, 2.0, 3.0, 4.0]] and enc.f_x.cpu().tolist() == [[0.0, 0.0, 3.0, 4.0]]
    assert sae.decode(torch.ones(2, 4).cuda()).shape == (2, 1, 4)
    out = sae(x)
    assert out.x_hats.shape == (1, 1, 4) and out.f_x.shape == (1, 4) and out.h_x.shape == (1, 4)
    assert out.x_hats.cpu().tolist() == [[[0.0, 0.0, 3.0, 4.0]]]
    idx, val = sae.encode_sparse(x)
    assert idx.cpu().tolist() == [[2, 3]] and val.cpu().tolist() == [[3.0, 4.0]]


def test_decode_prefixes_match_oracle():
    g = load_golden("g2_decode")
    m = M()
    sae = m.SparseAutoencoder(m.SparseAutoencoderConfig(d_model=48, d_sae=320, activation=m.TopK(top_k=8)))
    with torch.no_grad():
        sae.W_dec.copy_(g["W_dec"]); sae.b_dec.copy_(g["b_dec"])
    sae = sae.cuda()
    out = sae.decode(g["f"].cuda(), prefixes=g["prefixes"]).cpu()
    torch.testing.assert_close(out, g["x_hats_p3"], rtol=1e-5, atol=1e-5)
    with pytest.raises(AssertionError):
        sae.decode(g["f"].cuda(), prefixes=torch.tensor([100, 7, 320]))


def test_remove_parallel_grads_orthogonal_for_unnormalised_rows():
    m = M()
    sae = m.SparseAutoencoder(m.SparseAutoencoderConfig(d_model=4, d_sae=4, normalize_w_dec=False, remove_parallel_grads=True)).cuda()
    with torch.no_grad():
        sae.W_dec.copy_(torch.randn(4, 4))
    sae.W_dec.grad = torch.randn(4, 4).cuda()
    sae.remove_parallel_grads()
    dots = (sae.W_dec.grad * sae.W_dec).sum(dim=1).cpu()
    assert torch.allclose(dots, torch.zeros(4), atol=1e-6)


def test_dump_load_roundtrip_from_gpu_module(tmp_path):
    m = M()
    sae = m.SparseAutoencoder(m.SparseAutoencoderConfig(d_model=64, d_sae=256, activation=m.TopK(top_k=8))).cuda()
    x = torch.randn(32, 64).cuda()
    before = sae(x).x_hats.cpu()
    m.dump(tmp_path / "sae.pt", sae)
    back = m.load(tmp_path / "sae.pt", device="cuda")
    torch.testing.assert_close(back(x).x_hats.cpu(), before, rtol=0, atol=0)
    w_dec = back.W_dec.detach().clone()
    with torch.no_grad():
        back.W_enc.mul_(2.0)
    assert torch.equal(back.W_dec, w_dec), "W_enc and W_dec do not share storage"


def test_cpu_tensors_are_rejected_loudly():
    sae = identity_sae(4, 2)
    with pytest.raises(Exception, match="float32 on cuda|cuda"):
        sae.encode(torch.zeros(1, 4))


# ---- objective / AuxK ---------------------------------------------------------------------------


def objective_for(sae, thr=10_000_000):
    o = O()
    return o.get_objective(o.Matryoshka(n_prefixes=1, dead_threshold_tokens=thr))


def test_objective_total_and_eval_mode():
    sae = identity_sae(4, 4)
    obj = objective_for(sae)
    loss, out = obj(sae, torch.randn(2, 4).cuda())
    assert torch.allclose(loss.loss.detach(), loss.mse + loss.sparsity.to(loss.mse.device) + loss.aux)
    assert set(loss.metrics()) == {"loss", "mse", "l0", "l1", "sparsity", "aux", "n_dead"}
    sae.eval(); obj.eval()
    loss, _ = obj(sae, torch.tensor([[1.0, 2.0, 3.0, 4.0]]).cuda())
    assert loss.aux.item() == 0.0 and int(loss.n_dead) == 0
    # reference default: 10 Matryoshka prefixes (needs d_sae >= n_prefixes)
    m = M()
    big = m.SparseAutoencoder(m.SparseAutoencoderConfig(d_model=32, d_sae=256, activation=m.TopK(top_k=8))).cuda().train()
    obj10 = O().get_objective(O().Matryoshka()).train()
    torch.manual_seed(1)
    loss10, out10 = obj10(big, torch.randn(16, 32).cuda())
    assert out10.x_hats.shape == (16, 10, 32) and out10.prefixes.tolist()[-1] == 256 and torch.isfinite(loss10.mse)
    loss10.loss.backward()
    assert big.W_enc.grad is not None and big.W_enc.grad.abs().sum() > 0


def test_n_dead_tracks_dead_latents():  # tests/test_auxk.py:304-353
    thr = 10
    sae = identity_sae(4, 2, k_aux=512, alpha=1 / 32).train()
    obj = objective_for(sae, thr).train()
    x = torch.tensor([[2.0, 1.0, 0.0, -1.0]] * 2).cuda()
    loss, _ = obj(sae, x)
    assert int(loss.n_dead) == 0
    for _ in range(thr // 2 + 1):
        loss, _ = obj(sae, x)
    assert int(loss.n_dead) == 2
    loss, _ = obj(sae, torch.tensor([[0.0, 1.0, 3.0, -1.0]] * 2).cuda())
    assert int(loss.n_dead) == 1
    assert obj.toks_since_active.tolist()[2] == 0


def test_auxk_uses_preacts_of_dead_latents():  # tests/test_auxk.py:198-238
    sae = identity_sae(4, 2, k_aux=2, alpha=1.0).train()
    obj = objective_for(sae, thr=100).train()
    obj.toks_since_active = torch.tensor([100, 100, 0, 0])
    loss, out = obj(sae, torch.tensor([[1.0, 2.0, 3.0, 4.0]]).cuda())
    assert out.f_x.cpu().tolist() == [[0.0, 0.0, 3.0, 4.0]]
    assert int(loss.n_dead) == 2 and abs(loss.aux.item()) < 1e-6  # dead pre-acts [1,2] rebuild the residual exactly
    # k_aux = 1: only latent 1 (h = 2) is used -> diff = [-1, 0, 0, 0] -> mean 1/4
    sae1 = identity_sae(4, 2, k_aux=1, alpha=0.5).train()
    obj1 = objective_for(sae1, thr=100).train()
    obj1.toks_since_active = torch.tensor([100, 100, 0, 0])
    loss1, _ = obj1(sae1, torch.tensor([[1.0, 2.0, 3.0, 4.0]]).cuda())
    assert math.isclose(loss1.aux.item(), 0.5 * 0.25, rel_tol=1e-6)


@pytest.mark.parametrize("tag", ["nodead", "dead", "dead_few"])
def test_backward_populates_param_grads_like_autograd(tag):
    g = load_golden(f"g5_objective_{tag}")
    m, o = M(), O()
    cfg = m.SparseAutoencoderConfig(d_model=64, d_sae=512, normalize_w_dec=False, remove_parallel_grads=False,
                                    activation=m.TopK(top_k=int(g["k"]), aux=m.AuxK(k_aux=int(g["k_aux"]), alpha=float(g["alpha"]))))
    sae = m.SparseAutoencoder(cfg)
    sae.load_state_dict({k: g["p_" + k] for k in R.PARAM_ORDER})
    sae = sae.cuda().train()
    obj = o.get_objective(o.Matryoshka(n_prefixes=1, dead_threshold_tokens=int(g["thr"]))).train()
    obj.toks_since_active = g["toks_before"]
    loss, out = obj(sae, g["x"].cuda())
    loss.loss.backward()
    assert math.isclose(loss.mse.item(), g["mse"], rel_tol=1e-4) and int(loss.n_dead) == g["n_dead"]
    for k in R.PARAM_ORDER:
        torch.testing.assert_close(getattr(sae, k).grad.cpu(), g["g_" + k], rtol=1e-3, atol=1e-7, msg=lambda s: f"{k}: {s}")
    torch.testing.assert_close(out.f_x.cpu(), g["f"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out.h_x.cpu(), g["h"], rtol=1e-5, atol=1e-5)


# ---- train() / evaluate() ---------------------------------------------------------------------


def small_cfg(tmp_path, g, **kw):
    from saev_amd import data
    from saev_amd.framework import train as T

    m, o = M(), O()
    d, s, k, bsz = int(g["d"]), int(g["s"]), int(g["k"]), int(g["bsz"])
    dc = data.ShuffledConfig(batch_size=bsz, seed=3)
    return T.Config(
        train_data=dc, val_data=dc, n_train=int(g["n_train"]), n_val=10**9,
        sae=m.SparseAutoencoderConfig(d_model=d, d_sae=s, reinit_blend=0.0,
                                      activation=m.TopK(top_k=k, aux=m.AuxK(k_aux=int(g["k_aux"]), alpha=1 / 32))),
        objective=o.Matryoshka(n_prefixes=1, dead_threshold_tokens=int(g["thr"])),
        lr=float(g["lr"]), n_lr_warmup=int(g["n_warm"]), track=False, log_every=5, runs_root=tmp_path / "runs", **kw)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_evaluate_matches_reference_metrics(tmp_path, tag):
    """evaluate() sums are batch-order independent: load the reference's trained parameters and compare
    with the metrics the reference's evaluate() produced (golden G9)."""
    from saev_amd.framework import train as T

    g = load_golden(f"g9_train_{tag}")
    cfg = small_cfg(tmp_path, g)
    sae = M().SparseAutoencoder(cfg.sae)
    sae.load_state_dict({k: g["final_" + k] for k in R.PARAM_ORDER})
    saes = torch.nn.ModuleList([sae]).cuda()
    objs = torch.nn.ModuleList([O().get_objective(cfg.objective)])
    ev = T.evaluate([cfg], saes, objs, val_pool=g["val"])[0]
    flip = 4.0 / (g["val"].shape[0] * int(g["k"]))
    for key in ("l1", "mse", "normalized_mse", "sse_sae"):
        assert math.isclose(getattr(ev, key), float(g["ev_" + key]), rel_tol=max(1e-4, flip)), key
    assert math.isclose(ev.l0, float(g["ev_l0"]), rel_tol=1e-6)
    assert math.isclose(ev.sse_baseline, float(g["ev_sse_baseline"]), rel_tol=1e-9)
    assert abs(ev.n_dead - int(g["ev_n_dead"])) <= 1 and ev.n_dense == int(g["ev_n_dense"])
    assert (ev.freqs - g["ev_freqs"]).abs().max() < 2.0 / g["val"].shape[0]


def test_train_end_to_end_matches_oracle_on_same_batches(tmp_path):
    from saev_amd import data
    from saev_amd.framework import train as T
    from saev_amd.utils import scheduling

    g = load_golden("g9_train_a")
    cfg = small_cfg(tmp_path, g)
    ids = T.worker_fn([cfg], train_pool=g["acts"], val_pool=g["val"])
    run_dir = tmp_path / "runs" / ids[0]
    ckpt = M().load(run_dir / "checkpoint" / "sae.pt")
    assert json.loads((run_dir / "checkpoint" / "config.json").read_text())["lr"] == cfg.lr
    logs = [json.loads(line) for line in (run_dir / "metrics.jsonl").read_text().splitlines()]
    assert logs[0]["step"] == 4 and "metrics/normalized_mse" in logs[0] and "eval/normalized_mse" in logs[-1]
    # the same batches through the oracle
    torch.manual_seed(cfg.seed)
    init = M().SparseAutoencoder(cfg.sae)
    rcfg = R.RefConfig(d_model=int(g["d"]), d_sae=int(g["s"]), top_k=int(g["k"]), k_aux=int(g["k_aux"]),
                       dead_threshold_tokens=int(g["thr"]), lr=cfg.lr, n_lr_warmup=cfg.n_lr_warmup)
    state = R.TrainState.create({k: getattr(init, k).detach() for k in R.PARAM_ORDER})
    dl = data.ShuffledDataLoader(cfg.train_data, device="cpu", pool=g["acts"])
    lim = scheduling.BatchLimiter(dl, cfg.n_train)
    sched = R.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, len(lim), 0.0)
    mses = []
    for batch in lim:
        mses.append(R.train_step(state, batch["act"], rcfg, sched)["mse"])
    for k in R.PARAM_ORDER:
        torch.testing.assert_close(getattr(ckpt, k).detach(), state.params[k], rtol=2e-3, atol=5e-5, msg=lambda s: f"{k}: {s}")
    assert math.isclose(logs[-2]["loss/mse"], mses[logs[-2]["step"]], rel_tol=1e-3)
    assert mses[-1] < mses[0]


def test_train_is_deterministic(tmp_path):
    from saev_amd.framework import train as T

    g = load_golden("g9_train_b")
    outs = []
    for run in range(2):
        saes, objs, log, steps = T.train([small_cfg(tmp_path, g)], train_pool=g["acts"])
        outs.append({k: v.detach().cpu().clone() for k, v in saes[0].state_dict().items()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), f"{k} differs between identical runs"


def test_train_matryoshka_matches_oracle_on_same_batches_and_prefix_draws(tmp_path):
    """Default-style objective (several Matryoshka prefixes + AuxK): train() vs the oracle fed the same batches;
    both draw the prefix cut points from torch's global CPU RNG, which is seeded identically."""
    from saev_amd import data
    from saev_amd.framework import train as T
    from saev_amd.utils import scheduling

    g = load_golden("g9_train_b")
    cfg = small_cfg(tmp_path, g)
    cfg = dataclasses.replace(cfg, objective=dataclasses.replace(cfg.objective, n_prefixes=5), n_train=3072)
    saes, objs, log, steps = T.train([cfg], train_pool=g["acts"])
    torch.manual_seed(cfg.seed)
    init = M().SparseAutoencoder(cfg.sae)
    rcfg = R.RefConfig(d_model=int(g["d"]), d_sae=int(g["s"]), top_k=int(g["k"]), k_aux=int(g["k_aux"]),
                       dead_threshold_tokens=int(g["thr"]), lr=cfg.lr, n_lr_warmup=cfg.n_lr_warmup, n_prefixes=5)
    state = R.TrainState.create({k: getattr(init, k).detach() for k in R.PARAM_ORDER})
    dl = data.ShuffledDataLoader(cfg.train_data, device="cpu", pool=g["acts"])
    lim = scheduling.BatchLimiter(dl, cfg.n_train)
    sched = R.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, len(lim), 0.0)
    recs = [R.train_step(state, batch["act"], rcfg, sched) for batch in lim]
    assert len(recs) == steps and max(r["n_dead"] for r in recs) > 0
    got = {k: v.detach().cpu() for k, v in saes[0].state_dict().items()}
    for k in R.PARAM_ORDER:
        bad = ~torch.isclose(got[k], state.params[k], rtol=2e-3, atol=5e-5)
        assert bad.float().mean() < 5e-3, f"{k}: {bad.sum().item()} of {bad.numel()} elements off"
    last = log.records[0][-1][1]
    assert math.isclose(last["loss/mse"], recs[log.records[0][-1][0]]["mse"], rel_tol=2e-3)


@pytest.mark.parametrize("budget_gb", ["1000", "0"])  # resident pool / forced streaming reservoir
def test_train_from_a_shard_directory(tmp_path, monkeypatch, budget_gb):
    """worker_fn fed from a protocol-2.1 cache on disk (not an in-memory pool): every token is consumed once per
    epoch in both feed modes, the loss falls, and the run directory links back to the cache."""
    from saev_amd import data, disk
    from saev_amd.framework import train as T

    monkeypatch.setenv("SAEV_AMD_RESIDENT_GB", budget_gb)
    g = load_golden("g9_train_a")
    d, bsz = int(g["d"]), int(g["bsz"])
    acts = g["acts"].numpy().reshape(-1, 1, 8, d)  # 128 examples x 8 content tokens, one layer, no CLS
    shards = data.write_shards(tmp_path, acts, layers=(11,), cls_token=False, max_tokens_per_shard=8 * 20)
    dc = data.ShuffledConfig(shards=shards, layer=11, batch_size=bsz, seed=3, buffer_size=4)
    cfg = dataclasses.replace(small_cfg(tmp_path, g), train_data=dc, val_data=dc, runs_root=tmp_path / "saev" / "runs")
    # the feed itself, on the device
    dl = data.ShuffledDataLoader(dc, device="cuda")
    assert (dl.reservoir is not None) == (budget_gb == "0")
    rows = torch.cat([torch.stack([b["example_idx"], b["token_idx"]], 1) for b in dl]).cpu()
    assert rows.shape[0] == acts.shape[0] * 8 and len({tuple(r) for r in rows.tolist()}) == rows.shape[0]
    b = next(iter(dl))
    torch.testing.assert_close(b["act"].cpu(), torch.from_numpy(acts[b["example_idx"].cpu(), 0, b["token_idx"].cpu()]),
                               rtol=0, atol=0)
    ids = T.worker_fn([cfg])
    run = disk.Run(tmp_path / "saev" / "runs" / ids[0])
    assert run.train_shards == shards.resolve() and run.ckpt.exists()
    logs = [json.loads(line) for line in (run.run_dir / "metrics.jsonl").read_text().splitlines()]
    mses = [rec["loss/mse"] for rec in logs if "loss/mse" in rec]
    assert len(mses) >= 3 and all(math.isfinite(m) for m in mses)
    # per-batch losses of a random feed are noisy; judge progress on the whole cache instead
    x = g["acts"].cuda()
    torch.manual_seed(cfg.seed)
    before = M().SparseAutoencoder(cfg.sae).cuda()
    after = M().load(run.ckpt, device="cuda")
    mse = [((m(x).x_hats[:, -1] - x) ** 2).mean().item() for m in (before, after)]
    assert mse[1] < mse[0], mse


def test_make_saes_on_the_device_matches_the_reference_default_init():
    """SURVEY 8 a18 on the device: the reference's DEFAULT initialisation (reinit_blend = 0.8, train.py:108-189) fed CUDA
    batches -- the centring, blending and normalising then run on the GPU -- gives the weights the reference produced from
    the same seed and batches (fixture G10) up to the rounding of a different summation order in the column mean."""
    from saev_amd.framework import train as T

    g = load_golden("g10_make_saes")
    acts, bsz = g["acts"].cuda(), int(g["bsz"])

    class Loader:
        n_samples = acts.shape[0]

        def __iter__(self):
            for lo in range(0, acts.shape[0], bsz):
                yield {"act": acts[lo : lo + bsz]}

    m, o = M(), O()
    d, s = acts.shape[1], g["W_dec_0"].shape[0]
    cfgs = [(m.SparseAutoencoderConfig(d_model=d, d_sae=s, reinit_blend=float(b), activation=m.TopK(top_k=4)),
             o.Matryoshka(n_prefixes=1)) for b in g["blends"].tolist()]
    assert cfgs[0][0].reinit_blend == m.SparseAutoencoderConfig(d_model=d, d_sae=s).reinit_blend == 0.8  # the default
    torch.manual_seed(int(g["seed"]))
    saes, _, _ = T.make_saes(cfgs, Loader(), device="cuda")
    for i, sae in enumerate(saes):
        torch.testing.assert_close(sae.W_enc.detach().cpu(), g[f"W_enc_{i}"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(sae.W_dec.detach().cpu(), g[f"W_dec_{i}"], rtol=1e-5, atol=1e-6)
        assert torch.equal(sae.W_enc.detach(), sae.W_dec.detach().T)


def test_train_on_the_device_with_the_default_datapoint_init(tmp_path, monkeypatch):
    """train() itself with the reference's default reinit_blend = 0.8 from a shard directory, on the device: the weights the
    run starts from are those make_saes derives -- pinned to the reference by G10 on CPU -- from the first max(d_sae, 65 536)
    rows the loader delivers, and the run trains from there (loss on the whole cache falls)."""
    from saev_amd import data
    from saev_amd.framework import train as T

    g = load_golden("g9_train_a")
    d, s, bsz = int(g["d"]), int(g["s"]), int(g["bsz"])
    acts = g["acts"].numpy().reshape(-1, 1, 8, d)
    shards = data.write_shards(tmp_path, acts, layers=(11,), cls_token=False, max_tokens_per_shard=8 * 20)
    dc = data.ShuffledConfig(shards=shards, layer=11, batch_size=bsz, seed=3, buffer_size=4)
    base = small_cfg(tmp_path, g)
    cfg = dataclasses.replace(base, train_data=dc, val_data=dc, sae=dataclasses.replace(base.sae, reinit_blend=0.8))
    assert g["acts"].shape[0] >= s  # enough rows for a datapoint init of s latents
    started = {}
    real = T.make_saes

    def spy(cfgs, dl, device="cuda"):
        out = real(cfgs, dl, device)
        started.update({k: v.detach().cpu().clone() for k, v in out[0][0].state_dict().items()})
        return out

    monkeypatch.setattr(T, "make_saes", spy)
    saes, objs, run, steps = T.train([cfg])
    assert steps >= 10 and set(started) == set(R.PARAM_ORDER)
    # what the same seed and the same delivered batches give on the CPU (test_host_cpu.py pins that arithmetic to G10)
    # (make_saes is handed the step limiter, whose n_samples is n_train -- as in the reference, train.py:260-262 -- so the
    # sample is min(max(d_sae, 65 536), n_train) rows, drawn over as many epochs of the loader as that takes)
    from saev_amd.utils import scheduling

    n_want = min(max(s, 65_536), cfg.n_train)
    batches = []
    for b in scheduling.BatchLimiter(data.ShuffledDataLoader(dc, device="cuda"), cfg.n_train):
        batches.append(b["act"].cpu())
        if sum(len(t) for t in batches) >= n_want:
            break

    class Loader:
        n_samples = cfg.n_train

        def __iter__(self):
            return iter({"act": b} for b in batches)

    torch.manual_seed(cfg.seed)
    want, _, _ = real([(cfg.sae, cfg.objective)], Loader(), device="cpu")
    for key in R.PARAM_ORDER:
        torch.testing.assert_close(started[key], want[0].state_dict()[key], rtol=1e-5, atol=1e-6, msg=lambda m: f"{key}: {m}")
    assert (started["W_dec"].norm(dim=1) - 1).abs().max() < 1e-5 and torch.equal(started["W_enc"], started["W_dec"].T)
    assert not torch.equal(started["W_dec"], torch.nn.init.kaiming_uniform_(torch.empty(s, d)))  # (data went into it)
    x = g["acts"].cuda()
    before = M().SparseAutoencoder(cfg.sae)
    before.load_state_dict(started)
    mse = [((mdl.cuda()(x).x_hats[:, -1] - x) ** 2).mean().item() for mdl in (before, saes[0])]
    assert mse[1] < mse[0], mse


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_log_block_matches_the_reference_values(tmp_path, tag):
    """The train loop's log block (reference train.py:365-442) on the reference's own batch order, step by step,
    against the values the reference logged (golden G9, log_every = 1): explained variance, normalised MSE, dead-unit
    share, dictionary coherence, decoder row norm, grad norm -- not just the key names."""
    from saev_amd.framework import train as T
    from saev_amd.framework.ddp import DataParallelStepper

    g = load_golden(f"g9_train_{tag}")
    bsz, k = int(g["bsz"]), int(g["k"])
    cfg = dataclasses.replace(small_cfg(tmp_path, g), grad_clip=float(g.get("grad_clip", 1.0)))
    sae = M().SparseAutoencoder(cfg.sae)
    sae.load_state_dict({key: g["init_" + key] for key in R.PARAM_ORDER})
    sae = sae.cuda().train()
    obj = O().get_objective(cfg.objective).train()
    st = DataParallelStepper(obj._bind(sae, bsz))
    sched = R.WarmupCosine(0.0, cfg.n_lr_warmup, cfg.lr, math.ceil(cfg.n_train / bsz), 0.0)
    lr, seen, logs = 0.0, 0, []
    for x in R.limited_batches([b.cuda() for b in g["acts"].split(bsz)], cfg.n_train, bsz, drop_last=False):
        seen += len(x)
        pre = {}
        st.train_step(x, lr, cfg.grad_clip, pre_tail=lambda: pre.update(T._decoder_metrics(sae, cfg)))
        logs.append(T._log_metrics(sae, st.engine, x, lr, seen, cfg, pre))
        lr = sched.step()
    assert len(logs) == g["n_steps"]
    flip = 4.0 / (bsz * k)
    bands = {"loss/mse": max(1e-4, flip), "loss/l1": max(1e-4, flip), "loss/l0": 1e-6, "loss/loss": max(1e-4, flip),
             "metrics/grad_norm": 2e-3, "progress/learning_rate": 0.0, "metrics/normalized_mse": max(1e-4, flip),
             "metrics/sse_sae": max(1e-4, flip), "metrics/sse_baseline": 1e-9, "metrics/explained_variance": max(1e-4, flip),
             "metrics/avg_decoder_row_norm": 1e-6, "metrics/dictionary_coherence": 2e-3}
    for key, rtol in bands.items():
        want = g["log_" + key.replace("/", "_")].numpy()
        got = np.array([rec[key] for rec in logs], dtype=np.float64)
        np.testing.assert_allclose(got, want, rtol=rtol, atol=1e-12, err_msg=key)
    np.testing.assert_allclose([rec["loss/aux"] for rec in logs], g["log_loss_aux"].numpy(), rtol=max(1e-3, flip), atol=1e-8)
    dead = np.array([rec["metrics/dead_unit_pct"] for rec in logs])
    assert np.abs(dead - g["log_metrics_dead_unit_pct"].numpy()).max() <= 4.0 / int(g["s"])
    assert np.abs(np.array([rec["loss/n_dead"] for rec in logs]) - g["log_loss_n_dead"].numpy()).max() <= 1
    assert all(rec["loader/buffer_fill"] == 1.0 for rec in logs)  # resident feed


def test_streaming_feed_reports_its_fill(tmp_path, monkeypatch):
    """loader/buffer_fill is the reservoir's real fill in streaming mode (it was a constant)."""
    from saev_amd import data
    from saev_amd.framework import train as T

    monkeypatch.setenv("SAEV_AMD_RESIDENT_GB", "0")
    g = load_golden("g9_train_a")
    d, bsz = int(g["d"]), int(g["bsz"])
    acts = g["acts"].numpy().reshape(-1, 1, 8, d)
    shards = data.write_shards(tmp_path, acts, layers=(11,), cls_token=False, max_tokens_per_shard=8 * 20)
    dc = data.ShuffledConfig(shards=shards, layer=11, batch_size=bsz, seed=3, buffer_size=4)
    cfg = dataclasses.replace(small_cfg(tmp_path, g), train_data=dc, val_data=dc, log_every=2)
    saes, objs, run, steps = T.train([cfg])
    fills = [m["loader/buffer_fill"] for _, m in run.records[0]]
    assert fills and all(0.0 <= f <= 1.0 for f in fills) and any(f < 1.0 for f in fills)
    # the loader-coverage figures of the reference's log block (train.py:369-377) ride along: rows of a shuffled batch are
    # spread over many examples and over all 8 token positions
    for _, m in run.records[0]:
        assert 0.0 < m["loader/example_entropy_normalized"] <= 1.0 and 0.0 < m["loader/example_coverage"] <= 1.0
        assert m["loader/token_coverage"] == 1.0 and m["loader/token_entropy"] > 0.9 * math.log(8)


def test_group_of_saes_shares_the_batch_work_and_matches_single_runs(tmp_path):
    """Several SAEs trained on the same batches (reference train.py:334-348; grouping :669-695): the group builds the
    x statistics / centring / operand images once (saev_share_x) and every member ends with bit-for-bit the parameters
    it gets when trained alone on the same batches from the same start -- different top_k, learning rates and AuxK sizes
    included (AuxK is active in this fixture)."""
    from saev_amd.engine import EngineConfig, SaeEngine
    from saev_amd.framework import train as T

    g = load_golden("g9_train_b")
    d, s, bsz, thr = int(g["d"]), int(g["s"]), int(g["bsz"]), int(g["thr"])
    variants = [(16, 2e-3, 32), (8, 1e-3, 64), (24, 4e-3, 16)]  # (top_k, lr, k_aux)
    gen = torch.Generator().manual_seed(5)
    starts = []
    for _ in variants:
        p = {key: g["init_" + key].clone() for key in R.PARAM_ORDER}
        p["W_enc"] = p["W_enc"] + 0.01 * torch.randn(p["W_enc"].shape, generator=gen)
        starts.append(p)
    batches = [b.cuda() for b in g["acts"].split(bsz)] * 2

    def make(i):
        k, _, k_aux = variants[i]
        e = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, dead_threshold_tokens=thr, max_batch=bsz))
        e.load_params(starts[i])
        return e

    group = [make(i) for i in range(3)]
    for e in group[1:]:
        e.share_x(group[0])
    dead_seen = 0
    for t, x in enumerate(batches):
        for i, e in enumerate(group):
            e.train_step(x, 0.0 if t == 0 else variants[i][1], 1.0)
        dead_seen = max(dead_seen, group[0].read_stats().n_dead)
    assert dead_seen > 0, "the fixture is meant to exercise AuxK"
    for i in range(3):
        alone = make(i)
        for t, x in enumerate(batches):
            alone.train_step(x, 0.0 if t == 0 else variants[i][1], 1.0)
        assert torch.equal(alone.params, group[i].params), f"SAE {i}: group run differs from the single run"
        assert torch.equal(alone.adam_v, group[i].adam_v) and torch.equal(alone.toks_since_active, group[i].toks_since_active)

    # train() links the members of a parallel group the same way
    base = small_cfg(tmp_path, g)
    m = M()
    cfgs = [dataclasses.replace(base, lr=lr, sae=dataclasses.replace(base.sae, activation=m.TopK(top_k=k, aux=m.AuxK(k_aux=ka, alpha=1 / 32))))
            for k, lr, ka in variants]
    assert len(T.split_cfgs(cfgs)) == 1
    saes, objs, run, steps = T.train(cfgs, train_pool=g["acts"])
    engines = [o.__dict__["_eng_ref"] for o in objs]
    assert engines[1]._leader is engines[0] and engines[2]._leader is engines[0]
    assert len(run.records) == 3 and all(len(r) == len(run.records[0]) > 0 for r in run.records)


def test_shared_x_is_only_borrowed_for_the_same_batch():
    """saev_share_x never changes results: a follower borrows only what the leader built for the very same batch tensor,
    once; any other call order makes it build its own."""
    from saev_amd.engine import EngineConfig, SaeEngine

    d, s, n = 64, 512, 128
    gen = torch.Generator().manual_seed(3)
    p = R.init_params(R.RefConfig(d_model=d, d_sae=s), gen)
    xa, xb = torch.randn(n, d, generator=gen).cuda(), (3.0 * torch.randn(n, d, generator=gen) + 1.0).cuda()
    lead = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=8, max_batch=n))
    foll = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=4, max_batch=n))
    solo = SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=4, max_batch=n))
    for e in (lead, foll, solo):
        e.load_params(p)
    foll.share_x(lead)

    def codes(e, x):
        e.step_forward(x, training=False)
        i, v, xh = e.last_codes(n)
        return i.clone(), v.clone(), xh.clone(), e.read_stats().mse

    def same(a, b):
        return all(torch.equal(u, w) for u, w in zip(a[:3], b[:3])) and a[3] == b[3]

    lead.step_forward(xa, training=False)
    assert same(codes(foll, xa), codes(solo, xa))          # borrowed
    assert same(codes(foll, xb), codes(solo, xb))          # other tensor: built its own
    assert same(codes(foll, xa), codes(solo, xa))          # leader has not rebuilt since the last borrow: own again
    buf = xa.clone()
    lead.step_forward(buf, training=False)
    buf.copy_(xb)                                          # same address, new contents, leader not re-run ...
    lead.step_forward(buf, training=False)                 # ... then re-run: the follower may borrow again
    assert same(codes(foll, buf), codes(solo, xb))
    foll.share_x(None)
    assert same(codes(foll, xa), codes(solo, xa))


def test_extraction_hand_off_equals_the_cache_round_trip_and_trains(tmp_path):
    """BASELINE configs[4] on one GPU: a (small, randomly initialised) vision transformer's forward hooks fill the HBM
    reservoir and train() draws from it.  (1) Every (example, token) arrives exactly once per epoch and carries bit for
    bit the row a protocol-2.1 cache holds when the same recorded blocks are written to disk and read back through the
    shuffled loader (the reference's route: shards.py:697-850 -> shuffled.py); (2) train() runs from the feed."""
    from saev_amd import data
    from saev_amd.data.vit import VisionTransformer
    from saev_amd.framework import train as T

    torch.manual_seed(0)
    d, n_img, patches = 64, 96, 16
    vit = VisionTransformer(d_model=d, depth=4, heads=4, patch=4, image=16).cuda().eval()
    rec = data.ActivationRecorder(vit, vit.blocks, layers=(1, 3), content_tokens_per_example=patches, cls_token=True)
    imgs = torch.randn(n_img, 3, 16, 16)

    def images():
        for lo in range(0, n_img, 8):
            yield imgs[lo : lo + 8], torch.arange(lo, min(lo + 8, n_img))

    # the disk route: record every block, write a cache, read it back
    with torch.no_grad():
        blocks = [rec(b.cuda())[1].cpu().clone() for b, _ in images()]
    acts = torch.cat(blocks).numpy()  # (n_img, 2, 17, d)
    shards = data.write_shards(tmp_path, acts, layers=(1, 3), cls_token=True, max_tokens_per_shard=17 * 2 * 20)
    dl = data.ShuffledDataLoader(data.ShuffledConfig(shards=shards, layer=3, batch_size=256, seed=3), device="cuda")
    disk = {(e, t): a.clone() for b in dl for a, e, t in zip(b["act"].cpu(), b["example_idx"].tolist(), b["token_idx"].tolist())}
    assert len(disk) == n_img * patches
    # the direct route
    cfg_x = data.ExtractConfig(layer=3, batch_size=256, buffer_size=3, seed=5)
    feed = data.ExtractionFeed(cfg_x, rec, images, n_examples=n_img, d_model=d, device="cuda")
    direct = {}
    for b in feed:
        assert b["act"].is_cuda
        for a, e, t in zip(b["act"].cpu(), b["example_idx"].tolist(), b["token_idx"].tolist()):
            assert (e, t) not in direct
            direct[(e, t)] = a
    assert direct.keys() == disk.keys()
    assert all(torch.equal(direct[key], disk[key]) for key in disk)
    # train from the hooks: two epochs' worth of rows
    m, o = M(), O()
    cfg = T.Config(n_train=2 * n_img * patches, sae=m.SparseAutoencoderConfig(d_model=d, d_sae=512, reinit_blend=0.0,
                                                                              activation=m.TopK(top_k=8, aux=m.AuxK(k_aux=32))),
                   objective=o.Matryoshka(n_prefixes=1), lr=2e-3, n_lr_warmup=2, log_every=3, track=False, runs_root=tmp_path / "runs",
                   train_data=data.ShuffledConfig(batch_size=256), val_data=data.ShuffledConfig(batch_size=256))
    saes, objs, run, steps = T.train([cfg], train_feed=feed)
    # (two epochs of 6 full batches each leave the limiter one nominal batch short twice -- the reference's end-of-epoch
    # correction, utils/scheduling.py:109-122 -- so it takes 14 steps, not ceil(n_train / batch) = 12)
    assert steps == 14
    mses = [rec_["loss/mse"] for _, rec_ in run.records[0]]
    fills = [rec_["loader/buffer_fill"] for _, rec_ in run.records[0]]
    assert len(mses) == 4 and all(math.isfinite(v) for v in mses) and mses[-1] < mses[0]
    assert all(0.0 <= f <= 1.0 for f in fills)


def test_a_destroyed_leader_leaves_no_dangling_link():
    """a follower kept a raw pointer to its leader's context; rebuilding or closing the leader (a larger
    batch arrives, the module moves) then made the follower's next forward read freed memory.  saev_destroy now clears the
    back-links of its followers (and a follower's own destruction takes it off its leader's list): the follower simply builds
    its own x-derived buffers again."""
    from saev_amd.engine import EngineConfig, SaeEngine

    d, s, n = 64, 512, 128
    gen = torch.Generator().manual_seed(5)
    p = R.init_params(R.RefConfig(d_model=d, d_sae=s), gen)
    x = torch.randn(n, d, generator=gen).cuda()
    lead, foll, foll2, solo = (SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k_, max_batch=n)) for k_ in (8, 4, 4, 4))
    for e in (lead, foll, foll2, solo):
        e.load_params(p)
    foll.share_x(lead)
    foll2.share_x(lead)

    def codes(e):
        e.step_forward(x, training=False)
        i, v, xh = e.last_codes(n)
        return i.clone(), v.clone(), xh.clone()

    lead.step_forward(x, training=False)
    want = codes(solo)
    assert all(torch.equal(a, b) for a, b in zip(codes(foll), want))  # borrowed
    foll2.close()            # a follower goes first: the leader must forget it ...
    lead.step_forward(x, training=False)
    lead.close()             # ... and the leader's destruction must not touch it, but must unlink the other one
    torch.cuda.synchronize()
    for _ in range(3):
        assert all(torch.equal(a, b) for a, b in zip(codes(foll), want))
    foll.share_x(None)
    assert all(torch.equal(a, b) for a, b in zip(codes(foll), want))


def test_engine_rebuild_for_a_sharded_tail_keeps_the_optimizer_state_per_tensor():
    """a module that already owns an engine when a data-parallel run asks for the padded layout of a sharded tail
    (shard_world > 1: other offsets, another length) must carry Adam's moments over tensor by tensor, not as flat copies."""
    m = M()
    sae = m.SparseAutoencoder(m.SparseAutoencoderConfig(d_model=48, d_sae=200, activation=m.TopK(top_k=4), reinit_blend=0.0)).cuda()
    eng = sae._eng(64)
    g = torch.Generator(device="cuda").manual_seed(0)
    eng.adam_m.copy_(torch.randn(eng.n_params, device="cuda", generator=g))
    eng.adam_v.copy_(torch.rand(eng.n_params, device="cuda", generator=g))
    eng.adam_steps = 7
    want = {n: (eng.view(n, eng.adam_m).clone(), eng.view(n, eng.adam_v).clone(), getattr(sae, n).detach().clone()) for n in eng.offsets}
    sae._shard_world = 8
    eng2 = sae._eng(64)
    assert eng2 is not eng and eng2.cfg.shard_world == 8 and eng2.n_params > eng.n_params and eng2.adam_steps == 7
    for n in eng2.offsets:
        assert torch.equal(eng2.view(n, eng2.adam_m), want[n][0]) and torch.equal(eng2.view(n, eng2.adam_v), want[n][1])
        assert torch.equal(getattr(sae, n).detach(), want[n][2])
    pad = torch.ones(eng2.n_params, dtype=torch.bool, device="cuda")
    for n in eng2.offsets:
        pad[eng2.offsets[n] : eng2.offsets[n] + eng2.view(n).numel()] = False
    assert (eng2.adam_m[pad] == 0).all() and (eng2.adam_v[pad] == 0).all()


def test_engine_config_names_are_validated():
    from saev_amd.engine import EngineConfig, SaeEngine

    with pytest.raises(ValueError, match="guaranteed.*predicted"):
        SaeEngine(EngineConfig(d_model=32, d_sae=64, top_k=4, bounds="predcited"))
    with pytest.raises(ValueError, match="f16r"):
        SaeEngine(EngineConfig(d_model=32, d_sae=64, top_k=4, encoder="fp16"))
