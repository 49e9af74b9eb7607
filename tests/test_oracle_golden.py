"""Pin the CPU oracle (oracle/sae_ref.py) to vectors produced by RUNNING the reference
(oracle/gen_golden.py).  CPU only."""

import json
import math

import numpy as np
import pytest
import torch

import sae_ref as R
from conftest import load_golden

TIGHT = dict(rtol=1e-6, atol=1e-7)


def params(g, prefix="p_"):
    return {k: g[prefix + k] for k in R.PARAM_ORDER}


def test_g1_encode_topk():
    g = load_golden("g1_encode_topk")
    h = R.encode_pre(g["x"], g["p_W_enc"], g["p_b_enc"])
    torch.testing.assert_close(h, g["h"], **TIGHT)
    f = R.topk_activation(h, 8)
    torch.testing.assert_close(f, g["f"], **TIGHT)
    assert ((f != 0).sum(1) == g["n_sel"]).all()


def test_g1_k_larger_than_d_sae_keeps_everything():
    g = load_golden("g1_encode_topk_kfull")
    h = R.encode_pre(g["x"], g["p_W_enc"], g["p_b_enc"])
    f = R.topk_activation(h, 64)
    torch.testing.assert_close(f, g["f"], **TIGHT)
    assert (f < 0).any(), "negative pre-activations are kept (no ReLU)"


def test_g2_decode_single_and_matryoshka():
    g = load_golden("g2_decode")
    torch.testing.assert_close(R.decode(g["f"], g["W_dec"], g["b_dec"]), g["x_hats_p1"], **TIGHT)
    out = R.decode(g["f"], g["W_dec"], g["b_dec"], g["prefixes"])
    assert out.shape == g["x_hats_p3"].shape
    torch.testing.assert_close(out, g["x_hats_p3"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out[:, -1], g["x_hats_p1"][:, 0], rtol=1e-5, atol=1e-6)


def test_g3_mse():
    g = load_golden("g3_mse")
    torch.testing.assert_close(R.mean_squared_err(g["x_hat"], g["x"]), g["mse"], **TIGHT)
    big = R.mean_squared_err(g["big_x_hat"], g["big_x"])
    assert torch.isfinite(big).all()
    torch.testing.assert_close(big, g["big_mse"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("tag", ["lt", "eq", "gt"])
def test_g4_auxk_value_and_grads(tag):
    g = load_golden(f"g4_auxk_{tag}")
    h = g["h"].clone().requires_grad_(True)
    W = g["W_dec"].clone().requires_grad_(True)
    b = g["b_dec"].clone().requires_grad_(True)
    loss = R.auxk_loss(x=g["x"], h=h, x_hat_last=g["x_hat"], dead_mask=g["dead"], W_dec=W, b_dec=b,
                       k_aux=int(g["k_aux"]), alpha=float(g["alpha"]))
    loss.backward()
    torch.testing.assert_close(loss.detach(), torch.as_tensor(g["loss"]), **TIGHT)
    torch.testing.assert_close(h.grad, g["g_h"], **TIGHT)
    torch.testing.assert_close(W.grad, g["g_W_dec"], **TIGHT)
    torch.testing.assert_close(b.grad, g["g_b_dec"], **TIGHT)
    assert (h.grad[:, ~g["dead"]] == 0).all()


@pytest.mark.parametrize("tag", ["nodead", "dead", "dead_few"])
def test_g5_objective_forward_backward(tag):
    g = load_golden(f"g5_objective_{tag}")
    cfg = R.RefConfig(d_model=64, d_sae=512, top_k=int(g["k"]), k_aux=int(g["k_aux"]), alpha=float(g["alpha"]),
                      dead_threshold_tokens=int(g["thr"]))
    leaves = {k: v.clone().requires_grad_(True) for k, v in params(g).items()}
    toks = g["toks_before"].clone()
    out = R.objective_forward(leaves, g["x"], cfg, toks_since_active=toks, training=True)
    out.loss.backward()
    assert torch.equal(toks, g["toks_after"])
    assert out.n_dead == g["n_dead"]
    for name, val in (("mse", out.mse), ("aux", out.aux), ("l0", out.l0), ("l1", out.l1)):
        torch.testing.assert_close(val.detach(), torch.as_tensor(g[name]), **TIGHT)
    torch.testing.assert_close(out.f.detach(), g["f"], **TIGHT)
    torch.testing.assert_close(out.x_hats[:, -1].detach(), g["x_hat"], **TIGHT)
    for k in R.PARAM_ORDER:
        torch.testing.assert_close(leaves[k].grad, g["g_" + k], rtol=1e-5, atol=1e-8)
    if tag != "nodead":
        assert out.n_dead > 0 and out.aux.item() > 0


def test_g5_objective_eval_mode():
    g = load_golden("g5_objective_eval")
    cfg = R.RefConfig(d_model=64, d_sae=512, top_k=int(g["k"]))
    out = R.objective_forward(params(g), g["x"], cfg, toks_since_active=None, training=False)
    assert out.aux.item() == 0.0 and out.n_dead == 0 == g["n_dead"]
    torch.testing.assert_close(out.mse, torch.as_tensor(g["mse"]), **TIGHT)
    torch.testing.assert_close(out.f, g["f"], **TIGHT)


def test_g6_remove_parallel_grads():
    g = load_golden("g6_rpg")
    out = R.remove_parallel_grads(g["g_in"], g["W_dec"])
    torch.testing.assert_close(out, g["g_out"], **TIGHT)
    assert torch.equal(out[5], g["g_in"][5]), "zero-norm row untouched"


@pytest.mark.parametrize("tag", ["clipped", "unclipped"])
def test_g7_clip(tag):
    g = load_golden(f"g7_clip_{tag}")
    outs, total = R.clip_grad_norm([g[f"in{i}"] for i in range(4)], 1.0)
    torch.testing.assert_close(total, torch.as_tensor(g["total"]), **TIGHT)
    for i in range(4):
        torch.testing.assert_close(outs[i], g[f"out{i}"], **TIGHT)


def test_g8_adam_five_steps_first_lr_zero():
    g = load_golden("g8_adam")
    p = g["p0"].clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for i, lr in enumerate(g["lrs"].tolist()):
        R.adam_update(p, g["grads"][i], m, v, i + 1, lr)
        torch.testing.assert_close(p, g["p"][i], rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(m, g["m"][i], **TIGHT)
        torch.testing.assert_close(v, g["v"][i], **TIGHT)
    assert torch.equal(g["p"][0], g["p0"]), "lr = 0 on the first step leaves p unchanged"


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_g9_train_trajectory_and_eval(tag):
    g = load_golden(f"g9_train_{tag}")
    d, s, k, bsz = int(g["d"]), int(g["s"]), int(g["k"]), int(g["bsz"])
    cfg = R.RefConfig(d_model=d, d_sae=s, top_k=k, k_aux=int(g["k_aux"]), dead_threshold_tokens=int(g["thr"]),
                      lr=float(g["lr"]), n_lr_warmup=int(g["n_warm"]), grad_clip=float(g.get("grad_clip", 1.0)))
    if tag == "c":  # the reference's own run with the clip active on every step (train.py:356-362)
        assert (g["log_metrics_grad_norm"].numpy() > cfg.grad_clip).all()
    init = {key: g["init_" + key] for key in R.PARAM_ORDER}
    batches = list(g["acts"].split(bsz))
    state, log = R.train_loop(init, batches, cfg, n_train=int(g["n_train"]), batch_size=bsz)
    assert len(log) == g["n_steps"]
    for name, key in (("mse", "log_loss_mse"), ("aux", "log_loss_aux"), ("l0", "log_loss_l0"), ("l1", "log_loss_l1"),
                      ("grad_norm", "log_metrics_grad_norm"), ("lr", "log_progress_learning_rate")):
        got = np.array([rec[name] for rec in log])
        np.testing.assert_allclose(got, g[key].numpy(), rtol=2e-5, atol=1e-9, err_msg=name)
    assert [rec["n_dead"] for rec in log] == [int(v) for v in g["log_loss_n_dead"].tolist()]
    assert torch.equal(state.toks_since_active, g["toks_final"])
    for key in R.PARAM_ORDER:
        torch.testing.assert_close(state.params[key], g["final_" + key], rtol=1e-4, atol=1e-6)
    ev = R.evaluate(state.params, g["val"].split(bsz), cfg)
    for key in ("l0", "l1", "mse", "normalized_mse", "sse_sae", "sse_baseline"):
        assert math.isclose(ev[key], float(g["ev_" + key]), rel_tol=1e-5), key
    for key in ("n_dead", "n_almost_dead", "n_dense"):
        assert ev[key] == int(g["ev_" + key]), key
    torch.testing.assert_close(ev["freqs"], g["ev_freqs"])
    if tag == "b":
        assert max(rec["n_dead"] for rec in log) > 0, "fixture b exercises AuxK"


def test_g11_schedule_and_limiter():
    g = load_golden("g11_schedule")
    for tag in "abc":
        a = g[f"args_{tag}"].tolist()
        sc = R.WarmupCosine(a[0], int(a[1]), a[2], int(a[3]), a[4])
        got = [sc.step() for _ in range(len(g[f"lr_{tag}"]))]
        np.testing.assert_allclose(got, g[f"lr_{tag}"].numpy(), rtol=0, atol=0)
    for n_rows, bsz, n_train, ln, n_steps, n_rows_seen in g["limiter"].tolist():
        batches = list(torch.zeros(n_rows, 1).split(bsz))
        sizes = [len(b) for b in R.limited_batches(batches, n_train, bsz, drop_last=False)]
        assert math.ceil(n_train / bsz) == ln
        assert (len(sizes), sum(sizes)) == (n_steps, n_rows_seen)


def test_g12_checkpoint_header_shape():
    g = load_golden("g12_checkpoint")
    hdr = json.loads(bytes(g["header_json"].numpy().tolist()).decode())
    assert hdr["schema"] == 5
    assert hdr["cfg"]["activation"]["cls"] == "TopK"
    assert hdr["cfg"]["activation"]["params"]["aux"] == {"cls": "AuxK", "params": {"key": "auxk", "k_aux": 7, "alpha": 0.125}}
    assert list(g["keys"]) == ["W_dec", "b_dec", "W_enc", "b_enc"]


@pytest.mark.parametrize("tag", ["nodead", "dead"])
def test_g13_matryoshka_fixed_prefixes(tag):
    g = load_golden(f"g13_matryoshka_{tag}")
    cfg = R.RefConfig(d_model=64, d_sae=512, top_k=int(g["k"]), k_aux=int(g["k_aux"]), alpha=float(g["alpha"]),
                      dead_threshold_tokens=int(g["thr"]), n_prefixes=4)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params(g).items()}
    toks = g["toks_before"].clone()
    out = R.objective_forward(leaves, g["x"], cfg, toks_since_active=toks, training=True, prefixes=g["prefixes"])
    out.loss.backward()
    assert torch.equal(toks, g["toks_after"]) and out.n_dead == g["n_dead"]
    torch.testing.assert_close(out.mse.detach(), torch.as_tensor(g["mse"]), **TIGHT)
    torch.testing.assert_close(out.aux.detach(), torch.as_tensor(g["aux"]), **TIGHT)
    torch.testing.assert_close(out.x_hats.detach(), g["x_hats"], rtol=1e-5, atol=1e-6)
    for k in R.PARAM_ORDER:
        torch.testing.assert_close(leaves[k].grad, g["g_" + k], rtol=1e-5, atol=1e-8)
