"""The reference's own known-answer tests for the path, restated against the CPU oracle.

Each case names the reference test it restates (tests/test_auxk.py, tests/test_nn_activations.py,
tests/test_nn_objectives.py, tests/test_nn_modeling.py under /root/reference).  CPU only.
The same scenarios -- and fixtures G3, G4, G6, G7, G8 -- are fed to the HIP kernels in tests/test_gpu_known_answers.py."""

import pytest
import torch

import sae_ref as R


def eye_params(d):
    return {"W_dec": torch.eye(d), "b_dec": torch.zeros(d), "W_enc": torch.eye(d), "b_enc": torch.zeros(d)}


def aux(pre, x, x_hat, dead, k_aux, alpha=1.0, d=4):
    p = eye_params(d)
    return R.auxk_loss(x=x, h=pre, x_hat_last=x_hat, dead_mask=dead, W_dec=p["W_dec"], b_dec=p["b_dec"],
                       k_aux=k_aux, alpha=alpha)


# ---- tests/test_nn_activations.py:29-94 -----------------------------------------------------


def test_topk_basic_forward():
    x = torch.tensor([[5.0, 1.0, 3.0, 2.0], [2.0, 4.0, 1.0, 3.0]])
    want = torch.tensor([[5.0, 0.0, 3.0, 0.0], [0.0, 4.0, 0.0, 3.0]])
    torch.testing.assert_close(R.topk_activation(x, 2), want)


def test_topk_ties_select_exactly_k():
    y = R.topk_activation(torch.full((1, 4), 2.0), 2)
    assert (y != 0).sum() == 2 and y[y != 0].unique().item() == 2.0


def test_topk_k_equals_size():
    x = torch.tensor([[5.0, 1.0, 3.0, 2.0]])
    torch.testing.assert_close(R.topk_activation(x, 4), x)


def test_topk_negative_values_kept():
    x = torch.tensor([[-5.0, -1.0, -3.0, -2.0]])
    torch.testing.assert_close(R.topk_activation(x, 2), torch.tensor([[0.0, -1.0, 0.0, -2.0]]))


def test_topk_gradient_is_selection_mask():
    x = torch.tensor([[5.0, 1.0, 3.0, 2.0], [2.0, 4.0, 1.0, 3.0]], requires_grad=True)
    R.topk_activation(x, 2).sum().backward()
    torch.testing.assert_close(x.grad, torch.tensor([[1.0, 0.0, 1.0, 0.0], [0.0, 1.0, 0.0, 1.0]]))


# ---- tests/test_nn_objectives.py:13-52 ------------------------------------------------------


def test_mse_same():
    x = torch.ones(45, 12)
    torch.testing.assert_close(R.mean_squared_err(x.clone(), x), torch.zeros(45, 12))


def test_mse_zero_x_hat():
    torch.testing.assert_close(R.mean_squared_err(torch.zeros(3, 2), torch.ones(3, 2)), torch.ones(3, 2))


def test_mse_nonzero_matches_plain_square():
    x, x_hat = torch.full((3, 2), 3.0), torch.ones(3, 2)
    torch.testing.assert_close(R.mean_squared_err(x_hat, x), (x_hat - x) ** 2)


def test_mse_large_x_is_finite():
    out = R.mean_squared_err(torch.ones(3, 2), torch.full((3, 2), 3e18))
    assert torch.isfinite(out).all()


# ---- tests/test_auxk.py:25-353 --------------------------------------------------------------


def test_auxk_zero_dead_returns_zero():
    loss = aux(torch.ones(2, 4), torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(4, dtype=torch.bool), 2)
    assert loss.item() == 0.0


def test_auxk_topk_value_matches_manual():
    loss = aux(torch.tensor([[1.0, 2.0, 3.0, 4.0]]), torch.zeros(1, 4), torch.zeros(1, 4), torch.ones(4, dtype=torch.bool), 2)
    assert torch.allclose(loss, torch.tensor(6.25))


def test_auxk_alpha_scales_loss():
    pre = torch.tensor([[1.0, 2.0, 3.0, 4.0]])
    a = aux(pre, torch.zeros(1, 4), torch.zeros(1, 4), torch.ones(4, dtype=torch.bool), 2, alpha=1.0)
    b = aux(pre, torch.zeros(1, 4), torch.zeros(1, 4), torch.ones(4, dtype=torch.bool), 2, alpha=0.5)
    assert torch.allclose(b, a * 0.5)


def test_auxk_clamps_k_to_dead_count():
    loss = aux(torch.tensor([[0.0, 0.0, 5.0, 0.0]]), torch.zeros(1, 4), torch.zeros(1, 4),
               torch.tensor([False, True, True, False]), 8)
    assert torch.allclose(loss, torch.tensor(6.25))


def test_auxk_gradients_only_on_dead_selected_latents():
    pre = torch.tensor([[1.0, 2.0, 3.0, 0.5]], requires_grad=True)
    aux(pre, torch.zeros(1, 4), torch.zeros(1, 4), torch.tensor([False, True, True, False]), 1).backward()
    assert pre.grad[0, 2] != 0 and pre.grad[0, 1] == 0 and pre.grad[0, 0] == 0 and pre.grad[0, 3] == 0


def test_auxk_gradients_flow_to_decoder_dead_rows_only():
    p = eye_params(4)
    W = p["W_dec"].requires_grad_(True)
    R.auxk_loss(x=torch.zeros(1, 4), h=torch.tensor([[1.0, 0.0, 3.0, 0.0]]), x_hat_last=torch.zeros(1, 4),
                dead_mask=torch.tensor([True, True, False, False]), W_dec=W, b_dec=p["b_dec"], k_aux=1,
                alpha=1.0).backward()
    assert W.grad[0].abs().sum() > 0 and W.grad[2].abs().sum() == 0


def test_auxk_with_nonzero_residual():
    loss = aux(torch.tensor([[0.0, 0.0, 3.0, 4.0]]), torch.tensor([[1.0, 2.0, 0.0, 0.0]]), torch.zeros(1, 4),
               torch.ones(4, dtype=torch.bool), 2)
    assert torch.allclose(loss, torch.tensor(7.5))


def test_auxk_detaches_residual_from_live_path():
    x_hat = torch.tensor([[0.5, 0.5, 0.0, 0.0]], requires_grad=True)
    pre = torch.tensor([[0.0, 0.0, 3.0, 4.0]], requires_grad=True)
    aux(pre, torch.tensor([[1.0, 2.0, 0.0, 0.0]]), x_hat, torch.ones(4, dtype=torch.bool), 2).backward()
    assert x_hat.grad is None or torch.all(x_hat.grad == 0)
    assert pre.grad.abs().sum() > 0


def test_auxk_uses_preacts_not_postacts():
    p = eye_params(4)
    x = torch.tensor([[1.0, 2.0, 3.0, 4.0]])
    h = R.encode_pre(x, p["W_enc"], p["b_enc"])
    f = R.topk_activation(h, 2)
    assert f[0].tolist() == [0.0, 0.0, 3.0, 4.0] and h[0].tolist() == [1.0, 2.0, 3.0, 4.0]
    x_hat = R.decode(f, p["W_dec"], p["b_dec"])[:, -1]
    loss = aux(h, x, x_hat, torch.tensor([True, True, False, False]), 2)
    assert torch.allclose(loss, torch.zeros(()), atol=1e-6)


def test_auxk_batch_aggregation():
    pre = torch.tensor([[0.0, 0.0, 3.0, 1.0], [0.0, 0.0, 1.0, 5.0]])
    loss = aux(pre, torch.zeros(2, 4), torch.zeros(2, 4), torch.ones(4, dtype=torch.bool), 1)
    assert torch.allclose(loss, torch.tensor(4.25))


def test_decode_returns_prefix_dim():
    p = eye_params(3)
    assert R.decode(torch.ones(2, 3), p["W_dec"], p["b_dec"]).shape == (2, 1, 3)


def test_objective_total_is_mse_plus_aux():
    cfg = R.RefConfig(d_model=4, d_sae=4, top_k=4, k_aux=2, alpha=1.0, normalize_w_dec=False, remove_parallel_grads=False)
    out = R.objective_forward(eye_params(4), torch.randn(2, 4), cfg, toks_since_active=torch.zeros(4, dtype=torch.int64))
    assert torch.allclose(out.loss, out.mse + out.aux)


def test_eval_mode_aux_zero_and_no_tracking():
    cfg = R.RefConfig(d_model=4, d_sae=4, top_k=2, k_aux=2, alpha=1.0)
    out = R.objective_forward(eye_params(4), torch.tensor([[1.0, 2.0, 3.0, 4.0]]), cfg, toks_since_active=None, training=False)
    assert out.aux.item() == 0.0 and out.n_dead == 0


def test_n_dead_tracks_dead_latents():
    thr = 10
    cfg = R.RefConfig(d_model=4, d_sae=4, top_k=2, dead_threshold_tokens=thr, normalize_w_dec=False)
    p = eye_params(4)
    toks = torch.zeros(4, dtype=torch.int64)
    x = torch.tensor([[2.0, 1.0, 0.0, -1.0]] * 2)
    assert R.objective_forward(p, x, cfg, toks_since_active=toks).n_dead == 0
    for _ in range(thr // 2 + 1):
        out = R.objective_forward(p, x, cfg, toks_since_active=toks)
    assert out.n_dead == 2
    x2 = torch.tensor([[0.0, 1.0, 3.0, -1.0]] * 2)
    assert R.objective_forward(p, x2, cfg, toks_since_active=toks).n_dead == 1


# ---- tests/test_nn_modeling.py:323-338 ------------------------------------------------------


def test_remove_parallel_grads_orthogonal_for_unnormalised_rows():
    g = torch.Generator().manual_seed(0)
    W, grad = torch.randn(4, 4, generator=g), torch.randn(4, 4, generator=g)
    out = R.remove_parallel_grads(grad, W)
    assert torch.allclose((out * W).sum(1), torch.zeros(4), atol=1e-6)
