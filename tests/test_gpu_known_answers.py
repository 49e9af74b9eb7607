"""The reference's own known-answer tests and golden vectors, fed to the HIP kernels (needs an MI355X: -m gpu).

tests/test_oracle_known_answers.py restates these scenarios against the CPU oracle; here the SAME scenarios reach
`aux_small_*` / the dense AuxK algebra, `decode_kernel`'s rescaled MSE, `rpg_kernel`, the clip inside the tail and
`adam_kernel` through the C ABI.  Reference tests restated: tests/test_auxk.py:25-353, tests/test_nn_objectives.py:13-52,
tests/test_nn_activations.py:318-348; fixtures G3 (MSE incl. |x| ~ 1e20), G4 (AuxK value + gradients for n_dead <, =, > k_aux),
G6 (remove_parallel_grads), G7 (clip_grad_norm_), G8 (five Adam steps, lr = 0 first).

How a scenario that hands the loss arbitrary (x, pre-activations, x_hat, dead mask) tensors reaches a step that computes
all of those itself -- `embed()`: d_model grows by one indicator dimension per batch row (value c, a power of two).  A
"filler" latent per row reads that dimension with weight BIG / c, wins the top-1 of its row with the code BIG and decodes
to [x_hat[row] | c e_row] exactly, so the main path reconstructs the indicator dimensions without error and produces the
scenario's x_hat on the original ones.  The scenario's own latents keep their decoder rows (zero on the new dimensions) and
get either the scenario's encoder columns or, when the test dictates the pre-activations, weights pre[row, j] / c on the
indicator dimensions.  Everything the auxiliary loss touches is then the scenario's, except the mean's denominator:
n d' instead of n d, i.e. every aux quantity is the reference value times d / d' -- and the indicator rows of dW_enc hand
back the per-row gradient of the pre-activations, c d / d' times the reference's `pre.grad`.
"""

import math

import hypothesis
import hypothesis.strategies as st
import pytest
import torch

import sae_ref as R
from conftest import load_golden

pytestmark = pytest.mark.gpu

BIG = 64.0  # (the opt-in f16x3 encoder pre-scales W_enc by 2^8 before its fp16 split: |W_enc| must stay below 255 there)
THR = 1000


def _engine(d, s, k, *, k_aux, alpha=1.0, thr=THR, max_batch=64, **kw):
    from saev_amd.engine import EngineConfig, SaeEngine

    return SaeEngine(EngineConfig(d_model=d, d_sae=s, top_k=k, k_aux=k_aux, alpha=alpha, dead_threshold_tokens=thr, max_batch=max(max_batch, 8),
                                  normalize_w_dec=False, remove_parallel_grads=False, **kw))


def embed(x, x_hat, dead, *, k_aux, alpha, pre=None, W_enc=None, b_enc=None, W_dec=None, b_dec=None, dense=False, c=1.0,
          training=True):
    """Run one forward + tracker + backward of the embedded scenario (module docstring).  Returns the aux loss, n_dead, the
    route taken and the auxiliary gradients mapped back to the scenario's tensors: ``pre_grad`` (n, T), ``g_W_dec`` (T, d),
    ``g_b_dec`` (d) -- all still carrying the d / d' factor, reported as ``ratio``."""
    n, d = x.shape
    T = dead.numel()
    if W_dec is None:
        W_dec = torch.eye(T, d)
    if b_dec is None:
        b_dec = torch.zeros(d)
    dp = (d + n + 3) // 4 * 4
    sp = (T + n + 3) // 4 * 4
    xp = torch.zeros(n, dp)
    xp[:, :d] = x
    xp[torch.arange(n), d + torch.arange(n)] = c
    We, be = torch.zeros(dp, sp), torch.zeros(sp)
    Wd, bd = torch.zeros(sp, dp), torch.zeros(dp)
    if pre is not None:  # the test dictates the pre-activations
        We[d:d + n, :T] = pre / c
    else:
        We[:d, :T] = W_enc
        be[:T] = b_enc
    Wd[:T, :d] = W_dec
    bd[:d] = b_dec
    for b in range(n):
        We[d + b, T + b] = BIG / c
        Wd[T + b, :d] = (x_hat[b] - b_dec) / BIG
        Wd[T + b, d + b] = c / BIG
    be[T + n:] = -1.0e4  # padding latents: never selected, never dead
    eng = _engine(dp, sp, 1, k_aux=k_aux, alpha=alpha, max_batch=n, aux_small_max=-1 if dense else 0)
    eng.load_params({"W_dec": Wd, "b_dec": bd, "W_enc": We, "b_enc": be})
    toks = torch.zeros(sp, dtype=torch.int64)
    toks[:T][dead] = THR
    eng.set_tracker(toks)
    xg = xp.cuda()
    eng.step_forward(xg, training=training)
    if not training:
        st_ = eng.read_stats()
        return {"aux": st_.aux, "n_dead": st_.n_dead, "eng": eng}
    eng.step_dead(n)
    eng.step_backward()
    st_ = eng.read_stats()
    idx, val, xh = eng.last_codes(n)
    assert idx.cpu().flatten().tolist() == [T + b for b in range(n)], "every row must select its filler latent"
    torch.testing.assert_close(xh.cpu()[:, :d], x_hat, rtol=1e-6, atol=1e-6)
    assert torch.equal(xh.cpu()[:, d:], xp[:, d:]), "indicator dimensions are reconstructed exactly"
    gv = {k: v.cpu() for k, v in eng.grad_views().items()}
    main_db = 2.0 / (n * dp) * (x_hat - x).sum(0)  # what the MSE term alone puts into db_dec on the original dimensions
    out = {
        "aux": st_.aux, "n_dead": st_.n_dead, "route": eng.aux_route(), "ratio": d / dp,
        "pre_grad": gv["W_enc"][d:d + n, :T] / c, "g_W_dec": gv["W_dec"][:T, :d], "g_b_dec": gv["b_dec"][:d] - main_db,
        "g_W_dec_extra": gv["W_dec"][:T, d:], "db_enc": gv["b_enc"][:T], "g_W_enc_model": gv["W_enc"][:d, :T],
        "filler_W_dec": gv["W_dec"][T:T + n], "mse": st_.mse, "x": xp, "dp": dp,
    }
    eng.close()
    return out


def ref_aux(x, pre, x_hat, dead, k_aux, alpha, W_dec=None, b_dec=None):
    """The oracle on the un-embedded scenario: loss and autograd gradients (pre, W_dec, b_dec)."""
    T, d = dead.numel(), x.shape[1]
    W = (torch.eye(T, d) if W_dec is None else W_dec.clone()).requires_grad_(True)
    b = (torch.zeros(d) if b_dec is None else b_dec.clone()).requires_grad_(True)
    h = pre.clone().requires_grad_(True)
    loss = R.auxk_loss(x=x, h=h, x_hat_last=x_hat, dead_mask=dead, W_dec=W, b_dec=b, k_aux=k_aux, alpha=alpha)
    if loss.requires_grad:
        loss.backward()
    z = torch.zeros_like
    return loss.detach(), (h.grad if h.grad is not None else z(h)), (W.grad if W.grad is not None else z(W)), (b.grad if b.grad is not None else z(b))


def check_against_oracle(out, x, pre, x_hat, dead, k_aux, alpha, W_dec=None, b_dec=None, tol=1e-5):
    loss, g_h, g_W, g_b = ref_aux(x, pre, x_hat, dead, k_aux, alpha, W_dec, b_dec)
    r = out["ratio"]
    assert math.isclose(out["aux"], loss.item() * r, rel_tol=1e-5, abs_tol=1e-9), (out["aux"], loss.item() * r)
    torch.testing.assert_close(out["pre_grad"], g_h * r, rtol=tol, atol=1e-8)
    torch.testing.assert_close(out["g_W_dec"], g_W * r, rtol=tol, atol=1e-8)
    torch.testing.assert_close(out["g_b_dec"], g_b * r, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(out["db_enc"], (g_h * r).sum(0), rtol=tol, atol=1e-8)
    assert (out["g_W_dec_extra"] == 0).all(), "the scenario's decoder rows get nothing on the indicator dimensions"
    assert (out["pre_grad"][:, ~dead] == 0).all(), "live latents get no auxiliary gradient"
    return loss.item()


ROUTES = pytest.mark.parametrize("dense", [False, True], ids=["auto-route", "dense-route"])
ALL4 = torch.ones(4, dtype=torch.bool)


# ---- tests/test_auxk.py:25-353, one scenario each, through whichever route the dead count selects and through the dense
# ---- algebra forced (saev_debug_cfg.aux_small_max = -1) -------------------------------------------------------------------


@ROUTES
def test_auxk_zero_dead_returns_zero_and_no_grad(dense):  # test_auxk.py:25-38
    out = embed(torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(4, dtype=torch.bool), k_aux=2, alpha=1.0, pre=torch.ones(2, 4), dense=dense)
    assert out["aux"] == 0.0 and out["n_dead"] == 0 and out["route"] == 0
    assert (out["pre_grad"] == 0).all() and (out["g_W_dec"] == 0).all() and (out["db_enc"] == 0).all()


@ROUTES
def test_auxk_topk_value_matches_manual(dense):  # test_auxk.py:41-53: (3^2 + 4^2) / 4
    x, pre = torch.zeros(1, 4), torch.tensor([[1.0, 2.0, 3.0, 4.0]])
    out = embed(x, torch.zeros(1, 4), ALL4, k_aux=2, alpha=1.0, pre=pre, dense=dense)
    assert math.isclose(out["aux"] / out["ratio"], 6.25, rel_tol=1e-6)
    assert out["n_dead"] == 4 and out["route"] == 3  # four dead latents, two selected: select + mask, i.e. the dense algebra
    check_against_oracle(out, x, pre, torch.zeros(1, 4), ALL4, 2, 1.0)


@ROUTES
def test_auxk_alpha_scales_loss(dense):  # test_auxk.py:56-73
    x, pre = torch.zeros(1, 4), torch.tensor([[1.0, 2.0, 3.0, 4.0]])
    a = embed(x, torch.zeros(1, 4), ALL4, k_aux=2, alpha=1.0, pre=pre, dense=dense)
    b = embed(x, torch.zeros(1, 4), ALL4, k_aux=2, alpha=0.5, pre=pre, dense=dense)
    assert math.isclose(b["aux"], 0.5 * a["aux"], rel_tol=1e-6)
    torch.testing.assert_close(b["pre_grad"], 0.5 * a["pre_grad"], rtol=1e-6, atol=0)


@ROUTES
def test_auxk_clamps_k_to_dead_count(dense):  # test_auxk.py:76-87: k_aux = 8 > 2 dead latents -> 5^2 / 4
    x, pre = torch.zeros(1, 4), torch.tensor([[0.0, 0.0, 5.0, 0.0]])
    dead = torch.tensor([False, True, True, False])
    out = embed(x, torch.zeros(1, 4), dead, k_aux=8, alpha=1.0, pre=pre, dense=dense)
    assert math.isclose(out["aux"] / out["ratio"], 6.25, rel_tol=1e-6) and out["n_dead"] == 2
    assert out["route"] == (3 if dense else 2)  # two dead latents, all selected: the few-dead-latents kernels unless forced
    check_against_oracle(out, x, pre, torch.zeros(1, 4), dead, 8, 1.0)


@ROUTES
def test_auxk_gradients_only_on_dead_selected_latents(dense):  # test_auxk.py:90-105
    x, pre = torch.zeros(1, 4), torch.tensor([[1.0, 2.0, 3.0, 0.5]])
    dead = torch.tensor([False, True, True, False])
    out = embed(x, torch.zeros(1, 4), dead, k_aux=1, alpha=1.0, pre=pre, dense=dense)
    g = out["pre_grad"]
    assert g[0, 2] != 0  # top dead latent selected
    assert g[0, 1] == 0  # dead but not selected by the top-k_aux
    assert g[0, 0] == 0 and g[0, 3] == 0  # live latents
    check_against_oracle(out, x, pre, torch.zeros(1, 4), dead, 1, 1.0)


@ROUTES
def test_auxk_gradients_flow_to_decoder_dead_rows_only(dense):  # test_auxk.py:108-121
    x, pre = torch.zeros(1, 4), torch.tensor([[1.0, 0.0, 3.0, 0.0]])
    dead = torch.tensor([True, True, False, False])
    out = embed(x, torch.zeros(1, 4), dead, k_aux=1, alpha=1.0, pre=pre, dense=dense)
    assert out["g_W_dec"][0].abs().sum() > 0  # dead and selected
    assert out["g_W_dec"][2].abs().sum() == 0  # live: no auxiliary gradient
    check_against_oracle(out, x, pre, torch.zeros(1, 4), dead, 1, 1.0)


@ROUTES
def test_auxk_with_nonzero_residual(dense):  # test_auxk.py:163-182: (1 + 4 + 9 + 16) / 4
    x, pre = torch.tensor([[1.0, 2.0, 0.0, 0.0]]), torch.tensor([[0.0, 0.0, 3.0, 4.0]])
    out = embed(x, torch.zeros(1, 4), ALL4, k_aux=2, alpha=1.0, pre=pre, dense=dense)
    assert math.isclose(out["aux"] / out["ratio"], 7.5, rel_tol=1e-6)
    check_against_oracle(out, x, pre, torch.zeros(1, 4), ALL4, 2, 1.0)


@ROUTES
def test_auxk_detaches_residual_from_live_path(dense):  # test_auxk.py:185-204
    """The residual is a constant of the auxiliary term: what reaches the live path's parameters (the filler latent's
    decoder row, through x_hat) is the MSE gradient and nothing else, while the pre-activations do get a gradient."""
    x, x_hat = torch.tensor([[1.0, 2.0, 0.0, 0.0]]), torch.tensor([[0.5, 0.5, 0.0, 0.0]])
    pre = torch.tensor([[0.0, 0.0, 3.0, 4.0]])
    out = embed(x, x_hat, ALL4, k_aux=2, alpha=1.0, pre=pre, dense=dense)
    assert out["pre_grad"].abs().sum() > 0
    n, d, dp = 1, 4, out["dp"]
    g_main = torch.zeros(1, dp)
    g_main[:, :d] = 2.0 / (n * dp) * (x_hat - x)  # dL_mse / dx_hat; the filler's code is BIG
    torch.testing.assert_close(out["filler_W_dec"], BIG * g_main, rtol=1e-6, atol=0)
    check_against_oracle(out, x, pre, x_hat, ALL4, 2, 1.0)


@ROUTES
def test_auxk_uses_preacts_not_postacts(dense):  # test_auxk.py:207-251
    """Dead latents 0, 1 have pre-activations 1, 2 and post-activations 0 (TopK(2) keeps 3, 4): the auxiliary
    reconstruction [1, 2, 0, 0] equals the residual x - x_hat = [1, 2, 0, 0] -> loss 0."""
    x = torch.tensor([[1.0, 2.0, 3.0, 4.0]])
    x_hat = torch.tensor([[0.0, 0.0, 3.0, 4.0]])  # what the live latents 2, 3 reconstruct
    dead = torch.tensor([True, True, False, False])
    out = embed(x, x_hat, dead, k_aux=2, alpha=1.0, W_enc=torch.eye(4), b_enc=torch.zeros(4), dense=dense)
    assert abs(out["aux"]) < 1e-6 and out["n_dead"] == 2
    check_against_oracle(out, x, x.clone(), x_hat, dead, 2, 1.0)


@ROUTES
def test_auxk_batch_aggregation(dense):  # test_auxk.py:254-274: (9 + 25) / 8
    x = torch.zeros(2, 4)
    pre = torch.tensor([[0.0, 0.0, 3.0, 1.0], [0.0, 0.0, 1.0, 5.0]])
    out = embed(x, torch.zeros(2, 4), ALL4, k_aux=1, alpha=1.0, pre=pre, dense=dense)
    assert math.isclose(out["aux"] / out["ratio"], 4.25, rel_tol=1e-6)
    check_against_oracle(out, x, pre, torch.zeros(2, 4), ALL4, 1, 1.0)
    # per row: only the row's top dead latent carries a gradient
    assert (out["pre_grad"] != 0).tolist() == [[False, False, True, False], [False, False, False, True]]


def test_auxk_eval_mode_returns_zero():  # test_auxk.py:277-288 (dead_mask = None in eval mode: no tracker, no aux)
    out = embed(torch.tensor([[1.0, 2.0, 3.0, 4.0]]), torch.zeros(1, 4), ALL4, k_aux=2, alpha=1.0, pre=torch.tensor([[1.0, 2.0, 3.0, 4.0]]),
                training=False)
    assert out["aux"] == 0.0 and out["n_dead"] == 0
    # test_auxk.py:291-320, the two mask assertions, at the boundary this path has: the tracker phase belongs to a
    # TRAINING forward -- after an eval-mode forward it is refused (the reference asserts "must be None during eval")
    from saev_amd._lib import SaevError

    with pytest.raises(SaevError, match="no training forward"):
        out["eng"].step_dead(1)
    with pytest.raises(SaevError, match="no training forward"):
        out["eng"].step_backward()
    out["eng"].close()


def test_full_backward_updates_mse_and_aux_paths():  # test_auxk.py:148-160
    from saev_amd.nn import modeling as m
    from saev_amd.nn import objectives as o

    cfg = m.SparseAutoencoderConfig(d_model=4, d_sae=4, normalize_w_dec=False, remove_parallel_grads=False,
                                    activation=m.TopK(top_k=4, aux=m.AuxK(k_aux=1, alpha=1.0)))
    sae = m.SparseAutoencoder(cfg)
    with torch.no_grad():
        sae.W_dec.copy_(torch.eye(4)); sae.W_enc.copy_(torch.eye(4)); sae.b_dec.zero_(); sae.b_enc.zero_()
    sae = sae.cuda().train()
    obj = o.get_objective(o.Matryoshka(n_prefixes=1)).train()
    loss, _ = obj(sae, torch.tensor([[1.0, 0.0, 0.0, 0.0]]).cuda())
    assert torch.allclose(loss.loss.detach(), loss.mse + loss.sparsity.to(loss.mse.device) + loss.aux)  # test_auxk.py:132-138
    loss.loss.backward()
    assert sae.W_dec.grad is not None and sae.W_enc.grad is not None


# ---- G4: the reference's AuxK.loss on random tensors, n_dead <, =, > k_aux: value and all gradients --------------------------


@ROUTES
@pytest.mark.parametrize("tag", ["lt", "eq", "gt"])
def test_g4_auxk_value_and_gradients_on_the_hip_path(tag, dense):
    g = load_golden(f"g4_auxk_{tag}")
    k_aux, alpha = int(g["k_aux"]), float(g["alpha"])
    out = embed(g["x"], g["x_hat"], g["dead"], k_aux=k_aux, alpha=alpha, W_enc=g["W_enc"], b_enc=g["b_enc"], W_dec=g["W_dec"],
                b_dec=g["b_dec"], dense=dense)
    r = out["ratio"]
    n_dead = int(g["dead"].sum())
    assert out["n_dead"] == n_dead
    assert out["route"] == (3 if (dense or n_dead > k_aux) else 2)
    assert math.isclose(out["aux"], float(g["loss"]) * r, rel_tol=2e-5)
    # the reference's own gradients (fixture), not only the oracle's
    torch.testing.assert_close(out["pre_grad"], g["g_h"] * r, rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(out["g_W_dec"], g["g_W_dec"] * r, rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(out["g_b_dec"], g["g_b_dec"] * r, rtol=1e-3, atol=1e-7)
    torch.testing.assert_close(out["g_W_enc_model"], g["x"].t() @ (g["g_h"] * r), rtol=1e-4, atol=1e-8)
    assert (out["pre_grad"][:, ~g["dead"]] == 0).all()
    assert ((out["pre_grad"] != 0).sum(1) == min(k_aux, n_dead)).all(), "exactly min(k_aux, n_dead) latents per row carry a gradient"


# ---- tests/test_nn_objectives.py:13-52 and G3 through decode_kernel -----------------------------------------------------------


def mse_through_decode(x_hat, x):
    """x_hat (n, d) reaches decode_kernel as the decode of one code per row (module docstring; eval-mode forward).  Returns
    (mean of the reference's per-element MSE as the kernel reports it, fp64 SSE, max|x|)."""
    n, d = x.shape
    upper = x.abs().max().item()
    c = 2.0 ** math.floor(math.log2(upper)) if upper > 0 else 1.0  # <= max|x|: the indicator dimensions do not move `upper`
    dp, sp = (d + n + 3) // 4 * 4, (n + 3) // 4 * 4
    xp = torch.zeros(n, dp)
    xp[:, :d] = x
    xp[torch.arange(n), d + torch.arange(n)] = c
    We, Wd = torch.zeros(dp, sp), torch.zeros(sp, dp)
    for b in range(n):
        We[d + b, b] = 1.0 / c
        Wd[b, :d] = x_hat[b]
        Wd[b, d + b] = c
    be = torch.zeros(sp)
    be[n:] = -1.0
    eng = _engine(dp, sp, 1, k_aux=0, max_batch=n)
    eng.load_params({"W_dec": Wd, "b_dec": torch.zeros(dp), "W_enc": We, "b_enc": be})
    eng.step_forward(xp.cuda(), training=False)
    st_ = eng.read_stats()
    idx, val, xh = eng.last_codes(n)
    assert idx.cpu().flatten().tolist() == list(range(n)) and (val.cpu() == 1.0).all()
    assert torch.equal(xh.cpu()[:, :d], x_hat) and torch.equal(xh.cpu()[:, d:], xp[:, d:])
    eng.close()
    return st_.mse * dp / d, st_.sse, st_.upper


def test_mse_same():  # test_nn_objectives.py:13-18
    x = torch.ones(45, 12)
    mse, sse, _ = mse_through_decode(x.clone(), x)
    assert mse == 0.0 and sse == 0.0


def test_mse_zero_x_hat():  # test_nn_objectives.py:21-26
    mse, sse, _ = mse_through_decode(torch.zeros(3, 2), torch.ones(3, 2))
    assert math.isclose(mse, 1.0, rel_tol=1e-6) and math.isclose(sse, 6.0, rel_tol=1e-6)


def test_mse_nonzero_matches_plain_square():  # test_nn_objectives.py:29-34
    mse, _, _ = mse_through_decode(torch.ones(3, 2), torch.full((3, 2), 3.0))
    assert math.isclose(mse, 4.0, rel_tol=1e-6)


@pytest.mark.encoder_modes("f32", "f16r")  # the opt-in f16x3 encoder splits x into fp16 halves without a scale: |x| is limited to the fp16 range there
def test_safe_mse_large_x(encoder_mode):  # test_nn_objectives.py:37-45 (the reference uses 3e28 with norm=True; this path is norm=False, where 3e18 keeps mse * upper^2 finite)
    x, x_hat = torch.full((3, 2), 3e18), torch.ones(3, 2)
    mse, _, upper = mse_through_decode(x_hat, x)
    assert math.isfinite(mse) and upper == torch.tensor(3e18).item()
    want = R.mean_squared_err(x_hat, x).double().mean().item()
    assert math.isfinite(want) and math.isclose(mse, want, rel_tol=1e-5)


def test_g3_golden_mse_through_decode_kernel(encoder_mode):
    g = load_golden("g3_mse")
    x, x_hat = g["x"].reshape(-1, 10), g["x_hat"].reshape(-1, 10)
    mse, sse, _ = mse_through_decode(x_hat, x)
    assert math.isclose(mse, g["mse"].double().mean().item(), rel_tol=1e-5)
    assert math.isclose(sse, g["mse"].double().sum().item(), rel_tol=1e-5)
    if encoder_mode == "f16x3":
        return  # (|x| ~ 1e20 below: see test_safe_mse_large_x)
    x, x_hat = g["big_x"].reshape(-1, 8), g["big_x_hat"].reshape(-1, 8)
    mse, sse, upper = mse_through_decode(x_hat, x)
    assert math.isfinite(mse) and upper == x.abs().max().item()
    assert math.isclose(mse, g["big_mse"].double().mean().item(), rel_tol=1e-5), "the reference's rescaled form at |x| ~ 1e20"


# ---- G6 / G7 / G8 fed straight to the tail's kernels ---------------------------------------------------------------------------


_once = pytest.mark.encoder_modes("f16r")  # independent of the encoder: collected once


@_once
@pytest.mark.parametrize("entry", ["saev_remove_parallel_grads", "saev_tail_prepare"])
def test_g6_remove_parallel_grads_kernel(entry, encoder_mode):
    from saev_amd.engine import EngineConfig, SaeEngine

    g = load_golden("g6_rpg")
    eng = SaeEngine(EngineConfig(d_model=24, d_sae=96, top_k=8, max_batch=8, normalize_w_dec=False, remove_parallel_grads=True))
    eng.view("W_dec").copy_(g["W_dec"])
    eng.grads.zero_()
    eng.view("W_dec", eng.grads).copy_(g["g_in"])
    if entry == "saev_remove_parallel_grads":
        eng.remove_parallel_grads()
    else:
        eng.tail_prepare()  # the generic tail: projection in place + the squares of the projected rows
        want = g["g_out"].double().pow(2).sum().item()
        assert math.isclose(eng.sumsq.item(), want, rel_tol=1e-6)
    out = eng.view("W_dec", eng.grads).cpu()
    torch.testing.assert_close(out, g["g_out"], rtol=1e-6, atol=1e-7)
    assert torch.equal(out[5], g["g_in"][5]), "zero-norm row untouched"


@_once
@pytest.mark.parametrize("tag", ["clipped", "unclipped"])
def test_g7_clip_inside_the_tail(tag, encoder_mode):
    """The clip coefficient is applied inside Adam (the clipped gradient is never written back): Adam's first moment after
    one step from zero is 0.1 x the clipped gradient, which is how the fixture's outputs are observed."""
    from saev_amd.engine import EngineConfig, SaeEngine

    g = load_golden(f"g7_clip_{tag}")
    eng = SaeEngine(EngineConfig(d_model=24, d_sae=96, top_k=8, max_batch=8, normalize_w_dec=False, remove_parallel_grads=False))
    for i, name in enumerate(R.PARAM_ORDER):
        eng.view(name, eng.grads).copy_(g[f"in{i}"])
    eng.step_tail(0.0, 1.0)
    st_ = eng.read_stats()
    assert math.isclose(st_.grad_norm, float(g["total"]), rel_tol=1e-6)
    assert (float(g["total"]) > 1.0) == (tag == "clipped")
    for i, name in enumerate(R.PARAM_ORDER):
        torch.testing.assert_close(eng.view(name, eng.adam_m).cpu() * 10.0, g[f"out{i}"], rtol=2e-6, atol=1e-9)
        # the gradient buffer itself keeps the unclipped values
        assert torch.equal(eng.view(name, eng.grads).cpu(), g[f"in{i}"])


@_once
def test_g8_adam_five_steps_first_lr_zero_on_the_hip_kernel(encoder_mode):
    from saev_amd.engine import EngineConfig, SaeEngine

    g = load_golden("g8_adam")
    n = g["p0"].numel()
    eng = SaeEngine(EngineConfig(d_model=16, d_sae=32, top_k=8, max_batch=8, normalize_w_dec=False, remove_parallel_grads=False))
    assert eng.n_params >= n
    eng.params.zero_()
    eng.params[:n].copy_(g["p0"].flatten())
    for i, lr in enumerate(g["lrs"].tolist()):
        eng.grads.zero_()
        eng.grads[:n].copy_(g["grads"][i].flatten())
        eng.step_tail(lr, -1.0)  # max_norm < 0: no clipping (the fixture is torch.optim.Adam alone)
        torch.testing.assert_close(eng.params[:n].cpu(), g["p"][i].flatten(), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(eng.adam_m[:n].cpu(), g["m"][i].flatten(), rtol=1e-6, atol=1e-9)
        torch.testing.assert_close(eng.adam_v[:n].cpu(), g["v"][i].flatten(), rtol=1e-6, atol=1e-12)
        if i == 0:
            assert torch.equal(eng.params[:n].cpu(), g["p0"].flatten()), "lr = 0 on the first step leaves p unchanged"
    assert (eng.params[n:] == 0).all() and (eng.adam_m[n:] == 0).all()


# ---- tests/test_nn_activations.py:318-348: the gradient of TopK is the selection mask times the upstream gradient ------------


@hypothesis.settings(deadline=None, max_examples=12, suppress_health_check=[hypothesis.HealthCheck.function_scoped_fixture])
@hypothesis.given(top_k=st.sampled_from([1, 2, 4, 8]), batch=st.integers(min_value=1, max_value=8),
                  d_sae=st.integers(min_value=64, max_value=512).map(lambda v: 4 * v), seed=st.integers(0, 10_000))
def test_topk_gradient_properties_on_step_backward(top_k, batch, d_sae, seed):
    """saev_step_backward never materialises the gradient of the pre-activations; the indicator dimensions of the embedding
    hand it back row by row (dW_enc[d + b, j] = c * dL/dh[b, j]).  Properties 1-4 of the reference's test."""
    d, c = 32, 1.0
    gen = torch.Generator().manual_seed(seed)
    dp = (d + batch + 3) // 4 * 4
    x = torch.zeros(batch, dp)
    x[:, :d] = torch.randn(batch, d, generator=gen)
    x[torch.arange(batch), d + torch.arange(batch)] = c
    We = torch.randn(dp, d_sae, generator=gen) / math.sqrt(d)
    We[d:] = 0.0
    Wd = torch.randn(d_sae, dp, generator=gen)
    be, bd = 0.1 * torch.randn(d_sae, generator=gen), 0.1 * torch.randn(dp, generator=gen)
    eng = _engine(dp, d_sae, top_k, k_aux=0, max_batch=batch)
    eng.load_params({"W_dec": Wd, "b_dec": bd, "W_enc": We, "b_enc": be})
    eng.step_forward(x.cuda(), training=True)
    eng.step_dead(batch)
    eng.step_backward()
    idx, val, xh = (t.cpu() for t in eng.last_codes(batch))
    gW = eng.view("W_enc", eng.grads).cpu()
    eng.close()
    h_grad = gW[d:d + batch] / c  # (batch, d_sae)
    fwd_mask = torch.zeros(batch, d_sae, dtype=torch.bool)
    fwd_mask[torch.arange(batch)[:, None], idx.long()] = val != 0
    # 1: gradient sparsity matches the forward pass; 2: exactly k per sample; 3: exact zeros elsewhere
    assert torch.equal(h_grad != 0, fwd_mask)
    assert (h_grad != 0).sum(1).eq(top_k).all()
    assert (h_grad[~fwd_mask] == 0).all()
    # 4: selected elements carry the upstream gradient dL/df = W_dec . dL/dx_hat
    up = (2.0 / (batch * dp) * (xh - x)) @ Wd.t()
    torch.testing.assert_close(h_grad[fwd_mask], up[fwd_mask], rtol=1e-4, atol=1e-7)
